"""Where a lone batch's time goes on the HOST side: wall time of wz_submit_device (descriptor fill, node-parameter update, hipGraphLaunch) against
the wait behind it, per batch size, and the same with WZ_GRAPH=0 (kernel-by-kernel launches).  GPU tool.
    python tools/submit_probe.py [--robust]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from watsor_amd import engine as eb                                   # noqa: E402
from watsor_amd.runtime import HipEngine                              # noqa: E402
from watsor_amd.synth import synthetic_frame, synthetic_weights      # noqa: E402

path = "/tmp/wz_submit_probe/mi355x.bin"
os.makedirs(os.path.dirname(path), exist_ok=True)
eb.save_engine(eb.build_engine(synthetic_weights(1234), robust="--robust" in sys.argv), path)
eng = HipEngine(path, 0, 8, 640, 480)
import ctypes as C
for n in (8, 1):
    d = [eng.upload(synthetic_frame(640, 480, 1234 + i)) for i in range(n)]
    ptrs = (C.c_void_p * n)(*d)
    ws = (C.c_int32 * n)(*([640] * n))
    hs = (C.c_int32 * n)(*([480] * n))
    lib, h = eng._lib, eng._h
    for _ in range(30):
        lib.wz_submit_device(h, 0, n, ptrs, ws, hs, None)
        lib.wz_wait(h, 0)
    sub, wait = [], []
    for _ in range(300):
        t0 = time.perf_counter()
        lib.wz_submit_device(h, 0, n, ptrs, ws, hs, None)
        t1 = time.perf_counter()
        lib.wz_wait(h, 0)
        t2 = time.perf_counter()
        sub.append((t1 - t0) * 1e6)
        wait.append((t2 - t1) * 1e6)
    print("batch %d, graph %s: wz_submit_device p50 %.1f us (p90 %.1f), wz_wait behind it p50 %.1f us, together %.1f us"
          % (n, os.environ.get("WZ_GRAPH", "1"), np.median(sub), np.percentile(sub, 90), np.median(wait), np.median(np.array(sub) + np.array(wait))))
eng.close()
