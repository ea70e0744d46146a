"""Phases of wz_k_nms (per frame) on head outputs shaped like a TRAINED detector's (tests/test_gpu_parity.py:
trained_like_head_outputs) -- the random-init network of the benchmark yields ~270 first-band candidates of mostly one class;
a trained one yields clusters of overlapping same-class anchors, saturated ties, and every logit above the score threshold."""
import ctypes as C
import os
os.environ.setdefault("WATSOR_HIP_DEV", "1")   # tools run on the development library (stage entry points, knobs, profiling)
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as tg                                           # noqa: E402
from watsor_amd import engine as eb, _lib                               # noqa: E402
from watsor_amd.runtime import HipEngine                                # noqa: E402
from watsor_amd.synth import synthetic_weights                         # noqa: E402

path = "/tmp/wz_probe/mi355x.bin"
os.makedirs("/tmp/wz_probe", exist_ok=True)
eb.save_engine(eb.build_engine(synthetic_weights(1234)), path)
e = HipEngine(path, 0, 8, 640, 480)
for label, kw in (("14 objects", dict(n_objects=14)), ("40 objects", dict(n_objects=40)), ("3 objects", dict(n_objects=3))):
    for seed in (1, 2):
        be, lg = tg.trained_like_head_outputs(seed, **kw)
        for _ in range(int(os.environ.get("NMS_WARM", "3"))):              # (the band hint settles)
            e.stage_postprocess(be, lg)
        out = np.zeros((8, 16), np.uint64)
        _lib.check(e._lib.wz_debug_nms(e._h, 1, C.c_void_p(out.ctypes.data)))
        t = out[0].astype(np.int64)
        us = lambda a, b: (t[b] - t[a]) / 100.0
        print("%-11s seed %d: total %6.1f us | load keys %5.1f | sort %5.1f | walk %5.1f (pairwise %5.1f) | candidates %4d kept %3d"
              % (label, seed, us(0, 4), us(1, 2), (t[5] - t[2]) / 100.0, us(6, 7), us(11, 12), t[8], t[9]))
e.close()
