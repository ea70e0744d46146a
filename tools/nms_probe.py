import os, sys, ctypes as C
os.environ.setdefault("WATSOR_HIP_DEV", "1")   # tools run on the development library (stage entry points, knobs, profiling)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from watsor_amd import engine as eb, _lib
from watsor_amd.synth import synthetic_frame, synthetic_weights
from watsor_amd.runtime import HipEngine
path = "/tmp/wz_probe/mi355x.bin"; os.makedirs("/tmp/wz_probe", exist_ok=True)
eb.save_engine(eb.build_engine(synthetic_weights(1234)), path)
e = HipEngine(path, 0, 8, 640, 480)
frames = [synthetic_frame(640, 480, 1234 + i) for i in range(8)]
d = [e.upload(f) for f in frames]
for it in range(3):
    e.submit_device(0, d, [640] * 8, [480] * 8); e.wait(0)
out = np.zeros((8, 16), np.uint64)
_lib.check(e._lib.wz_debug_nms(e._h, 8, C.c_void_p(out.ctypes.data)))
for f in range(8):
    t = out[f].astype(np.int64)
    us = lambda a, b: (t[b] - t[a]) / 100.0
    print("   walk: init %.1f | pairwise %.1f | per-class %.1f | rows %.1f" % ((t[11] - t[6]) / 100.0, us(11, 12), us(12, 13), (t[7] - t[13]) / 100.0))
    print("frame %d: thr %.1f us | load keys %.1f | band total %.1f (sort %.1f, gather %.1f, walk %.1f) | out %.1f | total %.1f   cnt %d kept %d processed %d"
          % (f, us(0, 1), us(1, 2), us(2, 3), (t[5] - t[2]) / 100.0, us(5, 6), us(6, 7), us(3, 4), us(0, 4), t[8], t[9], t[10]))
