"""Per-phase cycle counts of the register-staged convolution kernel wz_k_conv_rs (first workgroup), batch 8.

The counters are compiled in only with -DWZ_RS_STAMPS=1:
    make -C watsor_amd/csrc clean && make -C watsor_amd/csrc CXXFLAGS_EXTRA=-DWZ_RS_STAMPS=1
"""
import os, sys, ctypes as C
os.environ.setdefault("WATSOR_HIP_DEV", "1")   # tools run on the development library (stage entry points, knobs, profiling)
import numpy as np
sys.path.insert(0, "/root/repo")
os.environ["WZ_MB_DEBUG"] = "1"; os.environ.setdefault("WZ_GRAPH", "0")
from watsor_amd import engine as eb, _lib
from watsor_amd.synth import synthetic_frame, synthetic_weights
from watsor_amd.runtime import HipEngine
path = "/tmp/wz_probe/mi355x.bin"; os.makedirs("/tmp/wz_probe", exist_ok=True)
eb.save_engine(eb.build_engine(synthetic_weights(1234)), path)
e = HipEngine(path, 0, 8, 640, 480)
d = [e.upload(synthetic_frame(640, 480, 1234 + i)) for i in range(8)]
for it in range(3):
    e.submit_device(0, d, [640] * 8, [480] * 8); e.wait(0)
ops = e.ops(); out = np.zeros((len(ops), 16), np.uint64); grp = np.zeros(len(ops), np.int32)
_lib.check(e._lib.wz_debug_mbconv(e._h, C.c_void_p(out.ctypes.data), C.c_void_p(grp.ctypes.data)))
for i, o in enumerate(ops):
    if o["kind"] == 4 or not out[i].any(): continue
    t = out[i].astype(np.int64)
    print("%-32s pairs %2d | first half-step cycles: ds_write(wait rb) %5d  load_b issue %5d  compute %5d  load_a issue %5d  barrier %5d | mover wave: ds_write %5d load_b %5d barrier %5d" % (o["name"].split("/")[-1][:32], t[5], t[0], t[1], t[2], t[3], t[4], t[8], t[9], t[12]))
