"""Cycle counts of the split-operand block kernel wz_k_mbconv_hp (first workgroup, wave 0) per block; batch 8 by default.

The counters are compiled in only with -DWZ_HP_STAMPS=1:
    touch watsor_amd/csrc/k_mbconv_hp.hip && make -C watsor_amd/csrc CXXFLAGS_EXTRA=-DWZ_HP_STAMPS=1
(the stamped build waits for the halo and the first weights before the chunk loop: `landed` is then a latency, not a stall
 spread over the first chunk).
"""
import os, sys, ctypes as C
os.environ.setdefault("WATSOR_HIP_DEV", "1")   # tools run on the development library (stage entry points, knobs, profiling)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WZ_MB_DEBUG"] = "1"; os.environ.setdefault("WZ_GRAPH", "0")
from watsor_amd import engine as eb, _lib
from watsor_amd.synth import synthetic_frame, synthetic_weights
from watsor_amd.runtime import HipEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
path = "/tmp/wz_probe/mi355x.bin"; os.makedirs("/tmp/wz_probe", exist_ok=True)
eb.save_engine(eb.build_engine(synthetic_weights(1234), robust=os.environ.get("WZ_PROBE_ROBUST", "0") == "1"), path)   # WZ_PROBE_ROBUST=1: all 17 blocks
e = HipEngine(path, 0, B, 640, 480)
d = [e.upload(synthetic_frame(640, 480, 1234 + i)) for i in range(B)]
for it in range(3):
    e.submit_device(0, d, [640] * B, [480] * B); e.wait(0)
ops = e.ops(); out = np.zeros((len(ops), 16), np.uint64); grp = np.zeros(len(ops), np.int32)
_lib.check(e._lib.wz_debug_mbconv(e._h, C.c_void_p(out.ctypes.data), C.c_void_p(grp.ctypes.data)))
for i, o in enumerate(ops):
    t = out[i].astype(np.int64)
    if o["kind"] == 4 and t[5] > 0 and t[3] > 0:
        n = max(int(t[5]), 1)
        print("%-18s %3dx%-3d s%d cmid %3d | issue %5d  landed+barrier %5d | first chunk %5d  all %2d chunks %6d (per chunk: expand %4d  depthwise %4d  split+project %4d) | epilogue %5d | total %6d cycles"
              % (o["name"].split("/")[-1], o["hin"], o["win"], o["stride"], o["cmid"], t[0], t[1], t[2], t[5], t[3], t[6] // n, t[7] // n, t[9] // n,
                 t[4], t[0] + t[1] + t[3] + t[4]))
