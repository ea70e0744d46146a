"""Cycle counts of the wide head kernel wz_k_conv_wide_group (first workgroup, wave 0), batch 8 by default.

The counters are compiled in only with -DWZ_WIDE_STAMPS=1:
    make -C watsor_amd/csrc clean && make -C watsor_amd/csrc CXXFLAGS_EXTRA=-DWZ_WIDE_STAMPS=1
work = from a barrier's release to the arrival at the next barrier's wait (the MFMAs and everything issued between them),
sync = `s_waitcnt lgkmcnt(0)` + `s_barrier`.  (s_memtime needs an lgkmcnt(0) of its own: the LDS reads of a half are complete
before its first MFMA when stamped, which they need not be otherwise.)
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
eb.save_engine(eb.build_engine(synthetic_weights(1234)), path)
e = HipEngine(path, 0, B, 640, 480)
d = [e.upload(synthetic_frame(640, 480, 1234 + i)) for i in range(B)]
for it in range(3):
    e.submit_device(0, d, [640] * B, [480] * B); e.wait(0)
ops = e.ops(); out = np.zeros((len(ops), 16), np.uint64); grp = np.zeros(len(ops), np.int32)
_lib.check(e._lib.wz_debug_mbconv(e._h, C.c_void_p(out.ctypes.data), C.c_void_p(grp.ctypes.data)))
for i, o in enumerate(ops):
    t = out[i].astype(np.int64)
    if o["name"].startswith("BoxPredictor") and t[2] > 0:
        n = t[2]   # barrier-to-barrier intervals measured: one per step (the barrier sits between a step's two halves)
        print("%-20s steps %3d | per step: work %6.0f  sync %6.0f cycles | prologue %6d  loop %7d  epilogue (stores landed) %6d cycles"
              % (o["name"], t[2], t[0] / n, t[1] / n, t[3], t[4], t[5]))
