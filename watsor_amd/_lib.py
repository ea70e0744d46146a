"""ctypes binding of libwatsor_hip.so (the C ABI declared in include/watsor_hip.h).

The library is built in-tree (`watsor_amd/libwatsor_hip.so`, see `__graft_entry__.build()` /
`watsor_amd/csrc/Makefile`).  There is deliberately no CPU fallback: if the library is missing or
does not export every declared symbol, importing a detector fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

from .share import Detection

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WATSOR_HIP_LIBRARY") or os.path.join(_HERE, "libwatsor_hip.so")   # (the variable: measurement builds, tools/)

WZ_OK, WZ_EINVAL, WZ_ENOENT, WZ_EFORMAT, WZ_EHIP, WZ_ENODEV, WZ_ELIMIT = 0, -1, -2, -3, -4, -5, -6
WZ_SLOTS = 8
WZ_FMT_RGB24, WZ_FMT_NV12, WZ_FMT_I420 = 0, 1, 2
WZ_NUM_LABELS = 91
WZ_MAX_CAMS = 256

c_u8p = C.POINTER(C.c_uint8)
c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u16p = C.POINTER(C.c_uint16)
DetP = C.POINTER(Detection)

# name -> (restype, argtypes); the single source the "exports every declared symbol" test walks
SIGNATURES = {
    "wz_device_count": (C.c_int, []),
    "wz_device_name_of": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "wz_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "wz_destroy": (None, [C.c_void_p]),
    "wz_device_name": (C.c_char_p, [C.c_void_p]),
    "wz_last_error": (C.c_char_p, []),
    "wz_detect_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_i32p, c_i32p, c_i32p,
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_f32p]),
    "wz_submit_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), c_i32p, c_i32p, c_i32p]),
    "wz_submit_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), c_i32p, c_i32p, c_i32p]),
    "wz_detect_batch_fmt": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_i32p, c_i32p, c_i32p, c_i32p,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_f32p]),
    "wz_submit_device_fmt": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), c_i32p, c_i32p, c_i32p, c_i32p]),
    "wz_submit_host_fmt": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), c_i32p, c_i32p, c_i32p, c_i32p]),
    "wz_frame_bytes": (C.c_uint64, [C.c_int, C.c_int, C.c_int]),
    "wz_host_register": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "wz_host_unregister": (C.c_int, [C.c_void_p, C.c_void_p]),
    "wz_collect": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "wz_bind_frames": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_i32p, c_i32p, c_i32p, c_i32p, C.POINTER(C.c_void_p)]),
    "wz_submit_bound": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_i32p]),
    "wz_collect_bound": (C.c_int, [C.c_void_p, C.c_int]),
    "wz_wait": (C.c_int, [C.c_void_p, C.c_int]),
    "wz_slot_rows": (C.c_void_p, [C.c_void_p, C.c_int]),
    "wz_sync": (C.c_int, [C.c_void_p]),
    "wz_num_slots": (C.c_int, [C.c_void_p]),
    "wz_graph_nodes": (C.c_int, [C.c_void_p, C.c_int]),
    "wz_set_camera_filter": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_f64p, c_f64p, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    "wz_clear_camera_filter": (C.c_int, [C.c_void_p, C.c_int]),
    "wz_filter_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "wz_set_camera_drop": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "wz_tracker_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "wz_tracker_destroy": (None, [C.c_void_p]),
    "wz_tracker_reset": (C.c_int, [C.c_void_p]),
    "wz_tracker_count": (C.c_int, [C.c_void_p]),
    "wz_tracker_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "wz_tracker_sieve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "wz_debug_pyset_order": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "wz_debug_unused_order": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "wz_zones_from_alpha": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "wz_input_size": (C.c_int, [C.c_void_p]),
    "wz_precision": (C.c_int, [C.c_void_p]),
    "wz_num_anchors": (C.c_int, [C.c_void_p]),
    "wz_num_classes": (C.c_int, [C.c_void_p]),
    "wz_num_tensors": (C.c_int, [C.c_void_p]),
    "wz_tensor_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, c_i32p, c_i32p, c_i32p]),
    "wz_tensor_flags": (C.c_int, [C.c_void_p, C.c_int]),
    "wz_hp_blocks": (C.c_int, [C.c_void_p]),
    "wz_num_ops": (C.c_int, [C.c_void_p]),
    "wz_op_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, c_i32p]),
    "wz_num_stages": (C.c_int, [C.c_void_p]),
    "wz_stage_name": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "wz_profile_device": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_i32p, c_i32p, C.c_int, c_f32p]),
    "wz_profile_stages": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_i32p, c_i32p, C.c_int, C.c_int, c_f32p]),
    "wz_debug_nms": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "wz_debug_mbconv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "wz_dev_alloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "wz_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "wz_dev_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "wz_dev_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "wz_stage_preprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "wz_stage_preprocess_fmt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "wz_stage_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "wz_stage_read_tensor": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "wz_stage_postprocess": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "wz_stage_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


class HipLibraryMissing(ImportError):
    pass


def load():
    """dlopen the library once and attach prototypes.  Raises instead of falling back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipLibraryMissing(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C watsor_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)         # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().wz_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    """Map a WZ_E* return code to the exception the reference's worker expects (detector.py:97-100)."""
    if rc == WZ_OK:
        return
    msg = last_error() or what
    if rc == WZ_ENOENT:
        raise FileNotFoundError(msg)
    if rc in (WZ_EINVAL, WZ_ELIMIT):
        raise ValueError(msg)
    if rc == WZ_EFORMAT:
        raise ValueError(msg)
    if rc == WZ_ENODEV:
        raise RuntimeError(msg)
    raise RuntimeError(msg)
