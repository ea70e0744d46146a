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
LIB_PATH = os.path.join(_HERE, "libwatsor_hip.so")
# the development build (-DWZ_DEV_BUILD): stage-level entry points, profiling hooks, tuning knobs; WATSOR_HIP_DEV_LIBRARY points
# tools at a measurement build of it (e.g. one compiled with -DWZ_HP_STAMPS=1)
DEV_LIB_PATH = os.environ.get("WATSOR_HIP_DEV_LIBRARY") or os.path.join(_HERE, "libwatsor_hip_dev.so")

WZ_OK, WZ_EINVAL, WZ_ENOENT, WZ_EFORMAT, WZ_EHIP, WZ_ENODEV, WZ_ELIMIT, WZ_EINCOMPLETE = 0, -1, -2, -3, -4, -5, -6, -7
WZ_SLOTS = 8
WZ_FMT_RGB24, WZ_FMT_NV12, WZ_FMT_I420 = 0, 1, 2
WZ_NUM_LABELS = 91
WZ_MAX_CAMS = 256
WZ_SCHEDULE_THROUGHPUT, WZ_SCHEDULE_LATENCY = 0, 1

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
    "wz_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "wz_set_schedule": (C.c_int, [C.c_int]),
    "wz_get_schedule": (C.c_int, []),
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
    "wz_zones_from_alpha": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "wz_input_size": (C.c_int, [C.c_void_p]),
    "wz_precision": (C.c_int, [C.c_void_p]),
    "wz_num_anchors": (C.c_int, [C.c_void_p]),
    "wz_num_classes": (C.c_int, [C.c_void_p]),
    "wz_hp_blocks": (C.c_int, [C.c_void_p]),
    "wz_dev_alloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "wz_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "wz_dev_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "wz_dev_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
}

# ... and what only the development library (libwatsor_hip_dev.so, `make dev`) exports on top of them
DEV_SIGNATURES = {
    "wz_debug_pyset_order": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "wz_debug_unused_order": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "wz_num_tensors": (C.c_int, [C.c_void_p]),
    "wz_tensor_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, c_i32p, c_i32p, c_i32p]),
    "wz_tensor_flags": (C.c_int, [C.c_void_p, C.c_int]),
    "wz_num_ops": (C.c_int, [C.c_void_p]),
    "wz_op_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, c_i32p]),
    "wz_num_stages": (C.c_int, [C.c_void_p]),
    "wz_stage_name": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "wz_profile_device": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_i32p, c_i32p, C.c_int, c_f32p]),
    "wz_profile_stages": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_i32p, c_i32p, C.c_int, C.c_int, c_f32p]),
    "wz_debug_nms": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "wz_debug_mbconv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "wz_debug_lane_stamps": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "wz_debug_lane_launch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, c_i32p]),
    "wz_stage_preprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "wz_stage_preprocess_fmt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "wz_stage_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "wz_stage_read_tensor": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "wz_stage_postprocess": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "wz_stage_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None
_dev_lib = None


class HipLibraryMissing(ImportError):
    pass


class RowsIncomplete(ValueError):
    """WZ_EINCOMPLETE: the call wrote every row, but a frame's rows may be short (clip-after-NMS engines).  The ONLY failure after
    which the rows are valid: callers that must tell "rows written" from "rows not written" catch this class, not ValueError."""


def _open(path, signatures):
    if not os.path.isfile(path):
        raise HipLibraryMissing(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C watsor_amd/csrc`).  There is no CPU fallback." % path)
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)       # (local: the product and the development library define the same symbols)
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)         # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


def load(dev: bool = False):
    """dlopen the library once and attach prototypes.  Raises instead of falling back.  dev=True: the development library."""
    global _lib, _dev_lib
    if dev:
        if _dev_lib is None:
            _dev_lib = _open(DEV_LIB_PATH, {**SIGNATURES, **DEV_SIGNATURES})
        return _dev_lib
    if _lib is None:
        _lib = _open(LIB_PATH, SIGNATURES)
    return _lib


def last_error(lib=None) -> str:
    """The message of the calling thread's last failure in `lib` (default: whichever of the two libraries is loaded and has one)."""
    libs = [lib] if lib is not None else [l for l in (_lib, _dev_lib) if l is not None] or [load()]
    for l in libs:
        msg = l.wz_last_error().decode("utf-8", "replace")
        if msg:
            return msg
    return ""


def check(rc: int, what: str = "", lib=None) -> None:
    """Map a WZ_E* return code to the exception the reference's worker expects (detector.py:97-100).  `lib`: the library the
    call went to (its message is the one to show)."""
    if rc == WZ_OK:
        return
    msg = last_error(lib) or what
    if rc == WZ_ENOENT:
        raise FileNotFoundError(msg)
    if rc == WZ_EINCOMPLETE:
        raise RowsIncomplete(msg)
    if rc in (WZ_EINVAL, WZ_ELIMIT):
        raise ValueError(msg)
    if rc == WZ_EFORMAT:
        raise ValueError(msg)
    if rc == WZ_ENODEV:
        raise RuntimeError(msg)
    raise RuntimeError(msg)
