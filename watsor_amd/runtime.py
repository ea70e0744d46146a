"""Thin Python handle around one `wz_engine_t` (one MI355X).  numpy in, numpy / ctypes rows out.

Used by the detector plugin (`watsor_amd/detection/hip_gpu.py`), by `bench.py` and by the parity
tests.  No torch here: device memory, streams and graphs are owned by libwatsor_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, List, Optional, Sequence

import numpy as np

from . import _lib
from .share import Detection, DetectionArray, MAX_DETECTIONS

ROW_DTYPE = np.dtype([("label", "<i4"), ("zones", "<i4", (10,)), ("_pad", "<i4"), ("confidence", "<f8"),
                      ("x_min", "<i4"), ("y_min", "<i4"), ("x_max", "<i4"), ("y_max", "<i4")])
assert ROW_DTYPE.itemsize == C.sizeof(Detection) == 72
FMT_RGB24, FMT_NV12, FMT_I420 = _lib.WZ_FMT_RGB24, _lib.WZ_FMT_NV12, _lib.WZ_FMT_I420   # pixel formats of a frame (include/watsor_hip.h)


def device_count() -> int:
    return int(_lib.load().wz_device_count())


SCHEDULES = {"throughput": _lib.WZ_SCHEDULE_THROUGHPUT, "latency": _lib.WZ_SCHEDULE_LATENCY}


def set_schedule(name: str, dev: bool = False) -> None:
    """The process's schedule (include/watsor_hip.h: wz_set_schedule): "latency" | "throughput".  Before its first engine; afterwards
    only the schedule already in force is accepted (ValueError otherwise)."""
    try:
        code = SCHEDULES[str(name).lower()]
    except KeyError:
        raise ValueError("schedule %r: expected one of %s" % (name, ", ".join(sorted(SCHEDULES)))) from None
    lib = _lib.load(dev=dev)
    _lib.check(lib.wz_set_schedule(code), "wz_set_schedule", lib)


def get_schedule(dev: bool = False) -> str:
    """The schedule in force ("latency" | "throughput"); asking fixes it (WZ_SCHEDULE in the environment, else throughput)."""
    code = _lib.load(dev=dev).wz_get_schedule()
    return next(k for k, v in SCHEDULES.items() if v == code)


def device_name(device: int) -> str:
    buf = C.create_string_buffer(256)
    _lib.check(_lib.load().wz_device_name_of(device, buf, 256))
    return buf.value.decode()


class HipEngine:
    def __init__(self, engine_path: str, device: int = 0, max_batch: int = 8, max_width: int = 1920,
                 max_height: int = 1080, dev: Optional[bool] = None, schedule: Optional[str] = None):
        """dev=True: on the DEVELOPMENT library (libwatsor_hip_dev.so: stage-level entry points, per-kernel profiling, the WZ_*
        tuning knobs) -- parity tests, bench.py's roofline table and tools/; the product path never asks for it.  Default: the
        product library, unless WATSOR_HIP_DEV=1 is set (how tools/ switch every engine they create).
        schedule: "latency" | "throughput" -- the PROCESS's launch shapes (`set_schedule`), asked for before the engine exists; None:
        whatever is in force (WZ_SCHEDULE, else throughput)."""
        if dev is None:
            dev = os.environ.get("WATSOR_HIP_DEV", "0") not in ("", "0")
        self.dev = bool(dev)
        self._lib = _lib.load(dev=self.dev)
        if schedule is not None:
            set_schedule(schedule, dev=self.dev)
        self._h = C.c_void_p()
        self.max_batch = max_batch
        rc = self._lib.wz_create(os.fsencode(engine_path), device, max_batch, max_width, max_height, C.byref(self._h))
        self._ck(rc, "wz_create")
        self.input_size = self._lib.wz_input_size(self._h)
        self.precision = self._lib.wz_precision(self._h)
        self.num_anchors = self._lib.wz_num_anchors(self._h)
        self.num_classes = self._lib.wz_num_classes(self._h)
        self.num_slots = self._lib.wz_num_slots(self._h)
        self.hp_blocks = self._lib.wz_hp_blocks(self._h)   # leading blocks with split (hi + lo) matrix operands
        self.schedule = get_schedule(dev=self.dev)
        self._dev_allocs: List[int] = []
        self._bound_arrays = {}                            # (submit_bound before any bind_frames: the engine's EINVAL, not an AttributeError)

    def _ck(self, rc: int, what: str = "") -> None:
        if rc:
            _lib.check(rc, what, self._lib)

    def _dev_only(self, what: str) -> None:
        if not self.dev:
            raise AttributeError("%s needs the development library: HipEngine(..., dev=True) (libwatsor_hip_dev.so, `make dev`)" % what)

    # -- lifecycle ------------------------------------------------------------------------------
    def close(self) -> None:
        if self._h:
            for p in self._dev_allocs:
                self._lib.wz_dev_free(self._h, C.c_void_p(p))
            self._dev_allocs = []
            self._lib.wz_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_name(self) -> str:
        return self._lib.wz_device_name(self._h).decode()

    # -- hot call -------------------------------------------------------------------------------
    @staticmethod
    def _addr(obj) -> int:
        if isinstance(obj, np.ndarray):
            return obj.ctypes.data
        if isinstance(obj, int):
            return obj
        return C.addressof(obj)

    @staticmethod
    def frame_geometry(f: np.ndarray, fmt: int = FMT_RGB24):
        """(width, height) of a frame array: (H,W,3) uint8 for RGB24; for NV12 / I420 the usual planar view (H*3/2, W) [or (H*3/2, W, 1)] uint8
        -- H rows of luma, then H/2 rows holding the subsampled chroma (the bytes a decoder writes with `-pix_fmt nv12` /
        `yuv420p`).  Raises ValueError for anything else."""
        if f.dtype != np.uint8:
            raise ValueError("frames are uint8 arrays")
        if fmt == FMT_RGB24:
            if f.ndim != 3 or f.shape[2] != 3:
                raise ValueError("an RGB24 frame must be (H,W,3) uint8")
            return f.shape[1], f.shape[0]
        if fmt in (FMT_NV12, FMT_I420):
            if f.ndim == 3 and f.shape[2] == 1:      # (H*3/2, W, 1): a one-"channel" frame buffer of the reference (share.py:37-40)
                f = f[:, :, 0]
            if f.ndim != 2 or f.shape[0] % 3 or (f.shape[0] // 3 * 2) % 2 or f.shape[1] % 2:
                raise ValueError("an NV12 / I420 frame must be (H*3/2, W) uint8 with even H and W")
            return f.shape[1], f.shape[0] // 3 * 2
        raise ValueError("unknown pixel format %r" % (fmt,))

    def detect_batch(self, frames: Sequence[np.ndarray], out_rows: Sequence, cams: Optional[Sequence[int]] = None,
                     out_pass: Optional[Sequence[np.ndarray]] = None, formats: Optional[Sequence[int]] = None) -> float:
        """frames: (H,W,3) uint8 C-contiguous arrays (host) -- or, with `formats[i]` = FMT_NV12 / FMT_I420, (H*3/2, W) planar
        views; out_rows[i]: ctypes Detection[100] (or a ROW_DTYPE array of 100) written in place.  Returns the batch wall
        time in ms."""
        n = len(frames)
        ptrs = (C.c_void_p * n)()
        ws = (C.c_int32 * n)()
        hs = (C.c_int32 * n)()
        outs = (C.c_void_p * n)()
        keep = []
        for i, f in enumerate(frames):
            try:
                ws[i], hs[i] = self.frame_geometry(f, formats[i] if formats is not None else FMT_RGB24)
            except ValueError as exc:
                raise ValueError("frame %d: %s" % (i, exc)) from None
            if not f.flags["C_CONTIGUOUS"]:
                f = np.ascontiguousarray(f)
            keep.append(f)
            ptrs[i] = f.ctypes.data
            outs[i] = self._addr(out_rows[i])
        fmtv = (C.c_int32 * n)(*[int(x) for x in formats]) if formats is not None else None
        camv = None
        if cams is not None:
            camv = (C.c_int32 * n)(*[int(c) for c in cams])
        passv = None
        if out_pass is not None:
            passv = (C.c_void_p * n)(*[p.ctypes.data for p in out_pass])
        ms = (C.c_float * n)()
        self._ck(self._lib.wz_detect_batch_fmt(self._h, n, ptrs, ws, hs, fmtv, camv, outs, passv, ms))
        return float(ms[0])

    def submit_device(self, slot: int, d_frames: Sequence[int], widths: Sequence[int], heights: Sequence[int],
                      cams: Optional[Sequence[int]] = None, formats: Optional[Sequence[int]] = None) -> None:
        n = len(d_frames)
        ptrs = (C.c_void_p * n)(*d_frames)
        ws = (C.c_int32 * n)(*widths)
        hs = (C.c_int32 * n)(*heights)
        camv = (C.c_int32 * n)(*cams) if cams is not None else None
        fmtv = (C.c_int32 * n)(*[int(x) for x in formats]) if formats is not None else None
        self._ck(self._lib.wz_submit_device_fmt(self._h, slot, n, ptrs, ws, hs, fmtv, camv))

    def submit_host(self, slot: int, frames: Sequence[np.ndarray], cams: Optional[Sequence[int]] = None,
                    formats: Optional[Sequence[int]] = None) -> None:
        """Asynchronous detect of host frames ((H,W,3) uint8, C-contiguous; NV12 / I420: see `frame_geometry`) on lane
        `slot`; collect with `collect()` / `slot_rows()`.  The arrays must stay alive and unchanged until the slot is collected."""
        n = len(frames)
        ws = (C.c_int32 * n)()
        hs = (C.c_int32 * n)()
        for i, f in enumerate(frames):
            try:
                ws[i], hs[i] = self.frame_geometry(f, formats[i] if formats is not None else FMT_RGB24)
            except ValueError as exc:
                raise ValueError("frame %d: %s" % (i, exc)) from None
            if not f.flags["C_CONTIGUOUS"]:
                raise ValueError("frame %d must be C-contiguous (it is read in place, asynchronously)" % i)
        ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        camv = (C.c_int32 * n)(*cams) if cams is not None else None
        fmtv = (C.c_int32 * n)(*[int(x) for x in formats]) if formats is not None else None
        self._ck(self._lib.wz_submit_host_fmt(self._h, slot, n, ptrs, ws, hs, fmtv, camv))

    def host_register(self, arr: np.ndarray) -> None:
        """Page-lock the memory behind `arr` (frames handed over from it then travel by DMA)."""
        self._ck(self._lib.wz_host_register(self._h, C.c_void_p(arr.ctypes.data), arr.nbytes))

    def host_unregister(self, arr: np.ndarray) -> None:
        self._ck(self._lib.wz_host_unregister(self._h, C.c_void_p(arr.ctypes.data)))

    def host_register_address(self, address: int, nbytes: int) -> None:
        """The same for raw memory, e.g. a `multiprocessing.sharedctypes` array shared with other processes."""
        self._ck(self._lib.wz_host_register(self._h, C.c_void_p(address), nbytes))

    def host_unregister_address(self, address: int) -> None:
        self._ck(self._lib.wz_host_unregister(self._h, C.c_void_p(address)))

    # -- the worker's frame table (include/watsor_hip.h: wz_bind_frames) ----------------------------
    def bind_frames(self, pixel_addresses: Sequence[int], widths: Sequence[int], heights: Sequence[int],
                    formats: Sequence[int], cams: Sequence[int], row_addresses: Sequence[int]) -> None:
        """Describe every frame of every frame buffer once: entry i = (address of its pixels, width, height, WZ_FMT_*, camera id
        or -1, address of its `Detection[100]` rows).  Replaces the previous table."""
        n = len(pixel_addresses)
        self._ck(self._lib.wz_bind_frames(
            self._h, n, (C.c_void_p * n)(*pixel_addresses), (C.c_int32 * n)(*widths), (C.c_int32 * n)(*heights),
            (C.c_int32 * n)(*[int(f) for f in formats]), (C.c_int32 * n)(*[int(c) for c in cams]),
            (C.c_void_p * n)(*row_addresses)))
        self._bound_arrays = {}

    def submit_bound(self, slot: int, entries: Sequence[int]) -> None:
        """Asynchronous detect of the frames with these table indices on lane `slot` (one C call, no per-frame work here)."""
        n = len(entries)
        arr_t = self._bound_arrays.get(n)
        if arr_t is None:
            arr_t = self._bound_arrays[n] = C.c_int32 * n
        rc = self._lib.wz_submit_bound(self._h, slot, n, arr_t(*entries))
        if rc:
            self._ck(rc)

    def collect_bound(self, slot: int) -> None:
        """Waits for lane `slot`; its rows are written into the bound frames' own `Detection[100]` arrays."""
        rc = self._lib.wz_collect_bound(self._h, slot)
        if rc:
            self._ck(rc)

    def collect(self, slot: int, out_rows: Sequence, out_pass: Optional[Sequence[np.ndarray]] = None) -> None:
        n = len(out_rows)
        outs = (C.c_void_p * n)(*[self._addr(r) for r in out_rows])
        passv = (C.c_void_p * n)(*[p.ctypes.data for p in out_pass]) if out_pass is not None else None
        self._ck(self._lib.wz_collect(self._h, slot, outs, passv))

    def wait(self, slot: int) -> None:
        self._ck(self._lib.wz_wait(self._h, slot))

    def slot_rows(self, slot: int, n: int) -> np.ndarray:
        """View (no copy) of the pinned result rows of `slot`: ROW_DTYPE [n,100]."""
        p = self._lib.wz_slot_rows(self._h, slot)
        buf = (C.c_uint8 * (72 * MAX_DETECTIONS * n)).from_address(p)
        return np.frombuffer(buf, dtype=ROW_DTYPE).reshape(n, MAX_DETECTIONS)

    def graph_nodes(self, slot: int = 0) -> int:
        """Nodes of the hipGraph last replayed on lane `slot` (kernel launches + the descriptor copy); 0 when graphs are off."""
        return int(self._lib.wz_graph_nodes(self._h, slot))

    def sync(self) -> None:
        self._ck(self._lib.wz_sync(self._h))

    # -- device memory --------------------------------------------------------------------------
    def upload(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        self._ck(self._lib.wz_dev_alloc(self._h, arr.nbytes, C.byref(p)))
        self._dev_allocs.append(p.value)
        self._ck(self._lib.wz_dev_upload(self._h, p, C.c_void_p(arr.ctypes.data), arr.nbytes))
        return p.value

    def free(self, d_ptr: int) -> None:
        self._dev_allocs.remove(d_ptr)
        self._ck(self._lib.wz_dev_free(self._h, C.c_void_p(d_ptr)))

    # -- filters --------------------------------------------------------------------------------
    def set_camera_filter(self, cam: int, width: int, height: int, conf_thr: np.ndarray, area_thr: np.ndarray,
                          zone_fill: Optional[np.ndarray] = None, zone_allow: Optional[np.ndarray] = None) -> None:
        conf = np.ascontiguousarray(conf_thr, np.float64)
        area = np.ascontiguousarray(area_thr, np.float64)
        assert conf.shape == (_lib.WZ_NUM_LABELS,) and area.shape == (_lib.WZ_NUM_LABELS,)
        nz = 0 if zone_fill is None else int(zone_fill.shape[0])
        fill_p = allow_p = None
        if zone_fill is not None and nz == 0:      # a mask without zones: nothing can hit it (mask.py:44-59)
            self._no_zones = np.zeros(1, np.uint8)
            fill_p = C.c_void_p(self._no_zones.ctypes.data)
        if nz:
            zone_fill = np.ascontiguousarray(zone_fill, np.uint8)
            assert zone_fill.shape == (nz, height, width)
            fill_p = C.c_void_p(zone_fill.ctypes.data)
            if zone_allow is not None:
                zone_allow = np.ascontiguousarray(zone_allow, np.uint8)
                assert zone_allow.shape == (_lib.WZ_NUM_LABELS, nz)
                allow_p = C.c_void_p(zone_allow.ctypes.data)
        self._ck(self._lib.wz_set_camera_filter(
            self._h, cam, width, height, conf.ctypes.data_as(_lib.c_f64p), area.ctypes.data_as(_lib.c_f64p),
            nz, allow_p, fill_p))

    def set_camera_drop(self, cam: int, drop: bool) -> None:
        """Rows of camera `cam` that fail its filters are written as all-zero rows (include/watsor_hip.h)."""
        self._ck(self._lib.wz_set_camera_drop(self._h, cam, 1 if drop else 0))

    def clear_camera_filter(self, cam: int) -> None:
        self._ck(self._lib.wz_clear_camera_filter(self._h, cam))

    def filter_rows(self, cam: int, rows) -> np.ndarray:
        """Runs Confidence/Area/Mask of camera `cam` on 100 rows in place; returns pass[100] (uint8)."""
        out = np.zeros(MAX_DETECTIONS, np.uint8)
        self._ck(self._lib.wz_filter_rows(self._h, cam, C.c_void_p(self._addr(rows)), C.c_void_p(out.ctypes.data)))
        return out

    # -- introspection / profiling ---------------------------------------------------------------
    def tensors(self):
        self._dev_only("tensors()")
        out = []
        name = C.create_string_buffer(64)
        h, w, c = C.c_int32(), C.c_int32(), C.c_int32()
        for i in range(self._lib.wz_num_tensors(self._h)):
            self._ck(self._lib.wz_tensor_info(self._h, i, name, 64, C.byref(h), C.byref(w), C.byref(c)))
            out.append((name.value.decode(), h.value, w.value, c.value))
        return out

    def ops(self):
        self._dev_only("ops()")
        out = []
        name = C.create_string_buffer(80)
        dims = (C.c_int32 * 12)()
        keys = ("kind", "cin", "cout", "ksize", "stride", "hin", "win", "hout", "wout", "n_pad", "kc", "cmid")
        for i in range(self._lib.wz_num_ops(self._h)):
            self._ck(self._lib.wz_op_info(self._h, i, name, 80, dims))
            d = dict(zip(keys, list(dims)))
            d["name"] = name.value.decode()
            out.append(d)
        return out

    def stage_names(self) -> List[str]:
        self._dev_only("stage_names()")
        name = C.create_string_buffer(96)
        out = []
        for i in range(self._lib.wz_num_stages(self._h)):
            self._ck(self._lib.wz_stage_name(self._h, i, name, 96))
            out.append(name.value.decode())
        return out

    def profile_device(self, d_frames: Sequence[int], widths: Sequence[int], heights: Sequence[int], reps: int = 10,
                       inner: int = 1):
        """[(stage name, ms of its event bracket)]; inner > 1: every network kernel `inner` times back to back in its bracket."""
        self._dev_only("profile_device()")
        n = len(d_frames)
        ns = self._lib.wz_num_stages(self._h)
        ms = (C.c_float * ns)()
        self._ck(self._lib.wz_profile_stages(self._h, n, (C.c_void_p * n)(*d_frames), (C.c_int32 * n)(*widths),
                                               (C.c_int32 * n)(*heights), reps, inner, ms))
        return list(zip(self.stage_names(), [float(x) for x in ms]))

    # -- stage-level entry points (parity tests) --------------------------------------------------
    def tensor_is_pair(self, idx: int) -> bool:
        self._dev_only("tensor_is_pair()")
        return bool(self._lib.wz_tensor_flags(self._h, idx) & 1)

    def stage_preprocess(self, frame: np.ndarray, fmt: int = FMT_RGB24) -> np.ndarray:
        """(S,S,4) float16; for a pair input tensor (S,S,8): hi (r,g,b,0) then lo (r,g,b,0)."""
        self._dev_only("stage_preprocess()")
        frame = np.ascontiguousarray(frame, np.uint8)
        w, h = self.frame_geometry(frame, fmt)
        S = self.input_size
        out = np.empty((S, S, 8 if self.tensor_is_pair(0) else 4), np.float16)
        self._ck(self._lib.wz_stage_preprocess_fmt(self._h, C.c_void_p(frame.ctypes.data), w, h, int(fmt),
                                                     C.c_void_p(out.ctypes.data)))
        return out

    def stage_forward(self, x_half: np.ndarray):
        """x_half float16 [n,S,S,4] -> (box_enc float32 [n,A,4], logits float32 [n,A,C])."""
        self._dev_only("stage_forward()")
        x = np.ascontiguousarray(x_half, np.float16)
        n = x.shape[0]
        be = np.empty((n, self.num_anchors, 4), np.float32)
        lg = np.empty((n, self.num_anchors, self.num_classes), np.float32)
        self._ck(self._lib.wz_stage_forward(self._h, n, C.c_void_p(x.ctypes.data), C.c_void_p(be.ctypes.data),
                                              C.c_void_p(lg.ctypes.data)))
        return be, lg

    def stage_read_tensor(self, idx: int, frame: int = 0) -> np.ndarray:
        """(h,w,c) array as stored; a pair tensor comes back as float32 hi + lo."""
        self._dev_only("stage_read_tensor()")
        name, h, w, c = self.tensors()[idx]
        pair = self.tensor_is_pair(idx)
        out = np.empty((h, w, 2 * c if pair else c), np.float16 if (self.precision == 16 or name == "input") else np.float32)
        self._ck(self._lib.wz_stage_read_tensor(self._h, idx, frame, C.c_void_p(out.ctypes.data)))
        if pair:
            return out[..., :c].astype(np.float32) + out[..., c:].astype(np.float32)
        return out

    def stage_postprocess(self, box_enc: np.ndarray, logits: np.ndarray):
        self._dev_only("stage_postprocess()")
        be = np.ascontiguousarray(box_enc, np.float32)
        lg = np.ascontiguousarray(logits, np.float32)
        n = be.shape[0]
        boxes = np.empty((n, MAX_DETECTIONS, 4), np.float32)
        scores = np.empty((n, MAX_DETECTIONS), np.float32)
        classes = np.empty((n, MAX_DETECTIONS), np.int32)
        num = np.empty((n,), np.int32)
        self._ck(self._lib.wz_stage_postprocess(
            self._h, n, C.c_void_p(be.ctypes.data), C.c_void_p(lg.ctypes.data), C.c_void_p(boxes.ctypes.data),
            C.c_void_p(scores.ctypes.data), C.c_void_p(classes.ctypes.data), C.c_void_p(num.ctypes.data)))
        return boxes, scores, classes, num

    def stage_rows(self, width: int, height: int, boxes: np.ndarray, scores: np.ndarray, classes: np.ndarray):
        self._dev_only("stage_rows()")
        b = np.ascontiguousarray(boxes, np.float32)
        s = np.ascontiguousarray(scores, np.float32)
        c = np.ascontiguousarray(classes, np.int32)
        rows = np.zeros(MAX_DETECTIONS, ROW_DTYPE)
        self._ck(self._lib.wz_stage_rows(self._h, width, height, C.c_void_p(b.ctypes.data),
                                           C.c_void_p(s.ctypes.data), C.c_void_p(c.ctypes.data),
                                           C.c_void_p(rows.ctypes.data)))
        return rows


def zones_from_alpha(alpha: np.ndarray, max_zones: int = 40):
    """Host-side: alpha plane (H,W) uint8 -> (zone_fill uint8 [nz,H,W], centroids int32 [nz,2])."""
    a = np.ascontiguousarray(alpha, np.uint8)
    h, w = a.shape
    fill = np.zeros((max_zones, h, w), np.uint8)
    cent = np.zeros((max_zones, 2), np.int32)
    rc = _lib.load().wz_zones_from_alpha(C.c_void_p(a.ctypes.data), w, h, max_zones, C.c_void_p(fill.ctypes.data),
                                         C.c_void_p(cent.ctypes.data))
    if rc == _lib.WZ_EFORMAT:
        raise ZeroDivisionError("float division by zero")     # what mask.py:80 raises for a degenerate contour
    if rc < 0:
        raise ValueError("wz_zones_from_alpha failed (%d)" % rc)
    return fill[:rc].copy(), cent[:rc].copy()
