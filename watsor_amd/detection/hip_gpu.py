"""`HipObjectDetector` -- the MI355X detector plugin.

Satisfies the duck-typed protocol every reference detector implements
(`watsor/detection/tensorflow_cpu.py:13,64-92`, `tensorrt_gpu.py:20,55-91`,
`tensorflow_lite_cpu.py:16,25-62`, `edge_tpu.py:17,24-57`):

    __init__(model_path, device)    raises FileNotFoundError when model/mi355x.bin is absent
    device_name -> str
    __enter__ / __exit__            frees device state
    detect(image_shape, image_np, detections) -> milliseconds

so `ObjectDetector._run/_next_frame` (`watsor/detection/detector.py:84-112`) drives it unchanged.
`detect()` hands the frame pointer and the ctypes `Detection[100]` array straight to
`wz_detect_batch`; the 100 rows (class, score, pixel box -- `tensorflow_cpu.py:79-90`) are produced
on the GPU and written in place.  `detect_batch()` is the same call for several frames (BASELINE
config 2: batch = 8), used by `BatchedObjectDetector`.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import numpy as np

from ..runtime import FMT_I420, FMT_NV12, FMT_RGB24, HipEngine
from ..share import Detection

# what a decoder's `-pix_fmt` may say (watsor/stream/ffmpeg.py:78-88 reads whatever it writes; the reference's schema asks for
# rgb24, `watsor/config/schema.py:161` -- NV12 / yuv420p frames are half the bytes and are converted on the GPU)
PIXEL_FORMATS = {"rgb24": FMT_RGB24, "nv12": FMT_NV12, "yuv420p": FMT_I420, "i420": FMT_I420}

ENGINE_FILE = "mi355x.bin"       # the analogue of gpu.trt (watsor/detection/detector.py:44)


def pixel_format_code(name) -> int:
    try:
        return PIXEL_FORMATS[str(name).lower()]
    except KeyError:
        raise ValueError("pixel_format %r: expected one of %s" % (name, ", ".join(sorted(PIXEL_FORMATS)))) from None


def frame_formats(frames: Sequence[np.ndarray], cameras: Optional[Sequence[int]], default: int, by_camera: dict):
    """Pixel format of every frame of a call: the camera's configured one, else the detector's; a planar (2-D) array that would
    be read as RGB24 takes the one YUV format that is configured (NV12 if none is).  None = all RGB24."""
    yuv = next((f for f in [default] + list(by_camera.values()) if f != FMT_RGB24), FMT_NV12)
    out = []
    for i, f in enumerate(frames):
        fmt = by_camera.get(cameras[i], default) if cameras is not None else default
        if fmt == FMT_RGB24 and (f.ndim == 2 or (f.ndim == 3 and f.shape[2] == 1)):
            fmt = yuv
        out.append(fmt)
    return out if any(f != FMT_RGB24 for f in out) else None


class HipObjectDetector:
    """Performs object detection on AMD Instinct MI355X GPUs (hand-written HIP kernels)."""

    def __init__(self, model_path, device: int = 0, options: Optional[dict] = None, max_batch: Optional[int] = None,
                 max_width: Optional[int] = None, max_height: Optional[int] = None):
        """`options` (third positional argument, what `create_object_detectors` passes in `detector_args`):
        dict(max_batch=, max_width=, max_height=) -- the factory derives the frame size from the cameras' frame buffers;
        the keyword forms and the WATSOR_HIP_MAX_* environment variables are the fallbacks.
        `options["pixel_format"]`: "rgb24" (default) | "nv12" | "yuv420p", or {camera name: one of these} -- what the cameras'
        decoders write into their frame buffers.  An NV12 / yuv420p frame is handed over as the (H*3/2, W) uint8 array of its
        bytes (that is its `image_shape`).
        `options["schedule"]`: "latency" | "throughput" | "auto" (default) -- the launch shapes of this detector PROCESS
        (include/watsor_hip.h: wz_set_schedule).  "latency" makes a lone batch finish soonest -- the reference's normal load is one
        frame at a time (`_next_frame`, detector.py:102-112); "throughput" gets the most frames per second out of four batches in
        flight.  "auto" leaves what is in force (WZ_SCHEDULE, else throughput) -- the factory resolves it by the number of cameras
        (`hip_detector_options`: latency for up to 4).
        `options["numa"]`: True | False | "auto" (default) -- pin this process to the CPUs local to the GPU before the engine
        allocates its page-locked blocks (watsor_amd/numa.py); "auto": only on hosts with more than one GPU."""
        engine_path = os.path.join(model_path, ENGINE_FILE)
        if not os.path.isfile(engine_path):
            raise FileNotFoundError(engine_path)
        options = options or {}
        schedule = str(options.get("schedule") or "auto").lower()
        if schedule not in ("auto", "latency", "throughput", "auto:latency", "auto:throughput"):
            raise ValueError("schedule %r: expected latency, throughput or auto" % (options.get("schedule"),))
        # "auto:<name>" (what the factory's `auto` resolves to) is a PREFERENCE: the schedule is the process's (wz_set_schedule), and a
        # process that has fixed the other one already -- Thread delegates, an engine or filter created earlier, a second factory
        # call -- keeps it; only an operator's explicit "latency" / "throughput" is a demand (ValueError when it cannot be met)
        soft = schedule.startswith("auto:")
        schedule = None if schedule == "auto" else schedule.split(":")[-1]
        self.schedule_note = None
        if soft:
            from ..runtime import get_schedule, set_schedule
            try:
                set_schedule(schedule)
            except ValueError:
                self.schedule_note = "schedule %r preferred for this camera count, %r already in force in this process: kept" % (schedule, get_schedule())
                import logging
                logging.getLogger(__name__).info(self.schedule_note)
            schedule = None
        self.numa = None
        self.arena_nodes = None
        numa = options.get("numa", "auto")
        if numa is True or (numa == "auto" and self._several_gpus()):
            from ..numa import pin_to_gpu_node
            self.numa = pin_to_gpu_node(device)
        max_batch = max_batch or options.get("max_batch") or int(os.environ.get("WATSOR_HIP_MAX_BATCH", "8"))
        max_width = max_width or options.get("max_width") or int(os.environ.get("WATSOR_HIP_MAX_WIDTH", "1920"))
        max_height = max_height or options.get("max_height") or int(os.environ.get("WATSOR_HIP_MAX_HEIGHT", "1080"))
        pf = options.get("pixel_format") or "rgb24"
        self.__fmt_by_name = {str(k): pixel_format_code(v) for k, v in pf.items()} if isinstance(pf, dict) else {}
        self.__fmt_default = FMT_RGB24 if isinstance(pf, dict) else pixel_format_code(pf)
        self.__fmt_by_cam = {}
        self.__engine = HipEngine(engine_path, device, max_batch, max_width, max_height, schedule=schedule)
        self.__device = device
        self.__filters = []
        self.__pinned = []

    @staticmethod
    def _several_gpus() -> bool:
        from ..runtime import device_count
        try:
            return device_count() > 1
        except (OSError, RuntimeError):
            return False

    @property
    def schedule(self) -> str:
        return self.__engine.schedule

    def _formats(self, frames: Sequence[np.ndarray], cameras: Optional[Sequence[int]]):
        return frame_formats(frames, cameras, self.__fmt_default, self.__fmt_by_cam)

    @property
    def engine(self) -> HipEngine:
        return self.__engine

    @property
    def max_batch(self) -> int:
        return self.__engine.max_batch

    @property
    def device_name(self):
        return self.__engine.device_name

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        try:
            self.__engine.sync()
            for addr in self.__pinned:
                try:
                    self.__engine.host_unregister_address(addr)
                except (RuntimeError, ValueError):
                    pass
        finally:
            self.__pinned = []
            self.__engine.close()
            if self.numa:
                from ..numa import restore_affinity
                restore_affinity(self.numa)

    # -- what `BatchedObjectDetector` uses inside the worker process ----------------------------------------------------
    @property
    def num_lanes(self) -> int:
        return self.__engine.num_slots

    def bind_cameras(self, frame_buffers, camera_configs=None, drop: bool = False, logger=None):
        """Called once in the worker process.  Gives an id to every camera (key of `frame_buffers`) that needs one -- a
        configured GPU filter or a pixel format of its own; the others are -1, "no camera", so that any number of plain
        cameras fits beside the engine's 256 filter slots -- registers the GPU filters of the cameras whose (normalised)
        configuration is given -- `HipCameraFilter`, i.e. the reference's ConfidenceFilter / AreaFilter / MaskFilter
        constructors (`watsor/filter/{confidence,area,mask}.py`) -- and page-locks every `Frame.image` array
        (`watsor/stream/share.py:35-41`) so that frames travel without a host-side copy.  Returns {camera name: id}."""
        from ..filter.hip_filter import HipCameraFilter
        from .._lib import WZ_MAX_CAMS
        names = sorted(frame_buffers, key=str)
        need = [n for n in names if n in (camera_configs or {}) or str(n) in self.__fmt_by_name]
        if len(need) > WZ_MAX_CAMS:
            raise ValueError("%d cameras with GPU filters / pixel formats of their own on one detector: the engine has %d slots"
                             % (len(need), WZ_MAX_CAMS))
        ids = {name: -1 for name in names}
        ids.update({name: i for i, name in enumerate(need)})
        self.__fmt_by_cam = {i: self.__fmt_by_name.get(str(name), self.__fmt_default) for name, i in ids.items() if i >= 0}
        for name, cfg in (camera_configs or {}).items():
            if name in ids:
                self.__filters.append(HipCameraFilter(self.__engine, ids[name], cfg, drop=drop))
        import ctypes
        arenas = {}
        for cam_name, fb in frame_buffers.items():
            for frame in fb.frames:
                obj = frame.image.get_obj() if hasattr(frame.image, "get_obj") else frame.image
                addr, size = ctypes.addressof(obj), ctypes.sizeof(obj)
                arenas.setdefault(cam_name, addr)
                try:
                    self.__engine.host_register_address(addr, size)
                    self.__pinned.append(addr)
                except (RuntimeError, ValueError) as e:      # still correct, just not DMA speed
                    if logger is not None:
                        logger.warning("frame memory at 0x%x could not be page-locked: %s" % (addr, e))
        if self.numa is not None:      # (multi-GPU hosts: where do the frames this detector will DMA from actually live?)
            from ..numa import report_arena_nodes
            self.arena_nodes = report_arena_nodes(arenas, self.numa.get("numa_node", -1), logger)
        return ids

    def bind_frame_table(self, frame_buffers, ids):
        """Called once in the worker process, after `bind_cameras`: describes every Frame of every FrameBuffer to the engine
        (`wz_bind_frames`: pixels, size, pixel format, camera id, the address of `header.detections`) -- what the reference
        worker looks up and rebuilds per payload (`watsor/detection/detector.py:104-106`, `share.py:68-73`) never changes
        after the buffers exist.  Returns {camera name: (index of its frame 0 in the table, [frame.latch.next, ...])}."""
        import ctypes
        pix, ws, hs, fmts, cams, rows, table = [], [], [], [], [], [], {}
        yuv = next((f for f in [self.__fmt_default] + list(self.__fmt_by_cam.values()) if f != FMT_RGB24), FMT_NV12)
        for name in sorted(frame_buffers, key=str):
            cam = ids.get(name, -1)
            fmt0 = self.__fmt_by_cam.get(cam, self.__fmt_default) if cam >= 0 else self.__fmt_default
            latches = []
            table[name] = (len(pix), latches)
            for frame in frame_buffers[name].frames:
                hdr = frame.header.get_obj() if hasattr(frame.header, "get_obj") else frame.header
                obj = frame.image.get_obj() if hasattr(frame.image, "get_obj") else frame.image
                w, h, fmt = int(hdr.width), int(hdr.height), fmt0
                if fmt == FMT_RGB24 and int(hdr.channels) == 1:      # a planar buffer read as RGB24: the configured YUV format
                    fmt = yuv
                if fmt != FMT_RGB24:                                 # (H * 3 / 2, W) bytes of a W x H picture
                    if int(hdr.channels) != 1 or h % 3 or w % 2 or (h // 3 * 2) % 2:
                        raise ValueError("camera %r: an NV12 / I420 frame buffer must be (H*3/2, W, 1) with even H and W" % (name,))
                    h = h // 3 * 2
                elif int(hdr.channels) != 3:
                    raise ValueError("camera %r: an RGB24 frame buffer must have 3 channels" % (name,))
                if ctypes.sizeof(obj) < (w * h * 3 if fmt == FMT_RGB24 else w * h * 3 // 2):
                    raise ValueError("camera %r: frame memory smaller than its header says" % (name,))
                pix.append(ctypes.addressof(obj))
                ws.append(w)
                hs.append(h)
                fmts.append(fmt)
                cams.append(cam)
                rows.append(ctypes.addressof(hdr.detections))
                latches.append(frame.latch.next)
        self.__engine.bind_frames(pix, ws, hs, fmts, cams, rows)
        self.submit_bound = self.__engine.submit_bound      # (the worker calls these once per batch: no wrapper frames in between)
        self.collect_bound = self.__engine.collect_bound
        return table

    def submit_host(self, lane: int, images: Sequence[np.ndarray], cameras: Optional[Sequence[int]] = None) -> None:
        """Asynchronous `detect_batch`: the frames (views of shared memory, unchanged until `collect`) are enqueued on `lane`."""
        self.__engine.submit_host(lane, images, cameras, self._formats(images, cameras))

    def collect(self, lane: int, detections: Sequence) -> None:
        """Waits for `lane` and writes its rows into the given `Detection[100]` arrays (the frame headers)."""
        self.__engine.collect(lane, detections)

    def detect(self, image_shape, image_np, detections: List[Detection]):
        frames = [image_np.reshape(image_shape)]
        return self.__engine.detect_batch(frames, [detections], formats=self._formats(frames, None))

    def detect_batch(self, image_shapes: Sequence, images: Sequence[np.ndarray], detections: Sequence,
                     cameras: Optional[Sequence[int]] = None, passes: Optional[Sequence[np.ndarray]] = None):
        frames = [im.reshape(sh) for sh, im in zip(image_shapes, images)]
        return self.__engine.detect_batch(frames, detections, cameras, passes, self._formats(frames, cameras))
