"""The worker's frame table on the GPU (`wz_bind_frames` / `wz_submit_bound` / `wz_collect_bound`, include/watsor_hip.h): frames
described once, batches as table indices -- rows bit-identical to the synchronous `detect_batch` of the same frames, whatever the alignment of
the frame memory and whether it is page-locked or not."""
import numpy as np
import pytest

from conftest import make_engine
from watsor_amd.runtime import FMT_NV12, FMT_RGB24, ROW_DTYPE
from watsor_amd.synth import synthetic_frame

pytestmark = pytest.mark.gpu


def arena_with_frames(frames, lead):
    """One byte arena holding the frames back to back, the first one `lead` bytes in (so that frame addresses fall on every
    alignment the multiprocessing heap can produce: it aligns to 8, `watsor/stream/share.py:35-41`)."""
    total = lead + sum(f.size + 8 for f in frames)
    arena = np.zeros(total + 64, np.uint8)
    views, off = [], lead
    for f in frames:
        v = arena[off:off + f.size]
        v[:] = f.reshape(-1)
        views.append(v.reshape(f.shape))
        off += f.size + 8 - (f.size % 8 or 8) + 8
    return arena, views


@pytest.mark.parametrize("lead,register", [(0, True), (8, True), (3, True), (8, False)])
def test_bound_batches_equal_synchronous_calls(model_dir, lead, register):
    e = make_engine(model_dir, max_batch=8, max_width=1280, max_height=720)
    try:
        sizes = [(640, 480), (1280, 720), (300, 300), (641, 479), (640, 480), (1280, 720)]
        frames = [synthetic_frame(w, h, 4000 + i) for i, (w, h) in enumerate(sizes)]
        arena, views = arena_with_frames(frames, lead)
        assert all(v.ctypes.data % 16 != 0 for v in views[:1]) or lead == 0
        if register:
            e.host_register(arena)
        rows = np.zeros((len(frames), 100), ROW_DTYPE)
        e.bind_frames([v.ctypes.data for v in views], [s[0] for s in sizes], [s[1] for s in sizes], [FMT_RGB24] * len(frames),
                      [-1] * len(frames), [rows[i].ctypes.data for i in range(len(frames))])
        batches = [[0, 1, 2], [3, 4, 5, 0], [5], [2, 2, 1]]
        assert e.num_slots >= len(batches)
        for lane, b in enumerate(batches):                 # all four in flight at once
            e.submit_bound(lane, b)
        # collect in submit order and compare with the synchronous call of the same batch (a batch size picks its summation order)
        got = []
        for lane, b in enumerate(batches):
            e.collect_bound(lane)
            got.append(rows[b].copy())
        for b, g in zip(batches, got):
            ref = [np.zeros(100, ROW_DTYPE) for _ in b]
            e.detect_batch([frames[i] for i in b], ref)
            for k, i in enumerate(b):
                assert g[k].tobytes() == ref[k].tobytes(), (b, i)
                assert g[k]["label"][0] >= 1 and g[k]["confidence"][0] > 0
        e.bind_frames([], [], [], [], [], [])
        if register:
            e.host_unregister(arena)
    finally:
        e.close()


def test_bound_frames_take_nv12_and_camera_filters(model_dir):
    from oracle import yuv
    from watsor_amd.filter.hip_filter import HipCameraFilter
    from watsor_amd.coco import COCO_CLASSES
    e = make_engine(model_dir, max_batch=4, max_width=1280, max_height=720)
    try:
        rgb = synthetic_frame(640, 480, 51)
        nv = yuv.yuv420_from_rgb(synthetic_frame(1280, 720, 52), "nv12")
        arena = np.zeros(rgb.size + nv.size, np.uint8)
        arena[:rgb.size] = rgb.reshape(-1)
        arena[rgb.size:] = nv.reshape(-1)
        a, b = arena[:rgb.size].reshape(rgb.shape), arena[rgb.size:].reshape(nv.shape)
        e.host_register(arena)
        cfg = {"width": 640, "height": 480,
               "detect": [{n: {"area": 2, "confidence": 30, "zones": []}} for n in dict.fromkeys(COCO_CLASSES[1:])]}
        flt = HipCameraFilter(e, 3, cfg, drop=True)
        rows = np.zeros((2, 100), ROW_DTYPE)
        e.bind_frames([a.ctypes.data, b.ctypes.data], [640, 1280], [480, 720], [FMT_RGB24, FMT_NV12], [3, -1],
                      [rows[0].ctypes.data, rows[1].ctypes.data])
        e.submit_bound(1, [1, 0])
        e.collect_bound(1)
        ref = [np.zeros(100, ROW_DTYPE) for _ in range(2)]
        e.detect_batch([b, a], ref, cams=[-1, 3], formats=[FMT_NV12, FMT_RGB24])
        assert rows[1].tobytes() == ref[0].tobytes() and rows[0].tobytes() == ref[1].tobytes()
        dropped = rows[0]["label"] == 0
        assert dropped.any() and (rows[0]["confidence"][~dropped] >= 0.3).all()
        flt.close()
        e.sync()
        e.host_unregister(arena)
    finally:
        e.close()


def test_bound_api_errors(model_dir):
    e = make_engine(model_dir, max_batch=2, max_width=640, max_height=480)
    try:
        f = synthetic_frame(640, 480, 1)
        big = synthetic_frame(1280, 720, 2)
        rows = np.zeros((2, 100), ROW_DTYPE)
        with pytest.raises(ValueError):                       # larger than the engine was created for
            e.bind_frames([big.ctypes.data], [1280], [720], [FMT_RGB24], [-1], [rows[0].ctypes.data])
        with pytest.raises(ValueError):                       # odd-sized NV12
            e.bind_frames([f.ctypes.data], [639], [480], [FMT_NV12], [-1], [rows[0].ctypes.data])
        with pytest.raises(ValueError):                       # camera id beyond the filter slots
            e.bind_frames([f.ctypes.data], [640], [480], [FMT_RGB24], [256], [rows[0].ctypes.data])
        e.bind_frames([f.ctypes.data], [640], [480], [FMT_RGB24], [-1], [rows[0].ctypes.data])
        with pytest.raises(ValueError):
            e.submit_bound(0, [1])                            # no such entry
        with pytest.raises(ValueError):
            e.submit_bound(0, [0, 0, 0])                      # more than max_batch
        with pytest.raises(ValueError):
            e.collect_bound(0)                                # nothing bound in flight on that lane
        e.submit_bound(0, [0])
        e.collect_bound(0)
        assert rows[0]["label"][0] >= 1
    finally:
        e.close()


def test_many_plain_cameras_need_no_filter_slot(model_dir):
    """300 cameras on one detector -- ids only go to cameras with a GPU filter or a pixel format of their own."""
    import shm_standins as shm
    from watsor_amd.detection.hip_gpu import HipObjectDetector
    ctx = shm.spawn_context()
    cams = {"cam%03d" % c: shm.FrameBuffer(ctx, 1, 64, 48) for c in range(300)}
    with HipObjectDetector(model_dir, 0, max_batch=2, max_width=64, max_height=48) as det:
        ids = det.bind_cameras(cams, None, False)
        assert len(ids) == 300 and set(ids.values()) == {-1}
        table = det.bind_frame_table(cams, ids)
        assert table["cam299"][0] == 299
        det.submit_bound(0, [299, 0])
        det.collect_bound(0)
        assert cams["cam299"].frames[0].header.get_obj().detections[0].label >= 1
