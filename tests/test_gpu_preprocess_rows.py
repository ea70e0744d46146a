"""The row-staged resize kernel (`wz_k_preprocess_rows`, k_preprocess.hip) and the in-place read of page-locked host frames
(`WZ_HOST_READ`, wz_engine.hip) against the oracle and against the per-pixel kernel: same fp32 operations in the same order, so the
network input is bit-identical for every format, size and alignment; and a batch whose frames are read in place over PCIe gives
bit-identical ROWS to the same batch staged by DMA.  (SURVEY 8(d) "Host/PCIe side bound".)"""
import os

import numpy as np
import pytest

import conftest
from oracle import preprocess as pre
from oracle import yuv
from watsor_amd.runtime import FMT_I420, FMT_NV12, FMT_RGB24, ROW_DTYPE
from watsor_amd.synth import synthetic_frame

pytestmark = pytest.mark.gpu


def engine_with(model_dir, env, **kw):
    saved = {k: os.environ.get(k) for k in ("WZ_PRE_ROWS", "WZ_HOST_READ")}
    for k in saved:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        return conftest.make_engine(model_dir, dev=True, **kw)         # (the knobs are read when the engine is created)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


@pytest.fixture(scope="module")
def eng_rows(model_dir):
    e = engine_with(model_dir, {"WZ_PRE_ROWS": "1"})
    yield e
    e.close()


@pytest.mark.parametrize("wh", [(640, 480), (1280, 720), (1920, 1080), (300, 300), (301, 299), (64, 48), (643, 481), (2, 2)])
def test_row_kernel_bit_exact_against_the_oracle(eng_rows, wh):
    f = synthetic_frame(wh[0], wh[1], 7 + wh[0]) if wh[0] > 8 else np.arange(12, dtype=np.uint8).reshape(2, 2, 3) * 20
    got = eng_rows.stage_preprocess(f)
    ref = pre.preprocess_fp16(f)
    np.testing.assert_array_equal(got[..., :3].view(np.uint16), ref.view(np.uint16))
    lo = (pre.preprocess(f) - ref.astype(np.float32)).astype(np.float16)
    np.testing.assert_array_equal(got[..., 4:7].view(np.uint16), lo.view(np.uint16))
    assert not got[..., 3].any() and not got[..., 7].any()


@pytest.mark.parametrize("fmt", ["nv12", "i420"])
@pytest.mark.parametrize("size", [(640, 480), (1920, 1080), (322, 242), (300, 300)])
def test_row_kernel_yuv_formats(eng_rows, fmt, size):
    w, h = size
    frame = yuv.yuv420_from_rgb(synthetic_frame(w, h, 21 + w), fmt)
    frame = (frame.astype(np.int16) + np.random.default_rng(w).integers(-24, 25, frame.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)
    rgb = yuv.rgb_from_yuv420(frame, w, h, fmt)
    got = eng_rows.stage_preprocess(frame, {"nv12": FMT_NV12, "i420": FMT_I420}[fmt])
    want = eng_rows.stage_preprocess(rgb)
    np.testing.assert_array_equal(got.view(np.uint16), want.view(np.uint16))
    np.testing.assert_array_equal(want[..., :3].view(np.uint16), pre.preprocess_fp16(rgb).view(np.uint16))


@pytest.mark.parametrize("mode", ["1", "2"])
def test_frames_read_in_place_give_the_same_rows_as_staged_frames(model_dir, mode):
    """Mixed batch (640x480 / 1920x1080 RGB24, NV12) from page-locked memory at odd addresses: rows of the in-place path ==
    rows of the staged path, bit for bit; also straight after the memory was registered behind a bound frame table."""
    staged = engine_with(model_dir, {"WZ_PRE_ROWS": "0", "WZ_HOST_READ": "0"}, max_batch=8)
    specs = [(640, 480, FMT_RGB24), (1920, 1080, FMT_RGB24), (1920, 1080, FMT_NV12), (1280, 720, FMT_RGB24), (640, 480, FMT_NV12),
             (1920, 1080, FMT_RGB24)]
    arena = np.zeros(sum(w * h * 3 for w, h, _ in specs) + 64 * len(specs) + 7, np.uint8)
    frames, off = [], 5                                          # (frame starts at every residue mod 16 the allocator might give)
    for i, (w, h, fmt) in enumerate(specs):
        if fmt == FMT_RGB24:
            src = synthetic_frame(w, h, 300 + i)
        else:
            src = yuv.yuv420_from_rgb(synthetic_frame(w, h, 300 + i), "nv12")
        view = arena[off:off + src.size].reshape(src.shape)
        view[...] = src
        frames.append(view)
        off += src.size + 13 + 2 * i
    fmts = [s[2] for s in specs]
    try:
        want = [np.zeros(100, ROW_DTYPE) for _ in frames]
        staged.detect_batch(frames, want, formats=fmts)
    finally:
        staged.close()
    direct = engine_with(model_dir, {"WZ_HOST_READ": mode}, max_batch=8)
    try:
        direct.host_register(arena)
        try:
            for _ in range(2):
                direct.submit_host(1, frames, formats=fmts)
                direct.wait(1)
                got = direct.slot_rows(1, len(frames)).copy()
                for i in range(len(frames)):
                    assert got[i].tobytes() == want[i].tobytes(), i
                    assert got[i]["confidence"][0] > 0
        finally:
            direct.sync()
            direct.host_unregister(arena)
        # ... and unregistered (pageable) memory still goes through the staging copy
        direct.submit_host(0, frames, formats=fmts)
        direct.wait(0)
        got = direct.slot_rows(0, len(frames)).copy()
        assert all(got[i].tobytes() == want[i].tobytes() for i in range(len(frames)))
    finally:
        direct.close()
