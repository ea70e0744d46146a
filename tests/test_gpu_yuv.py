"""NV12 / I420 frames through the HIP path (the decoder side).  The colour conversion is integer arithmetic: the
resize kernel must see exactly the RGB bytes oracle/yuv.py works out, so everything downstream is bit-identical to the RGB24 path
on those bytes."""
import numpy as np
import pytest

import conftest


def make_engine(*args, **kwargs):
    """Stage-level entry points and WZ_* knobs live in the development library (include/watsor_hip.h, WZ_DEV_BUILD section)."""
    kwargs.setdefault("dev", True)
    return conftest.make_engine(*args, **kwargs)

from oracle import yuv
from watsor_amd.runtime import FMT_I420, FMT_NV12, FMT_RGB24, ROW_DTYPE
from watsor_amd.synth import synthetic_frame

pytestmark = pytest.mark.gpu
FMT = {"nv12": FMT_NV12, "i420": FMT_I420}


@pytest.fixture(scope="module")
def eng(model_dir):
    e = make_engine(model_dir)
    yield e
    e.close()


def _yuv_frame(w, h, seed, fmt):
    """A picture with real chroma detail: the synthetic RGB frame converted, then every byte perturbed (so that out-of-range
    levels, clipping and odd chroma values occur)."""
    f = yuv.yuv420_from_rgb(synthetic_frame(w, h, seed), fmt).astype(np.int16)
    rng = np.random.default_rng(seed)
    f += rng.integers(-24, 25, f.shape, dtype=np.int16)
    f[::37, ::11] = rng.integers(0, 256, f[::37, ::11].shape)
    return np.clip(f, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("fmt", ["nv12", "i420"])
@pytest.mark.parametrize("size", [(640, 480), (1280, 720), (1920, 1080), (300, 300), (322, 242)])
def test_resize_kernel_sees_the_restated_rgb(eng, fmt, size):
    w, h = size
    frame = _yuv_frame(w, h, 21 + w, fmt)
    rgb = yuv.rgb_from_yuv420(frame, w, h, fmt)
    got = eng.stage_preprocess(frame, FMT[fmt])
    want = eng.stage_preprocess(rgb)
    assert got.shape == want.shape
    np.testing.assert_array_equal(got.view(np.uint16), want.view(np.uint16))


def test_rows_of_a_mixed_batch(eng):
    """RGB24, NV12 and I420 frames of different sizes in ONE batch == each of them as the RGB24 frame the oracle converts it to."""
    specs = [("rgb", 640, 480), ("nv12", 640, 480), ("i420", 1280, 720), ("nv12", 1920, 1080), ("rgb", 1280, 720), ("i420", 640, 480)]
    frames, formats, as_rgb = [], [], []
    for i, (fmt, w, h) in enumerate(specs):
        if fmt == "rgb":
            f = synthetic_frame(w, h, 50 + i)
            frames.append(f); formats.append(FMT_RGB24); as_rgb.append(f)
        else:
            f = _yuv_frame(w, h, 50 + i, fmt)
            frames.append(f); formats.append(FMT[fmt]); as_rgb.append(yuv.rgb_from_yuv420(f, w, h, fmt))
    got = [np.zeros(100, ROW_DTYPE) for _ in frames]
    eng.detect_batch(frames, got, formats=formats)
    ref = [np.zeros(100, ROW_DTYPE) for _ in frames]
    eng.detect_batch(as_rgb, ref)
    for a, b in zip(got, ref):
        assert a.tobytes() == b.tobytes()
    assert any((r["confidence"] > 0.3).any() for r in got)
    # the asynchronous host path takes the formats as well
    eng.submit_host(0, frames, formats=formats)
    again = [np.zeros(100, ROW_DTYPE) for _ in frames]
    eng.collect(0, again)
    for a, b in zip(again, ref):
        assert a.tobytes() == b.tobytes()


def test_odd_sides_and_unknown_formats_are_refused(eng):
    rows = [np.zeros(100, ROW_DTYPE)]
    with pytest.raises(ValueError):
        eng.detect_batch([np.zeros((481 * 3 // 2, 640), np.uint8)], rows, formats=[FMT_NV12])
    with pytest.raises(ValueError):
        eng.detect_batch([np.zeros((480, 640, 3), np.uint8)], rows, formats=[9])
    d = eng.upload(np.zeros((720, 641), np.uint8))
    with pytest.raises(ValueError):
        eng.submit_device(0, [d], [641], [480], formats=[FMT_NV12])   # the C ABI checks it again
    eng.sync()
    eng.free(d)
