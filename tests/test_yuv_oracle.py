"""The YUV 4:2:0 -> RGB24 restatement (oracle/yuv.py) against the known answers of 8-bit BT.601 video levels, and the
format plumbing of the host API (no GPU)."""
import numpy as np
import pytest

from oracle import yuv
from watsor_amd import _lib
from watsor_amd.runtime import FMT_I420, FMT_NV12, FMT_RGB24, HipEngine


def _flat(y, u, v, w=4, h=2, fmt="nv12"):
    yp = np.full((h, w), y, np.uint8)
    if fmt == "nv12":
        c = np.tile(np.array([u, v], np.uint8), (h // 2, w // 2)).reshape(h // 2, w)
    else:
        c = np.concatenate([np.full((h // 2) * (w // 2), u, np.uint8), np.full((h // 2) * (w // 2), v, np.uint8)]).reshape(h // 2, w)
    return np.concatenate([yp, c], axis=0)


@pytest.mark.parametrize("fmt", ["nv12", "i420"])
@pytest.mark.parametrize("yuv_in, rgb_out", [
    ((16, 128, 128), (0, 0, 0)),          # video black
    ((235, 128, 128), (255, 255, 255)),   # video white
    ((126, 128, 128), (128, 128, 128)),   # mid grey: 298 * 110 + 128 >> 8
    ((0, 128, 128), (0, 0, 0)),           # below black: clipped
    ((255, 128, 128), (255, 255, 255)),   # above white: clipped
    ((81, 90, 240), (255, 0, 0)),         # 100 % red   (Rec. 601 primaries at video levels)
    ((145, 54, 34), (0, 255, 0)),         # 100 % green
    ((41, 240, 110), (0, 0, 255)),        # 100 % blue
    ((162, 44, 142), (191, 191, 0)),      # 75 % yellow bar
])
def test_known_answers(fmt, yuv_in, rgb_out):
    got = yuv.rgb_from_yuv420(_flat(*yuv_in, fmt=fmt), 4, 2, fmt)
    assert got.shape == (2, 4, 3)
    assert np.abs(got.astype(int) - np.array(rgb_out)).max() <= 1, got[0, 0]   # the published 8.8 coefficients are rounded: +-1
    assert (got == got[0, 0]).all()


def test_chroma_is_taken_from_the_2x2_block():
    w, h = 8, 4
    rng = np.random.default_rng(3)
    frame = rng.integers(0, 256, (h * 3 // 2, w), dtype=np.uint8)
    a = yuv.rgb_from_yuv420(frame, w, h, "nv12")
    # the same picture as I420: de-interleave the chroma rows
    uv = frame[h:].reshape(h // 2, w // 2, 2)
    i420 = np.concatenate([frame[:h].reshape(-1), uv[..., 0].reshape(-1), uv[..., 1].reshape(-1)]).reshape(h * 3 // 2, w)
    b = yuv.rgb_from_yuv420(i420, w, h, "i420")
    np.testing.assert_array_equal(a, b)
    # luma-only change inside a block leaves the block's chroma contribution alone: R - 298/256 * (Y - 16) is constant per block
    y = frame[:h].astype(int)
    e = uv[..., 1].astype(int) - 128
    want_r = np.clip((298 * (y - 16) + 409 * np.repeat(np.repeat(e, 2, 0), 2, 1) + 128) >> 8, 0, 255)
    np.testing.assert_array_equal(a[..., 0], want_r)


def test_round_trip_stays_close():
    rng = np.random.default_rng(5)
    smooth = np.repeat(np.repeat(rng.integers(0, 256, (8, 10, 3)), 2, 0), 2, 1).astype(np.uint8)   # constant 2x2 blocks
    for fmt in ("nv12", "i420"):
        back = yuv.rgb_from_yuv420(yuv.yuv420_from_rgb(smooth, fmt), 20, 16, fmt)
        assert np.abs(back.astype(int) - smooth.astype(int)).max() <= 3


def test_frame_geometry_and_sizes():
    g = HipEngine.frame_geometry
    assert g(np.zeros((480, 640, 3), np.uint8)) == (640, 480)
    assert g(np.zeros((720, 640), np.uint8), FMT_NV12) == (640, 480)
    assert g(np.zeros((1620, 1920), np.uint8), FMT_I420) == (1920, 1080)
    assert g(np.zeros((720, 640, 1), np.uint8), FMT_NV12) == (640, 480)        # a one-channel FrameBuffer of the reference
    for bad, fmt in ((np.zeros((480, 640), np.uint8), FMT_RGB24), (np.zeros((480, 640, 3), np.uint8), FMT_NV12),
                     (np.zeros((721, 640), np.uint8), FMT_NV12), (np.zeros((720, 641), np.uint8), FMT_I420),
                     (np.zeros((480, 640, 3), np.float32), FMT_RGB24)):
        with pytest.raises(ValueError):
            g(bad, fmt)
    lib = _lib.load()
    assert lib.wz_frame_bytes(640, 480, FMT_RGB24) == 640 * 480 * 3
    assert lib.wz_frame_bytes(640, 480, FMT_NV12) == lib.wz_frame_bytes(640, 480, FMT_I420) == 640 * 480 * 3 // 2
    assert lib.wz_frame_bytes(641, 480, FMT_NV12) == 0 and lib.wz_frame_bytes(640, 480, 7) == 0 and lib.wz_frame_bytes(0, 4, 0) == 0


def test_plugin_picks_the_format_per_camera():
    from watsor_amd.detection.hip_gpu import frame_formats, pixel_format_code
    rgb, planar = np.zeros((4, 4, 3), np.uint8), np.zeros((6, 4), np.uint8)
    assert pixel_format_code("NV12") == FMT_NV12 and pixel_format_code("yuv420p") == FMT_I420 and pixel_format_code("rgb24") == FMT_RGB24
    with pytest.raises(ValueError):
        pixel_format_code("bgr24")
    assert frame_formats([rgb, rgb], None, FMT_RGB24, {}) is None                                  # nothing to say: the plain call
    assert frame_formats([rgb, planar], None, FMT_RGB24, {}) == [FMT_RGB24, FMT_NV12]               # a planar array cannot be RGB24
    assert frame_formats([planar], None, FMT_I420, {}) == [FMT_I420]                               # the detector's format
    by_cam = {0: FMT_RGB24, 1: FMT_I420, 2: FMT_NV12}
    assert frame_formats([rgb, planar, planar], [0, 1, 2], FMT_RGB24, by_cam) == [FMT_RGB24, FMT_I420, FMT_NV12]
    assert frame_formats([planar], [0], FMT_RGB24, by_cam) == [FMT_I420]                           # 2-D under an RGB camera: the configured YUV format
    assert frame_formats([planar[:, :, None]], None, FMT_RGB24, {}) == [FMT_NV12]                   # (H*3/2, W, 1) is planar, too
    assert frame_formats([rgb], [7], FMT_RGB24, by_cam) is None                                    # unknown camera id: the default
