"""CPU restatement of the colour conversion in front of the detector when a decoder writes NV12 / I420 frames  --  TEST
INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference never sees such frames: its schema makes every decoder write rawvideo RGB24 (`watsor/config/schema.py:161`,
`config/config.yaml:36-40`), i.e. ffmpeg's swscale turns the decoder's YUV 4:2:0 pictures into RGB on the host, and
`watsor/stream/ffmpeg.py:78-88` reads the result into the FrameBuffer.  SURVEY.md 8(f)-3 moves that conversion to the GPU
(half the bytes per frame on PCIe).  What is restated here is the arithmetic the HIP resize kernel applies to such a frame
(`watsor_amd/csrc/k_preprocess.hip: wz_fetch_rgb`), as numpy integer arithmetic:

    8-bit BT.601 limited range ("video levels": Y in 16..235, U/V in 16..240), the chroma sample of a pixel's 2x2 block (no
    chroma interpolation -- swscale's default for its unscaled YUV -> RGB converters), 8.8 fixed point:
        C = Y - 16, D = U - 128, E = V - 128
        R = clip8((298 C + 409 E + 128) >> 8)
        G = clip8((298 C - 100 D - 208 E + 128) >> 8)
        B = clip8((298 C + 516 D + 128) >> 8)

Pinning status: PARITY UNPINNED against ffmpeg (absent here and on the GPU box; swscale's C and SIMD converters use 16.16
tables / 13-bit multiplies and differ from each other -- and from this formula -- by at most one LSB).  The formula itself is
pinned by its known answers (tests/test_yuv_oracle.py): video black / white, the 75 % colour bars, clipping.
"""
import numpy as np


def _planes(buf, w, h, fmt):
    """(Y [h,w], U [h/2,w/2], V [h/2,w/2]) int32 views of a (h*3/2, w) planar frame; fmt 'nv12' | 'i420'."""
    buf = np.asarray(buf, np.uint8).reshape(-1)
    if w % 2 or h % 2 or buf.size != w * h * 3 // 2:
        raise ValueError("NV12 / I420 frames are w*h*3/2 bytes with even w and h")
    y = buf[:w * h].reshape(h, w).astype(np.int32)
    c = buf[w * h:]
    if fmt == "nv12":
        uv = c.reshape(h // 2, w // 2, 2).astype(np.int32)
        return y, uv[..., 0], uv[..., 1]
    if fmt == "i420":
        q = (w // 2) * (h // 2)
        return y, c[:q].reshape(h // 2, w // 2).astype(np.int32), c[q:].reshape(h // 2, w // 2).astype(np.int32)
    raise ValueError(fmt)


def rgb_from_yuv420(buf, w, h, fmt):
    """The (h, w, 3) uint8 RGB24 frame the detector sees for an NV12 / I420 frame."""
    y, u, v = _planes(buf, w, h, fmt)
    u = np.repeat(np.repeat(u, 2, axis=0), 2, axis=1)   # nearest: every pixel of a 2x2 block takes the block's sample
    v = np.repeat(np.repeat(v, 2, axis=0), 2, axis=1)
    c, d, e = y - 16, u - 128, v - 128
    r = (298 * c + 409 * e + 128) >> 8                  # (arithmetic shift on int32: floor, as in the kernel)
    g = (298 * c - 100 * d - 208 * e + 128) >> 8
    b = (298 * c + 516 * d + 128) >> 8
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def yuv420_from_rgb(rgb, fmt):
    """A test-input generator, NOT part of the restated path: an NV12 / I420 frame ((h*3/2, w) uint8) made from an RGB24 picture
    by the forward BT.601 limited-range matrix (rounded), chroma averaged over each 2x2 block."""
    rgb = np.asarray(rgb, np.float64)
    h, w = rgb.shape[:2]
    if w % 2 or h % 2:
        raise ValueError("even sides only")
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    y = 16 + (65.481 * r + 128.553 * g + 24.966 * b) / 255.0
    u = 128 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255.0
    v = 128 + (112.0 * r - 93.786 * g - 18.214 * b) / 255.0
    sub = lambda p: p.reshape(h // 2, 2, w // 2, 2).mean(axis=(1, 3))
    y8 = np.clip(np.rint(y), 0, 255).astype(np.uint8)
    u8 = np.clip(np.rint(sub(u)), 0, 255).astype(np.uint8)
    v8 = np.clip(np.rint(sub(v)), 0, 255).astype(np.uint8)
    if fmt == "nv12":
        chroma = np.stack([u8, v8], axis=-1).reshape(h // 2, w)
    elif fmt == "i420":
        chroma = np.concatenate([u8.reshape(-1), v8.reshape(-1)]).reshape(h // 2, w)
    else:
        raise ValueError(fmt)
    return np.concatenate([y8, chroma], axis=0)
