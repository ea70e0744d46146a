"""Oracle (test infrastructure): resize + normalise exactly as the TF graph does it.

The reference's CPU plugin feeds the *full resolution* uint8 frame (`tensorflow_cpu.py:113-115`)
and the graph resizes it: `ToFloat` -> `ResizeBilinear(300x300, align_corners=False)` (TF1 legacy,
no half-pixel centres) -> `(2/255) * x - 1` (the TRT plugin
restates the normalisation in Python at `tensorrt_gpu.py:179-180`).  PARITY UNPINNED vs real TF.

Everything is float32 with one rounding per operation, in the operation order of TF's
`resize_bilinear_op.cc` (`compute_interpolation_weights` / `compute_lerp`), so that a GPU kernel
written with un-contracted fp32 ops can match it bit for bit.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def interpolation_weights(out_size: int, in_size: int, half_pixel_centers: bool = False):
    """lower, upper (int64) and lerp (float32) per output index.  half_pixel_centers: the scaler of graphs exported with
    `ResizeBilinear(half_pixel_centers=True)` (TF >= 1.14 / TF2 exporters; `HalfPixelScaler` of image_resizer_state.h:
    (out + 0.5) * scale - 0.5, three roundings), instead of the legacy out * scale of the 2018 graph the README names."""
    scale = F32(in_size) / F32(out_size)                 # CalculateResizeScale, align_corners=False
    i = np.arange(out_size, dtype=np.float32)
    if half_pixel_centers:
        src = (((i + F32(0.5)).astype(np.float32) * scale).astype(np.float32) - F32(0.5)).astype(np.float32)
    else:
        src = (i * scale).astype(np.float32)             # legacy scaler: out * scale
    lo_f = np.floor(src)
    lower = np.maximum(lo_f.astype(np.int64), 0)
    upper = np.minimum(np.ceil(src).astype(np.int64), in_size - 1)
    lerp = (src - lo_f).astype(np.float32)
    return lower, upper, lerp


def resize_bilinear(image_u8: np.ndarray, out_h: int = 300, out_w: int = 300, half_pixel_centers: bool = False) -> np.ndarray:
    """(H,W,3) uint8 -> (out_h,out_w,3) float32 (values 0..255)."""
    h, w, _ = image_u8.shape
    yl, yu, yf = interpolation_weights(out_h, h, half_pixel_centers)
    xl, xu, xf = interpolation_weights(out_w, w, half_pixel_centers)
    img = image_u8.astype(np.float32)
    tl = img[yl][:, xl]
    tr = img[yl][:, xu]
    bl = img[yu][:, xl]
    br = img[yu][:, xu]
    xf_ = xf[None, :, None]
    yf_ = yf[:, None, None]
    top = (tl + ((tr - tl).astype(np.float32) * xf_).astype(np.float32)).astype(np.float32)
    bot = (bl + ((br - bl).astype(np.float32) * xf_).astype(np.float32)).astype(np.float32)
    return (top + ((bot - top).astype(np.float32) * yf_).astype(np.float32)).astype(np.float32)


def normalise(x: np.ndarray) -> np.ndarray:
    """(2/255) * x - 1 in float32 (two roundings)."""
    return ((F32(2.0 / 255.0) * x).astype(np.float32) - F32(1.0)).astype(np.float32)


def preprocess(image_u8: np.ndarray, size: int = 300, half_pixel_centers: bool = False) -> np.ndarray:
    """Full-resolution RGB24 frame -> normalised float32 (size,size,3) network input."""
    return normalise(resize_bilinear(image_u8, size, size, half_pixel_centers))


def preprocess_fp16(image_u8: np.ndarray, size: int = 300, half_pixel_centers: bool = False) -> np.ndarray:
    """What the fp16 engine must store: the float32 result rounded to nearest-even fp16."""
    return preprocess(image_u8, size, half_pixel_centers).astype(np.float16)
