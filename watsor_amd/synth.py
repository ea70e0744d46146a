"""Seeded synthetic SSD-MobileNet-v2 weights (TF variable names, TF layouts, unfolded BatchNorm).

There is no model file anywhere in the reference tree (`watsor/test/model/cpu.pb` is a stripped
blob) and no network to fetch `ssd_mobilenet_v2_coco_2018_03_29` (README.md:450), so benchmarks,
smoke and parity tests run on random-init weights *of that architecture*: same shapes, same
BatchNorm-then-ReLU6 structure, He-scaled so activations stay in the range a trained network
produces, class-head bias at logit(0.01) so that scores are sparse like a real detector's.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

from . import arch


def synthetic_weights(seed: int = 1234, program: "arch.Program | None" = None, class_gain: float = 1.0) -> Dict[str, np.ndarray]:
    """class_gain > 1 widens the class logits (the ClassPredictor weights are scaled after the draw, so every other tensor is
    that of class_gain = 1): 1.3 gives ~800 scores above 0.3 per frame in ~25 classes -- a busy scene for the NMS kernel --
    against ~170 in 4 classes (bench.py's `busy_scene` leg)."""
    prog = program or arch.build(fuse=False)   # one op per layer: the draw order of the RNG is part of the seed
    rng = np.random.Generator(np.random.PCG64(seed))
    W: Dict[str, np.ndarray] = {}

    def normal(shape, std):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    for op in prog.ops:
        if op.out_mode == arch.OUT_HEAD:
            fan_in = op.k * op.k * op.cin
            for sub, cols, gain, bias_mean, bias_std in (("BoxEncodingPredictor", op.n_box, 0.4, 0.0, 0.05),
                                                         ("ClassPredictor", op.cout - op.n_box, 0.45, -4.6, 0.3)):
                W["%s/%s/weights" % (op.scope, sub)] = normal((op.k, op.k, op.cin, cols), gain / math.sqrt(fan_in))
                W["%s/%s/biases" % (op.scope, sub)] = (normal((cols,), bias_std) + np.float32(bias_mean)).astype(np.float32)
            continue
        if op.kind == arch.OP_DW:
            fan_in = op.k * op.k
            W[op.scope + "/depthwise_weights"] = normal((op.k, op.k, op.cin, 1), math.sqrt(2.0 / fan_in))
        else:
            fan_in = op.k * op.k * op.cin
            if op.out_mode == arch.OUT_CLS:
                std = 0.45 / math.sqrt(fan_in)
            elif op.out_mode == arch.OUT_BOX:
                std = 0.4 / math.sqrt(fan_in)
            elif op.act == arch.ACT_RELU6:
                std = math.sqrt(2.0 / fan_in)
            else:
                std = math.sqrt(1.0 / fan_in)
            W[op.scope + "/weights"] = normal((op.k, op.k, op.cin, op.cout), std)
        if op.has_bn:
            c = op.cout
            W[op.scope + "/BatchNorm/gamma"] = (1.0 + 0.1 * rng.standard_normal(c, dtype=np.float32)).astype(np.float32)
            W[op.scope + "/BatchNorm/beta"] = normal((c,), 0.1) + np.float32(0.2 if op.act == arch.ACT_RELU6 else 0.0)
            W[op.scope + "/BatchNorm/moving_mean"] = normal((c,), 0.1)
            W[op.scope + "/BatchNorm/moving_variance"] = (
                0.75 + 0.5 * rng.random(c, dtype=np.float32)).astype(np.float32)
        else:
            if op.out_mode == arch.OUT_CLS:
                b = normal((op.cout,), 0.3) + np.float32(-4.6)
            else:
                b = normal((op.cout,), 0.05)
            W[op.scope + "/biases"] = b.astype(np.float32)
    if class_gain != 1.0:
        for k in W:
            if k.endswith("ClassPredictor/weights"):
                W[k] = (W[k] * np.float32(class_gain)).astype(np.float32)
    return W


def spread_channel_scales(W: Dict[str, np.ndarray], decades: float = 2.5, seed: int = 77) -> Dict[str, np.ndarray]:
    """A copy of `W` (MobileNet-v2 backbone variables) whose per-channel dynamic ranges are spread over `decades` decades, the
    way folding a TRAINED network's BatchNorm spreads them -- He-initialised weights keep every channel of a tensor at the same
    scale, which flatters fixed-point and fp16 storage.  Three families of per-channel factors in
    (10^-decades, 1], drawn log-uniformly:
      * the expanded tensor (expand conv's / the stem's BatchNorm gamma and beta times a_c, the depthwise filter of channel c
        divided by a_c -- exactly what a BatchNorm behind the depthwise conv does to a small-amplitude input channel);
      * the depthwise output (its BatchNorm gamma and beta times a'_c, row c of the project weights divided by a'_c);
      * the bottleneck tensors (project BatchNorm gamma and beta times s_c, row c of the consumers' weights divided by s_c; one
        s per residual stage, so the skip connections still add like with like).
    Apart from where ReLU6 clips, the network computes the same function; it is simply another random network."""
    rng = np.random.Generator(np.random.PCG64(seed))
    W = {k: v.copy() for k, v in W.items()}
    fe = arch.FE

    def factors(c):
        return np.power(10.0, -decades * rng.random(c)).astype(np.float32)

    def scale_bn(scope, f):
        W[scope + "/BatchNorm/gamma"] = W[scope + "/BatchNorm/gamma"] * f
        W[scope + "/BatchNorm/beta"] = W[scope + "/BatchNorm/beta"] * f

    idx = 0
    prev_scale = None           # factors of the tensor the next expand conv reads
    for t, c, n, stride in arch._INVERTED_RESIDUAL:
        s_stage = factors(c)
        for j in range(n):
            name = fe + arch._blk(idx)
            mid = W[name + "/depthwise/depthwise_weights"].shape[2]
            if t != 1:
                if prev_scale is not None:
                    W[name + "/expand/weights"] = W[name + "/expand/weights"] / prev_scale[None, None, :, None]
                a = factors(mid)
                scale_bn(name + "/expand", a)
            else:                       # block 0: the depthwise conv reads the stem's output
                a = factors(mid)
                scale_bn(fe + "Conv", a)
            W[name + "/depthwise/depthwise_weights"] = W[name + "/depthwise/depthwise_weights"] / a[None, None, :, None]
            a2 = factors(mid)
            scale_bn(name + "/depthwise", a2)
            W[name + "/project/weights"] = W[name + "/project/weights"] / a2[None, None, :, None]
            scale_bn(name + "/project", s_stage)
            prev_scale = s_stage
            idx += 1
    W[fe + "Conv_1/weights"] = W[fe + "Conv_1/weights"] / prev_scale[None, None, :, None]
    return {k: v.astype(np.float32) for k, v in W.items()}


def synthetic_frame(width: int, height: int, seed: int, n_shapes: int = 4) -> np.ndarray:
    """One packed RGB24 frame (H,W,3 uint8): low-frequency noise + a few filled shapes.

    Mirrors the spirit of the reference's `Artist` test source (`watsor/test/detect_stream.py:43-70`:
    4 random shapes per frame) while giving the bilinear taps non-trivial content.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    gh, gw = max(2, height // 40), max(2, width // 40)
    coarse = rng.random((gh, gw, 3), dtype=np.float32)
    ys = np.linspace(0, gh - 1, height, dtype=np.float32)
    xs = np.linspace(0, gw - 1, width, dtype=np.float32)
    y0 = np.floor(ys).astype(np.int64); y1 = np.minimum(y0 + 1, gh - 1); fy = (ys - y0)[:, None, None]
    x0 = np.floor(xs).astype(np.int64); x1 = np.minimum(x0 + 1, gw - 1); fx = (xs - x0)[None, :, None]
    top = coarse[y0][:, x0] * (1 - fx) + coarse[y0][:, x1] * fx
    bot = coarse[y1][:, x0] * (1 - fx) + coarse[y1][:, x1] * fx
    img = (top * (1 - fy) + bot * fy) * 200.0 + 20.0
    img += rng.standard_normal((height, width, 3), dtype=np.float32) * 6.0
    yy, xx = np.mgrid[0:height, 0:width]
    for _ in range(n_shapes):
        cx, cy = rng.integers(0, width), rng.integers(0, height)
        rw, rh = rng.integers(width // 16, width // 4), rng.integers(height // 16, height // 4)
        color = rng.integers(0, 256, 3).astype(np.float32)
        if rng.random() < 0.5:
            m = (np.abs(xx - cx) <= rw) & (np.abs(yy - cy) <= rh)
        else:
            m = ((xx - cx) / float(rw)) ** 2 + ((yy - cy) / float(rh)) ** 2 <= 1.0
        img[m] = color
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))   # packed RGB24, row-major


def synthetic_zone_mask(width: int, height: int, seed: int, n_blobs: int) -> np.ndarray:
    """Alpha plane (H,W uint8) of a camera's zone mask in the spirit of BASELINE configs[3] and the reference's
    `config/porch.png`: a few filled blobs (>= 8 px thick, alpha 255 = zone, `watsor/filter/mask.py:77-88`) on a
    translucent background (alpha 204), with an anti-aliased rim (alpha 230) that is NOT part of a zone."""
    rng = np.random.default_rng(seed)
    alpha = np.full((height, width), 204, np.uint8)
    yy, xx = np.mgrid[0:height, 0:width]
    for _ in range(n_blobs):
        cx, cy = int(rng.integers(width // 8, 7 * width // 8)), int(rng.integers(height // 8, 7 * height // 8))
        rx, ry = int(rng.integers(max(8, width // 24), width // 7)), int(rng.integers(max(8, height // 24), height // 7))
        if rng.random() < 0.5:
            inner = (np.abs(xx - cx) <= rx) & (np.abs(yy - cy) <= ry)
            rim = (np.abs(xx - cx) <= rx + 2) & (np.abs(yy - cy) <= ry + 2)
        else:
            d = ((xx - cx) / float(rx)) ** 2 + ((yy - cy) / float(ry)) ** 2
            inner, rim = d <= 1.0, d <= 1.08
        alpha[rim & (alpha != 255)] = 230
        alpha[inner] = 255
    return alpha
