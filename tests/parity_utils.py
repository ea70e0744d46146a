"""Shared helpers for the GPU parity tests and tools/gpu_diag.py (oracle side + comparison metrics)."""
from __future__ import annotations

import os

import numpy as np

from oracle import postprocess as post
from oracle import preprocess as pre
from oracle import detect as odet


def oracle_input_half(frames, size=300):
    """[n,size,size,4] float16 network input exactly as the engine stores it (4th channel zero)."""
    out = np.zeros((len(frames), size, size, 4), np.float16)
    for i, f in enumerate(frames):
        out[i, :, :, :3] = pre.preprocess_fp16(f, size)
    return out


def oracle_forward_from_half(oracle_net, x_half, keep=False):
    """Run the fp32 oracle on the *fp16-rounded* input (isolates network error from input rounding)."""
    x = x_half[..., :3].astype(np.float32)
    return oracle_net.forward(x, keep=keep)


_ANCHORS = None


def anchors_cs():
    global _ANCHORS
    if _ANCHORS is None:
        _ANCHORS = post.anchors_center_size(post.generate_anchors())
    return _ANCHORS


def oracle_postprocess(box_enc, logits, **kw):
    """[n,...] -> boxes [n,100,4], scores [n,100], classes int32 [n,100] (1-based), num [n]."""
    n = box_enc.shape[0]
    B = np.zeros((n, 100, 4), np.float32); S = np.zeros((n, 100), np.float32)
    Cc = np.zeros((n, 100), np.int32); N = np.zeros((n,), np.int32)
    for i in range(n):
        b, s, c, nd = post.postprocess(box_enc[i], logits[i], anchors_cs(), **kw)
        B[i], S[i], Cc[i], N[i] = b, s, c.astype(np.int32), nd
    return B, S, Cc, N


def rel_err(a, ref):
    a = a.astype(np.float64); ref = ref.astype(np.float64)
    return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-12))


from oracle.compare import box_iou_px, match_rows   # noqa: E402,F401  (shared with bench.py's live parity leg)
