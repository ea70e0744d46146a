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


def box_iou_px(a, b):
    """IoU of inclusive pixel boxes (x_min,y_min,x_max,y_max)."""
    ix0, iy0 = max(a[0], b[0]), max(a[1], b[1])
    ix1, iy1 = min(a[2], b[2]), min(a[3], b[3])
    iw, ih = max(ix1 - ix0 + 1, 0), max(iy1 - iy0 + 1, 0)
    inter = iw * ih
    ua = (a[2] - a[0] + 1) * (a[3] - a[1] + 1) + (b[2] - b[0] + 1) * (b[3] - b[1] + 1) - inter
    return inter / ua if ua > 0 else 0.0


def match_rows(gpu_rows, ref, min_score=0.0):
    """Greedy one-to-one matching of detections by (label, IoU).  gpu_rows: ROW_DTYPE[100];
    ref: dict from oracle.detect.rows_as_array.  Returns list of (ref_idx, gpu_idx, iou, dscore)
    and the list of unmatched reference indices (with confidence > min_score)."""
    used = set()
    pairs, missing = [], []
    for i in range(len(ref["label"])):
        if ref["confidence"][i] <= min_score:
            continue
        best, best_iou = -1, 0.0
        rb = ref["box"][i]
        for j in range(len(gpu_rows)):
            if j in used or gpu_rows["label"][j] != ref["label"][i] or gpu_rows["confidence"][j] <= 0:
                continue
            gb = (gpu_rows["x_min"][j], gpu_rows["y_min"][j], gpu_rows["x_max"][j], gpu_rows["y_max"][j])
            v = box_iou_px(rb, gb)
            if v > best_iou:
                best, best_iou = j, v
        if best >= 0 and best_iou >= 0.9:
            used.add(best)
            pairs.append((i, best, best_iou, float(gpu_rows["confidence"][best] - ref["confidence"][i])))
        else:
            missing.append(i)
    return pairs, missing
