"""Oracle (test infrastructure): row-level comparison of a detector's output with the oracle's.

Used by the GPU parity tests (tests/parity_utils.py), `__graft_entry__.smoke()` and the live parity leg of
`bench.py` -- as the checker, never on the product path.
"""
from __future__ import annotations


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
