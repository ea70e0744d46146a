"""Oracle (test infrastructure): SSD anchors, box decode, sigmoid, per-class NMS, top-100.

None of this is reference *code*: it lives inside the TF graph the reference executes
(`tensorflow_cpu.py:94-121`).  Restated from the TF1 Object Detection API as
summarised in SURVEY.md Appendix B.2-B.5, with the hyper-parameters the reference's own SSD
config witnesses (`watsor/test/model/prepare.py`: box coder scales 10/10/5/5 `:54-61`, 6-layer
anchors 0.2..0.95 with ratios 1,2,0.5,3,0.3333 `:108-119`, SIGMOID converter, IoU 0.6,
100/100 detections `:120-128`).  PARITY UNPINNED vs real TF.

Written as the *literal* algorithm (class by class NMS, then concatenate, sort, keep 100) so
that the GPU's single global-order formulation is checked against something independent.
All arithmetic float32, one rounding per operation.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from .ssd_mobilenet_v2 import ANCHORS_PER_LOC, feature_map_sizes

F32 = np.float32

MIN_SCALE, MAX_SCALE = 0.2, 0.95
ASPECT_RATIOS = (1.0, 2.0, 0.5, 3.0, 0.3333)
SCALE_FACTORS = (10.0, 10.0, 5.0, 5.0)        # y, x, h, w
SCORE_THRESHOLD = 1e-8                          # ssd_mobilenet_v2_coco pipeline
IOU_THRESHOLD = 0.6
MAX_PER_CLASS = 100
MAX_TOTAL = 100


def generate_anchors(size: int = 300) -> np.ndarray:
    """[1917,4] float32 corner boxes (ymin,xmin,ymax,xmax), order layer,y,x,anchor (App. B.2)."""
    grids = feature_map_sizes(size)
    n = len(grids)
    scales = [MIN_SCALE + (MAX_SCALE - MIN_SCALE) * i / (n - 1) for i in range(n)] + [1.0]
    out = []
    for layer, g in enumerate(grids):
        s, s_next = scales[layer], scales[layer + 1]
        if layer == 0:
            ls, lr = [0.1, s, s], [1.0, 2.0, 0.5]
        else:
            ls = [s] * len(ASPECT_RATIOS) + [float(np.sqrt(s * s_next))]
            lr = list(ASPECT_RATIOS) + [1.0]
        assert len(ls) == ANCHORS_PER_LOC[layer]
        ls = np.asarray(ls, np.float32)
        ratio_sqrt = np.sqrt(np.asarray(lr, np.float32)).astype(np.float32)
        heights = (ls / ratio_sqrt).astype(np.float32)
        widths = (ls * ratio_sqrt).astype(np.float32)
        stride = F32(1.0 / g)
        offset = F32(0.5 * (1.0 / g))
        cy = (np.arange(g, dtype=np.float32) * stride + offset).astype(np.float32)
        cx = cy.copy()
        # meshgrid order (y, x, anchor)
        yc = np.broadcast_to(cy[:, None, None], (g, g, len(ls)))
        xc = np.broadcast_to(cx[None, :, None], (g, g, len(ls)))
        hh = np.broadcast_to(heights[None, None, :], (g, g, len(ls)))
        ww = np.broadcast_to(widths[None, None, :], (g, g, len(ls)))
        ymin = (yc - F32(0.5) * hh).astype(np.float32)
        xmin = (xc - F32(0.5) * ww).astype(np.float32)
        ymax = (yc + F32(0.5) * hh).astype(np.float32)
        xmax = (xc + F32(0.5) * ww).astype(np.float32)
        out.append(np.stack([ymin, xmin, ymax, xmax], -1).reshape(-1, 4))
    return np.concatenate(out, 0).astype(np.float32)


def anchors_center_size(anchors: np.ndarray) -> np.ndarray:
    """BoxList.get_center_coordinates_and_sizes: [N,4] (ycenter, xcenter, h, w) float32."""
    ymin, xmin, ymax, xmax = (anchors[:, i] for i in range(4))
    w = (xmax - xmin).astype(np.float32)
    h = (ymax - ymin).astype(np.float32)
    yc = (ymin + h / F32(2.0)).astype(np.float32)
    xc = (xmin + w / F32(2.0)).astype(np.float32)
    return np.stack([yc, xc, h, w], 1).astype(np.float32)


def decode_boxes(rel_codes: np.ndarray, anchors_cs: np.ndarray) -> np.ndarray:
    """FasterRcnnBoxCoder._decode (App. B.3): [N,4] encodings -> [N,4] (ymin,xmin,ymax,xmax)."""
    yca, xca, ha, wa = (anchors_cs[:, i] for i in range(4))
    ty = (rel_codes[:, 0] / F32(SCALE_FACTORS[0])).astype(np.float32)
    tx = (rel_codes[:, 1] / F32(SCALE_FACTORS[1])).astype(np.float32)
    th = (rel_codes[:, 2] / F32(SCALE_FACTORS[2])).astype(np.float32)
    tw = (rel_codes[:, 3] / F32(SCALE_FACTORS[3])).astype(np.float32)
    w = (np.exp(tw).astype(np.float32) * wa).astype(np.float32)
    h = (np.exp(th).astype(np.float32) * ha).astype(np.float32)
    yc = ((ty * ha).astype(np.float32) + yca).astype(np.float32)
    xc = ((tx * wa).astype(np.float32) + xca).astype(np.float32)
    hh = (h / F32(2.0)).astype(np.float32)
    hw = (w / F32(2.0)).astype(np.float32)
    return np.stack([yc - hh, xc - hw, yc + hh, xc + hw], 1).astype(np.float32)


def sigmoid(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float32)
    return (F32(1.0) / (F32(1.0) + np.exp(-x).astype(np.float32))).astype(np.float32)


def clip_to_unit_window(boxes: np.ndarray) -> np.ndarray:
    """box_list_ops.clip_to_window with window [0,0,1,1]."""
    return np.minimum(np.maximum(boxes, F32(0.0)), F32(1.0)).astype(np.float32)


def area(boxes: np.ndarray) -> np.ndarray:
    return ((boxes[:, 2] - boxes[:, 0]).astype(np.float32) * (boxes[:, 3] - boxes[:, 1]).astype(np.float32)
            ).astype(np.float32)


def iou(a: np.ndarray, b: np.ndarray) -> np.float32:
    """tensorflow/core/kernels/non_max_suppression_op.cc IOU<float>() for two corner boxes."""
    ymin_i, xmin_i = min(a[0], a[2]), min(a[1], a[3])
    ymax_i, xmax_i = max(a[0], a[2]), max(a[1], a[3])
    ymin_j, xmin_j = min(b[0], b[2]), min(b[1], b[3])
    ymax_j, xmax_j = max(b[0], b[2]), max(b[1], b[3])
    area_i = F32(F32(ymax_i - ymin_i) * F32(xmax_i - xmin_i))
    area_j = F32(F32(ymax_j - ymin_j) * F32(xmax_j - xmin_j))
    if area_i <= 0 or area_j <= 0:
        return F32(0.0)
    iy0, ix0 = max(ymin_i, ymin_j), max(xmin_i, xmin_j)
    iy1, ix1 = min(ymax_i, ymax_j), min(xmax_i, xmax_j)
    inter = F32(max(F32(iy1 - iy0), F32(0.0)) * max(F32(ix1 - ix0), F32(0.0)))
    return F32(inter / F32(F32(area_i + area_j) - inter))


def iou_many(a: np.ndarray, sel: np.ndarray) -> np.ndarray:
    """iou(a, sel[k]) for every row k, same float32 operation order as iou()."""
    ymin_i, xmin_i = np.minimum(a[0], a[2]), np.minimum(a[1], a[3])
    ymax_i, xmax_i = np.maximum(a[0], a[2]), np.maximum(a[1], a[3])
    ymin_j, xmin_j = np.minimum(sel[:, 0], sel[:, 2]), np.minimum(sel[:, 1], sel[:, 3])
    ymax_j, xmax_j = np.maximum(sel[:, 0], sel[:, 2]), np.maximum(sel[:, 1], sel[:, 3])
    area_i = F32(F32(ymax_i - ymin_i) * F32(xmax_i - xmin_i))
    area_j = ((ymax_j - ymin_j).astype(np.float32) * (xmax_j - xmin_j).astype(np.float32)).astype(np.float32)
    iy0, ix0 = np.maximum(ymin_i, ymin_j), np.maximum(xmin_i, xmin_j)
    iy1, ix1 = np.minimum(ymax_i, ymax_j), np.minimum(xmax_i, xmax_j)
    inter = (np.maximum((iy1 - iy0).astype(np.float32), F32(0.0)) *
             np.maximum((ix1 - ix0).astype(np.float32), F32(0.0))).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (inter / ((area_i + area_j).astype(np.float32) - inter).astype(np.float32)).astype(np.float32)
    r[(area_j <= 0) | (area_i <= 0)] = F32(0.0)
    return r


def nms_single_class(boxes: np.ndarray, scores: np.ndarray, max_out: int, iou_thr: float) -> List[int]:
    """tf.image.non_max_suppression: greedy, descending score (ties: lower index first),
    a candidate is dropped iff IoU with an already selected box is > iou_thr."""
    order = np.lexsort((np.arange(len(scores)), -scores.astype(np.float64)))
    thr = F32(iou_thr)
    selected: List[int] = []
    sel_boxes = np.zeros((max_out, 4), np.float32)
    for i in order:
        if len(selected) >= max_out:
            break
        if selected and np.any(iou_many(boxes[i], sel_boxes[:len(selected)]) > thr):
            continue
        sel_boxes[len(selected)] = boxes[i]
        selected.append(int(i))
    return selected


def multiclass_nms(boxes: np.ndarray, scores: np.ndarray,
                   score_thr: float = SCORE_THRESHOLD, iou_thr: float = IOU_THRESHOLD,
                   max_per_class: int = MAX_PER_CLASS, max_total: int = MAX_TOTAL
                   ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, int]:
    """post_processing.multiclass_non_max_suppression (2018 graph order, App. B.5).

    boxes [N,4] decoded (unclipped), scores [N,C] without the background column.
    Per class: keep score > thr, clip to [0,1], drop zero-area, greedy NMS, <= max_per_class;
    concatenate classes in class order, sort by score (stable: ties keep concatenation order),
    keep max_total, pad with zeros.  Returns (boxes[max_total,4], scores, classes(0-based), n).
    """
    n, c = scores.shape
    sel_boxes, sel_scores, sel_classes = [], [], []
    clipped_all = clip_to_unit_window(boxes)
    valid_area = area(clipped_all) > F32(0.0)
    for cls in range(c):
        s = scores[:, cls]
        m = (s > F32(score_thr)) & valid_area
        idx = np.nonzero(m)[0]
        if idx.size == 0:
            continue
        b = clipped_all[idx]
        keep = nms_single_class(b, s[idx], min(max_per_class, idx.size), iou_thr)
        for k in keep:
            sel_boxes.append(b[k])
            sel_scores.append(s[idx[k]])
            sel_classes.append(cls)
    out_b = np.zeros((max_total, 4), np.float32)
    out_s = np.zeros((max_total,), np.float32)
    out_c = np.zeros((max_total,), np.float32)
    if sel_scores:
        order = sorted(range(len(sel_scores)), key=lambda i: (-float(sel_scores[i]), i))[:max_total]
        for r, i in enumerate(order):
            out_b[r], out_s[r], out_c[r] = sel_boxes[i], sel_scores[i], sel_classes[i]
        nd = len(order)
    else:
        nd = 0
    return out_b, out_s, out_c, nd


def multiclass_nms_clip_after(boxes: np.ndarray, scores: np.ndarray,
                              score_thr: float = SCORE_THRESHOLD, iou_thr: float = IOU_THRESHOLD,
                              max_per_class: int = MAX_PER_CLASS, max_total: int = MAX_TOTAL
                              ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, int]:
    """The OTHER order of the same steps (SURVEY App. B.5 "verify"): later exporters of the Object Detection API run the
    per-class NMS on the UNCLIPPED decoded boxes and clip what it selected afterwards
    (`post_processing.multiclass_non_max_suppression`: concatenate -> sort by score -> `_clip_window_prune_boxes` -> first
    max_total).  Per class: keep score > thr (no area test: `tf.image.non_max_suppression` takes degenerate boxes, their IoU
    is 0), greedy NMS on the boxes as decoded, <= max_per_class; concatenate in class order, stable sort by score; clip to
    [0,1]; boxes left without area are pruned (they still suppressed their neighbours and used a slot of their class);
    the first max_total remain.  An engine built with `clip_after_nms` computes this (csrc/k_post.hip), literal twin of
    `multiclass_nms` above."""
    n, c = scores.shape
    sel_boxes, sel_scores, sel_classes = [], [], []
    for cls in range(c):
        s = scores[:, cls]
        idx = np.nonzero(s > F32(score_thr))[0]
        if idx.size == 0:
            continue
        b = boxes[idx]
        keep = nms_single_class(b, s[idx], min(max_per_class, idx.size), iou_thr)
        for k in keep:
            sel_boxes.append(b[k])
            sel_scores.append(s[idx[k]])
            sel_classes.append(cls)
    out_b = np.zeros((max_total, 4), np.float32)
    out_s = np.zeros((max_total,), np.float32)
    out_c = np.zeros((max_total,), np.float32)
    nd = 0
    if sel_scores:
        order = sorted(range(len(sel_scores)), key=lambda i: (-float(sel_scores[i]), i))
        for i in order:
            cb = clip_to_unit_window(np.asarray(sel_boxes[i], np.float32)[None])[0]
            if not area(cb[None])[0] > F32(0.0):
                continue
            out_b[nd], out_s[nd], out_c[nd] = cb, sel_scores[i], sel_classes[i]
            nd += 1
            if nd == max_total:
                break
    return out_b, out_s, out_c, nd


def multiclass_nms_global_order(boxes: np.ndarray, scores: np.ndarray,
                                score_thr: float = SCORE_THRESHOLD, iou_thr: float = IOU_THRESHOLD,
                                max_per_class: int = MAX_PER_CLASS, max_total: int = MAX_TOTAL
                                ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, int]:
    """The same function as `multiclass_nms`, evaluated in ONE global descending-score order (a candidate's
    fate depends only on higher-scored boxes of its own class, and only the first `max_total` survivors reach the output,
    so the walk can stop there).  Used by bench.py's `cpu_baseline` leg so that the baseline is bounded by the network, not
    by 90 Python loops; `tests/test_oracle_postprocess_fast.py` holds it equal to the literal version above, which stays
    the checker of the GPU path.  Ties: (-score, class, anchor) == the literal version's concatenation + stable sort."""
    n, c = scores.shape
    clipped = clip_to_unit_window(boxes)
    valid = area(clipped) > F32(0.0)
    flat = np.where(valid[:, None], scores, F32(-1.0)).T.reshape(-1)          # [class][anchor]: index = class * n + anchor
    cand = np.nonzero(flat > F32(score_thr))[0]
    out_b = np.zeros((max_total, 4), np.float32)
    out_s = np.zeros((max_total,), np.float32)
    out_c = np.zeros((max_total,), np.float32)
    if cand.size == 0:
        return out_b, out_s, out_c, 0
    thr = F32(iou_thr)
    kept_boxes = np.zeros((c, max_per_class, 4), np.float32)
    kept_n = np.zeros(c, np.int64)
    # a class that has filled its quota takes no more candidates -- and its LATER (lower) survivors could never displace
    # anything: the walk is over the candidates in order, in chunks so that the sort is not over all 172 k of them
    def ordered(idx):
        return idx[np.lexsort((idx, -flat[idx].astype(np.float64)))]          # (-score, class * n + anchor)

    # the top of the order first (everything at or above the 4096th score, ties included); the rest only if the walk
    # gets that far -- sorting all 172 k candidates would cost more than the walk
    top_k = 4096
    if cand.size > top_k:
        cut = np.partition(flat[cand], cand.size - top_k)[cand.size - top_k]
        parts = [cand[flat[cand] >= cut], cand[flat[cand] < cut]]
    else:
        parts = [cand]
    nd = 0
    for part in parts:
        for i in ordered(part):
            cls, a = divmod(int(i), n)
            k = int(kept_n[cls])
            if k >= max_per_class:
                continue
            if k and np.any(iou_many(clipped[a], kept_boxes[cls, :k]) > thr):
                continue
            kept_boxes[cls, k] = clipped[a]
            kept_n[cls] = k + 1
            out_b[nd], out_s[nd], out_c[nd] = clipped[a], flat[i], cls
            nd += 1
            if nd == max_total:
                return out_b, out_s, out_c, nd
    return out_b, out_s, out_c, nd


def postprocess(box_enc: np.ndarray, cls_logits: np.ndarray, anchors_cs: np.ndarray, fast: bool = False,
                clip_after_nms: bool = False, **kw):
    """One frame: raw head outputs -> (detection_boxes[100,4], scores[100], classes[100] 1-based, n).

    `classes` carries the exporter's label offset (+1) on *every* row including the zero padding,
    which is why padded rows read class 1.  clip_after_nms: `multiclass_nms_clip_after`.
    """
    boxes = decode_boxes(box_enc.astype(np.float32), anchors_cs)
    scores = sigmoid(cls_logits)[:, 1:]
    if clip_after_nms:
        b, s, c, nd = multiclass_nms_clip_after(boxes, scores, **kw)
        return b, s, (c + F32(1.0)).astype(np.float32), nd
    b, s, c, nd = (multiclass_nms_global_order if fast else multiclass_nms)(boxes, scores, **kw)
    return b, s, (c + F32(1.0)).astype(np.float32), nd
