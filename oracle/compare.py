"""Oracle (test infrastructure): row-level comparison of a detector's output with the oracle's.

Used by the GPU parity tests (tests/parity_utils.py), `__graft_entry__.smoke()` and the live parity leg of
`bench.py` -- as the checker, never on the product path.

What is compared is what the reference's plugin hands to the rest of Watsor (`watsor/detection/tensorflow_cpu.py:79-90`): per row a
label, a confidence (float32 widened to double) and an integer pixel box.  Three statements are checked:

  * scores: |confidence_gpu - confidence_oracle| <= SCORE_TOL on every matched row (north star: 1e-3);
  * boxes:  every coordinate of a matched row within `box_tolerance_px(width, height)` pixels of the oracle's;
  * completeness: every oracle row without a partner, and every GPU row without one, has a stated reason that follows from the
    score tolerance itself -- it sits at the top-100 cut, or it lost / won a greedy-NMS decision against a same-class neighbour
    whose score is within 2 x tolerance of its own or whose IoU is at the 0.6 threshold.  Anything else is `unexplained` and fails.
"""
from __future__ import annotations

import math

SCORE_TOL = 1e-3            # north star: "box scores within 1e-3 of the CPU reference"
BOX_TOL_FRACTION = 1.0 / 1000.0  # of the longer image side, and never less than 1 px (the int() truncation of tensorflow_cpu.py:86-89
                                 # alone moves a coordinate by one pixel when the float lands on either side of an integer).  Measured
                                 # (profiles/r04_parity_report_*.txt, 3 900 rows over 300x300 ... 1920x1080, default and robust program,
                                 # spread weights included): no coordinate off by more than 1 px at any size; 2 - 6 % of the rows by 1
NMS_IOU = 0.6               # oracle/postprocess.py
_IOU_PX_SLACK = 0.04        # pixel boxes are truncated: their IoU differs from the float boxes' by a few percent for small boxes


def box_tolerance_px(width, height):
    """Stated box tolerance in pixels for a frame of that size: 1 px up to 1000 pixels on the longer side (300x300, 640x480), 2 px
    up to 2000 (1280x720, 1920x1080), 4 px for 3840x2160."""
    return max(1, int(math.ceil(max(width, height) * BOX_TOL_FRACTION)))


def box_iou_px(a, b):
    """IoU of inclusive pixel boxes (x_min,y_min,x_max,y_max)."""
    ix0, iy0 = max(a[0], b[0]), max(a[1], b[1])
    ix1, iy1 = min(a[2], b[2]), min(a[3], b[3])
    iw, ih = max(ix1 - ix0 + 1, 0), max(iy1 - iy0 + 1, 0)
    inter = iw * ih
    ua = (a[2] - a[0] + 1) * (a[3] - a[1] + 1) + (b[2] - b[0] + 1) * (b[3] - b[1] + 1) - inter
    return inter / ua if ua > 0 else 0.0


def _gpu_box(gpu_rows, j):
    return (int(gpu_rows["x_min"][j]), int(gpu_rows["y_min"][j]), int(gpu_rows["x_max"][j]), int(gpu_rows["y_max"][j]))


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
            v = box_iou_px(rb, _gpu_box(gpu_rows, j))
            if v > best_iou:
                best, best_iou = j, v
        if best >= 0 and best_iou >= 0.9:
            used.add(best)
            pairs.append((i, best, best_iou, float(gpu_rows["confidence"][best] - ref["confidence"][i])))
        else:
            missing.append(i)
    return pairs, missing


def _explain(label, score, box, others, cut_score, full, tol):
    """Why a row of one detector (label, score, pixel box) has no partner among the other detector's rows.
    `others`: list of (label, score, box, is-an-explained-disagreement) of the other detector's rows; `cut_score`: the lower of the two detectors' last kept
    scores when both lists are full (`full`), i.e. where the top-100 cut lies."""
    if full and score <= cut_score + 2.0 * tol:
        return "top-100 cut (score within 2 x tolerance of the last kept row)"
    for (ol, os_, ob, unmatched) in others:
        if ol != label:
            continue
        iou = box_iou_px(box, ob)
        if iou >= NMS_IOU - _IOU_PX_SLACK and abs(os_ - score) <= 2.0 * tol:
            return "NMS order: a same-class neighbour (IoU %.2f) scores within 2 x tolerance (%.5f vs %.5f)" % (iou, os_, score)
        if abs(iou - NMS_IOU) <= _IOU_PX_SLACK and os_ >= score - 2.0 * tol:
            return "NMS threshold: IoU %.3f with a higher-scored same-class neighbour lies at the %.1f threshold" % (iou, NMS_IOU)
    for (ol, os_, ob, explained) in others:     # second-order: the row that suppressed it is one the other detector kept for a stated reason
        if ol == label and explained and os_ >= score - 2.0 * tol and box_iou_px(box, ob) >= NMS_IOU - _IOU_PX_SLACK:
            return "NMS cascade: suppressed by a same-class row (IoU %.2f) that is itself an explained disagreement" % box_iou_px(box, ob)
    return None


def compare_rows(gpu_rows, ref, image_shape=None, tol=SCORE_TOL, min_score=0.0):
    """Full comparison of one frame's rows.  Returns a dict:
         pairs           [(ref_idx, gpu_idx, iou, dscore, dbox_px)]
         max_dscore, max_dbox_px, box_tolerance_px (None without image_shape)
         missing         oracle rows (> min_score) without a partner:   [(ref_idx, reason or None)]
         extra           GPU rows (> min_score) without a partner:      [(gpu_idx, reason or None)]
         unexplained     how many of those have no reason
    """
    all_pairs, all_missing = match_rows(gpu_rows, ref, 0.0)   # (ref rows in score order: the rows above min_score are a prefix)
    used_gpu = {p[1] for p in all_pairs}
    missing0 = [i for i in all_missing if ref["confidence"][i] > min_score]
    pairs = []
    for (i, j, iou, ds) in all_pairs:
        if ref["confidence"][i] <= min_score:
            continue
        gb = _gpu_box(gpu_rows, j)
        dpx = max(abs(int(gb[k]) - int(ref["box"][i][k])) for k in range(4))
        pairs.append((i, j, iou, ds, dpx))
    n_gpu = int((gpu_rows["confidence"] > 0).sum())
    n_ref = int((ref["confidence"] > 0).sum())
    full = n_gpu >= len(gpu_rows) and n_ref >= len(ref["label"])
    cut = 0.0
    if full:
        cut = min(float(gpu_rows["confidence"][n_gpu - 1]), float(ref["confidence"][n_ref - 1]))
    def rows_of(kind, flagged):
        if kind == "gpu":
            return [(int(gpu_rows["label"][j]), float(gpu_rows["confidence"][j]), _gpu_box(gpu_rows, j), j in flagged)
                    for j in range(len(gpu_rows)) if gpu_rows["confidence"][j] > 0]
        return [(int(ref["label"][i]), float(ref["confidence"][i]), tuple(int(v) for v in ref["box"][i]), i in flagged)
                for i in range(len(ref["label"])) if ref["confidence"][i] > 0]

    extra0 = [j for j in range(len(gpu_rows)) if j not in used_gpu and gpu_rows["confidence"][j] > max(min_score, 0.0)]
    why_missing, why_extra = {}, {}
    for _ in range(2):   # pass 1: first-order reasons; pass 2: cascades through rows explained in pass 1
        gpu_list, ref_list = rows_of("gpu", set(why_extra)), rows_of("ref", set(why_missing))
        for i in missing0:
            if i not in why_missing:
                w = _explain(int(ref["label"][i]), float(ref["confidence"][i]), tuple(int(v) for v in ref["box"][i]), gpu_list, cut, full, tol)
                if w:
                    why_missing[i] = w
        for j in extra0:
            if j not in why_extra:
                w = _explain(int(gpu_rows["label"][j]), float(gpu_rows["confidence"][j]), _gpu_box(gpu_rows, j), ref_list, cut, full, tol)
                if w:
                    why_extra[j] = w
    missing = [(i, why_missing.get(i)) for i in missing0]
    extra = [(j, why_extra.get(j)) for j in extra0]
    out = dict(pairs=pairs, missing=missing, extra=extra,
               max_dscore=max([abs(p[3]) for p in pairs], default=0.0),
               max_dbox_px=max([p[4] for p in pairs], default=0),
               unexplained=sum(1 for m in missing if m[1] is None) + sum(1 for x in extra if x[1] is None),
               rows_reference=n_ref, rows_gpu=n_gpu,
               box_tolerance_px=box_tolerance_px(image_shape[1], image_shape[0]) if image_shape is not None else None)
    return out


def assert_rows_match(gpu_rows, ref, image_shape, tol=SCORE_TOL, min_score=0.0, what=""):
    """The three statements of the module header, as assertions.  Returns compare_rows()'s dict."""
    r = compare_rows(gpu_rows, ref, image_shape, tol, min_score)
    assert r["pairs"], "%s: no row of the oracle found a partner" % what
    assert r["max_dscore"] <= tol, "%s: max |dscore| %.6f > %g" % (what, r["max_dscore"], tol)
    assert r["max_dbox_px"] <= r["box_tolerance_px"], \
        "%s: max |dbox| %d px > %d px" % (what, r["max_dbox_px"], r["box_tolerance_px"])
    bad = [m for m in r["missing"] if m[1] is None] + [x for x in r["extra"] if x[1] is None]
    assert not bad, "%s: rows without a partner and without an explanation: missing %s extra %s" % (
        what, [m for m in r["missing"] if m[1] is None], [x for x in r["extra"] if x[1] is None])
    # explained disagreements are single decisions at the cut / at an NMS tie: a handful per frame at most
    assert len(r["missing"]) + len(r["extra"]) <= 6, "%s: %d + %d rows without a partner" % (what, len(r["missing"]), len(r["extra"]))
    return r
