"""Golden vectors of the REAL reference detector -- the route from "parity unpinned" to "pinned" (DESIGN.md section 3).

The reference's CPU plugin is `sess.run` on `frozen_inference_graph.pb` of ssd_mobilenet_v2_coco_2018_03_29
(`watsor/detection/tensorflow_cpu.py:50-62,94-121`, README.md:446-451).  Neither TensorFlow nor that file exists in the
build container, so the oracle's resize rule, extras / heads, anchors and NMS are restated from the published graph
semantics only.  Run this script anywhere both exist:

    python tests/golden/make_tf_golden.py --pb /path/to/frozen_inference_graph.pb

It feeds the seeded synthetic frames of the test-suite through the graph exactly like the plugin does
(`image_tensor:0` <- the full-resolution uint8 frame, `tensorflow_cpu.py:113-115`) and writes
`tests/golden/tf_ssd_mobilenet_v2.npz`: the four fetched outputs of the plugin plus, where the graph has them, the
intermediate tensors the oracle restates (normalised resized image, raw box encodings, class logits, anchors).
`tests/test_tf_golden.py` consumes the file when it is present (with WATSOR_TF_PB pointing at the same .pb, which supplies
the weights): oracle vs TensorFlow stage by stage on the CPU, engine vs TensorFlow within the north star's 1e-3 on the GPU.
"""
import argparse
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

FRAMES = [(640, 480, 1234), (640, 480, 1235), (640, 480, 1236), (1280, 720, 2000), (1920, 1080, 3000), (300, 300, 4000),
          (301, 299, 4001)]
# (the anchors the graph generates travel in the file as `f<i>_anchors` where the graph exposes them -- OPTIONAL below --, and the
#  stage-by-stage test compares the oracle's anchor table with them: tests/test_tf_golden.py)
OUTPUTS = ["detection_boxes", "detection_scores", "detection_classes", "num_detections"]       # tensorflow_cpu.py:98-110
# intermediate tensors of the TF-OD-API export (name candidates by exporter version; recorded only when present)
OPTIONAL = {
    "preprocessed": ["Preprocessor/sub"],
    "box_encodings": ["concat", "Squeeze"],
    "class_logits": ["concat_1"],
    "anchors": ["MultipleGridAnchorGenerator/Concatenate/concat", "Concatenate/concat"],
    "scores_all": ["Postprocessor/convert_scores", "Postprocessor/scale_logits"],
}


def collect(frames, run, meta):
    """The arrays of the golden file: `run(frame)` -> {tensor key: array} for every (width, height, seed) of `frames`.  Shared by the
    TensorFlow route below and by the self-test of tests/test_tf_golden.py, which feeds it the ORACLE's outputs so that the file
    format and the comparison code are exercised in every CI run, not first on the day TensorFlow shows up."""
    from watsor_amd.synth import synthetic_frame
    out = {"frames": np.array(frames, np.int32)}
    out.update({k: np.array(v) for k, v in meta.items()})
    for i, (w, h, seed) in enumerate(frames):
        for k, v in run(synthetic_frame(w, h, seed)).items():
            out["f%d_%s" % (i, k)] = np.asarray(v)
    return out


def nms_order_witness(golden):
    """Which post-processing order the graph that made `golden` has -- decided from its OWN tensors, not from a flag: the oracle's two
    orders (clip -> drop zero-area -> per-class NMS, the 2018 exporter; per-class NMS on the boxes as decoded -> clip, later exporters:
    SURVEY.md App. B.5) are run on the graph's raw box encodings, class logits and anchors of every frame, and each is compared with the
    graph's final detections.  Returns {"order": "clip_before_nms" | "clip_after_nms" | "undecided" | "no raw head outputs in the file",
    "frames_matching_clip_before": n, "frames_matching_clip_after": n, "frames_telling_them_apart": n}: `python -m watsor_amd.engine`
    wants `--clip-after-nms` exactly when the answer is clip_after_nms.  (Frames on which both orders give the same rows -- nothing
    near the image border -- tell nothing and count in neither column.)"""
    from oracle import postprocess as post
    n_frames = len(golden["frames"])
    before = after = apart = 0
    seen = False
    for i in range(n_frames):
        if not all(("f%d_%s" % (i, k)) in golden for k in ("box_encodings", "class_logits", "detection_boxes", "detection_scores")):
            continue
        seen = True
        be = np.asarray(golden["f%d_box_encodings" % i], np.float32).reshape(-1, 4)
        lg = np.asarray(golden["f%d_class_logits" % i], np.float32).reshape(be.shape[0], -1)
        anchors = np.asarray(golden["f%d_anchors" % i], np.float32).reshape(-1, 4) if ("f%d_anchors" % i) in golden else post.generate_anchors()
        acs = post.anchors_center_size(anchors)
        n = int(np.asarray(golden["f%d_num_detections" % i]).reshape(-1)[0]) if ("f%d_num_detections" % i) in golden else 100
        tb = np.asarray(golden["f%d_detection_boxes" % i], np.float32).reshape(-1, 4)[:n]
        ts = np.asarray(golden["f%d_detection_scores" % i], np.float32).reshape(-1)[:n]

        def same(res):
            b, s_, _, _ = res
            return bool(np.allclose(b[:n], tb, rtol=0, atol=2e-5) and np.allclose(s_[:n], ts, rtol=0, atol=2e-5))
        r0 = post.postprocess(be, lg, acs, fast=True)
        r1 = post.postprocess(be, lg, acs, clip_after_nms=True)
        differ = not (np.allclose(r0[0], r1[0], rtol=0, atol=1e-6) and np.allclose(r0[1], r1[1], rtol=0, atol=1e-7))
        if not differ:
            continue
        apart += 1
        before += int(same(r0))
        after += int(same(r1))
    if not seen:
        order = "no raw head outputs in the file"
    elif before > 0 and after == 0:
        order = "clip_before_nms"
    elif after > 0 and before == 0:
        order = "clip_after_nms"
    else:
        order = "undecided"
    return dict(order=order, frames_matching_clip_before=before, frames_matching_clip_after=after, frames_telling_them_apart=apart)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pb", required=True)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "tf_ssd_mobilenet_v2.npz"))
    args = ap.parse_args()
    import tensorflow as tf
    tf1 = tf.compat.v1 if hasattr(tf, "compat") and hasattr(tf.compat, "v1") else tf

    graph = tf.Graph()
    with graph.as_default():                                       # tensorflow_cpu.py:50-62
        gd = tf1.GraphDef()
        with open(args.pb, "rb") as f:
            blob = f.read()
        gd.ParseFromString(blob)
        tf.import_graph_def(gd, name="")
    names = {op.name for op in graph.get_operations()}
    fetch = {k: graph.get_tensor_by_name(k + ":0") for k in OUTPUTS}
    found = {}
    for key, cands in OPTIONAL.items():
        for c in cands:
            if c in names:
                fetch[key] = graph.get_tensor_by_name(c + ":0")
                found[key] = c
                break
    meta = {"tf_version": tf.__version__, "pb_sha256": hashlib.sha256(blob).hexdigest(), "optional_tensor_names": repr(found)}
    with tf1.Session(graph=graph) as sess:
        out = collect(FRAMES, lambda frame: sess.run(fetch, feed_dict={graph.get_tensor_by_name("image_tensor:0"): frame[None]}), meta)
    witness = nms_order_witness(out)           # settles `--clip-after-nms` by data
    out["nms_order_witness"] = np.array(repr(witness))
    np.savez_compressed(args.out, **out)
    print("wrote %s (%d arrays; intermediates found: %s)" % (args.out, len(out), found))
    print("post-processing order of this graph, from its own tensors: %s" % witness)
    if witness["order"] == "clip_after_nms":
        print("  -> build its engine with `python -m watsor_amd.engine --clip-after-nms ...`")


if __name__ == "__main__":
    main()
