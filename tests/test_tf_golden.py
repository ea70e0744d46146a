"""Oracle and engine against golden vectors of the REAL TensorFlow detector (tests/golden/make_tf_golden.py).

The two TensorFlow tests are skipped unless `tests/golden/tf_ssd_mobilenet_v2.npz` exists and WATSOR_TF_PB names the frozen graph
it was made from (the .pb supplies the weights through watsor_amd/frozen_graph.py).  With them present the oracle is PINNED: every
stage it restates (`watsor/detection/tensorflow_cpu.py:94-121`) is compared with what TensorFlow computed.

The route itself is kept warm without TensorFlow: the `self_test` tests build a golden file of the same
format from the ORACLE's outputs (the generator's own `collect()`), a frozen graph from the seeded weights (tests/pb_writer.py),
and run exactly the comparison code the TensorFlow tests run -- file format, sha check, weight import, stage comparisons and the
row matching are exercised in every CI run; they prove nothing about TensorFlow."""
import hashlib
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "tf_ssd_mobilenet_v2.npz")
PB = os.environ.get("WATSOR_TF_PB", "")

needs_tf_vectors = pytest.mark.skipif(not (os.path.isfile(GOLDEN) and os.path.isfile(PB)),
                                      reason="no TensorFlow golden vectors / frozen graph here (see tests/golden/make_tf_golden.py)")


def load_golden(npz_path, pb_path):
    g = dict(np.load(npz_path, allow_pickle=False))
    assert hashlib.sha256(open(pb_path, "rb").read()).hexdigest() == str(g["pb_sha256"]), "the .pb is not the graph the vectors were made from"
    return g


@pytest.fixture(scope="module")
def golden():
    return load_golden(GOLDEN, PB)


@pytest.fixture(scope="module")
def tf_weights():
    from watsor_amd.frozen_graph import read_frozen_graph_variables
    return read_frozen_graph_variables(PB)


@pytest.fixture(scope="module")
def self_made(tmp_path_factory):
    """(golden dict, weights) made HERE: the oracle stands in for TensorFlow, the seeded weights for the checkpoint."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_tf_golden as gen
    from pb_writer import write_detection_graph
    from oracle import postprocess as post
    from oracle import preprocess as pre
    from oracle.detect import OracleObjectDetector
    from watsor_amd.frozen_graph import read_frozen_graph_variables
    from watsor_amd.synth import synthetic_weights
    d = tmp_path_factory.mktemp("tf_route")
    pb = str(d / "frozen_inference_graph.pb")
    write_detection_graph(pb, synthetic_weights(1234))
    det = OracleObjectDetector(weights=read_frozen_graph_variables(pb))

    def run(frame):                                                # what sess.run(fetch) returns, shapes included
        b, c, s, be, lg = det.raw(frame)
        return {"detection_boxes": b[None], "detection_scores": s[None], "detection_classes": c[None],
                "num_detections": np.array([float((s > 0).sum())], np.float32), "preprocessed": pre.preprocess(frame)[None],
                "box_encodings": be[None], "class_logits": lg[None], "anchors": post.generate_anchors()}

    out = gen.collect(gen.FRAMES[:2] + gen.FRAMES[3:4], run, {"tf_version": "none (oracle stand-in)",
                                                              "pb_sha256": hashlib.sha256(open(pb, "rb").read()).hexdigest(),
                                                              "optional_tensor_names": "{}"})
    npz = str(d / "tf_ssd_mobilenet_v2.npz")
    np.savez_compressed(npz, **out)
    return load_golden(npz, pb), read_frozen_graph_variables(pb)


def frames_of(g):
    from watsor_amd.synth import synthetic_frame
    return [synthetic_frame(int(w), int(h), int(s)) for w, h, s in g["frames"]]


def check_oracle_stage_by_stage(golden, tf_weights):
    from oracle import postprocess as post
    from oracle import preprocess as pre
    from oracle.detect import OracleObjectDetector
    det = OracleObjectDetector(weights=tf_weights)
    for i, f in enumerate(frames_of(golden)):
        b, c, s, be, lg = det.raw(f)
        if "f%d_preprocessed" % i in golden:                      # resize coordinate rule + normalisation
            np.testing.assert_allclose(pre.preprocess(f)[None], golden["f%d_preprocessed" % i], rtol=0, atol=2e-6)
        if "f%d_anchors" % i in golden:
            a = golden["f%d_anchors" % i].reshape(-1, 4)          # (ymin, xmin, ymax, xmax) corners
            np.testing.assert_allclose(post.generate_anchors(), a, rtol=0, atol=1e-6)
        if "f%d_box_encodings" % i in golden:
            np.testing.assert_allclose(be, golden["f%d_box_encodings" % i].reshape(be.shape), rtol=0, atol=2e-4)
        if "f%d_class_logits" % i in golden:
            np.testing.assert_allclose(lg, golden["f%d_class_logits" % i].reshape(lg.shape), rtol=0, atol=2e-4)
        n = int(golden["f%d_num_detections" % i][0])
        np.testing.assert_array_equal(c[:n], golden["f%d_detection_classes" % i][0][:n])
        np.testing.assert_allclose(s[:n], golden["f%d_detection_scores" % i][0][:n], rtol=0, atol=2e-5)
        np.testing.assert_allclose(b[:n], golden["f%d_detection_boxes" % i][0][:n], rtol=0, atol=2e-5)


@needs_tf_vectors
def test_oracle_matches_tensorflow_stage_by_stage(golden, tf_weights):
    check_oracle_stage_by_stage(golden, tf_weights)


def test_self_test_of_the_oracle_comparison(self_made):
    check_oracle_stage_by_stage(*self_made)
    g = dict(self_made[0])                                         # ... and the comparison does notice a difference
    g["f0_detection_scores"] = g["f0_detection_scores"] + np.float32(1e-3)
    with pytest.raises(AssertionError):
        check_oracle_stage_by_stage(g, self_made[1])


def test_nms_order_witness_tells_the_two_exporter_generations_apart():
    """`make_tf_golden.nms_order_witness`: which clip / NMS order a graph has, read off its own raw head outputs and final detections
    (the first TensorFlow run settles `--clip-after-nms` by data).  Stand-in "graphs": the oracle in either order
    on head outputs whose boxes straddle the image border (the two orders keep different rows there)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_tf_golden as gen
    from oracle import postprocess as post
    rng = np.random.default_rng(5)
    anchors = post.generate_anchors()
    acs = post.anchors_center_size(anchors)
    be = np.zeros((1917, 4), np.float32)
    be[:, 2:] = -3.0
    lg = (rng.standard_normal((1917, 91)) - 7.0).astype(np.float32)
    for _ in range(40):                                            # objects near / across the border, a few jittered anchors each
        cy, cx = rng.uniform(-0.15, 1.15, 2)
        h, w = rng.uniform(0.15, 0.6, 2)
        cls = int(rng.integers(1, 91))
        near = np.argsort((acs[:, 0] - cy) ** 2 + (acs[:, 1] - cx) ** 2)[:18]
        for a in rng.choice(near, 6, replace=False):
            jy, jx = rng.normal(0, 0.02, 2)
            ay, ax, ah, aw = acs[a]
            be[a] = [((cy + jy) - ay) / ah * 10, ((cx + jx) - ax) / aw * 10, np.log(h / ah) * 5, np.log(w / aw) * 5]
            lg[a, cls] = rng.uniform(0.5, 4.0)
    for clip_after, want in ((False, "clip_before_nms"), (True, "clip_after_nms")):
        b, s_, c, n = post.postprocess(be, lg, acs, clip_after_nms=clip_after, **({} if clip_after else {"fast": True}))
        g = {"frames": np.array([[640, 480, 1]], np.int32), "f0_box_encodings": be[None], "f0_class_logits": lg[None], "f0_anchors": anchors,
             "f0_detection_boxes": b[None], "f0_detection_scores": s_[None], "f0_num_detections": np.array([float(n)], np.float32)}
        w_ = gen.nms_order_witness(g)
        assert w_["order"] == want and w_["frames_telling_them_apart"] == 1, w_
    assert gen.nms_order_witness({"frames": np.array([[1, 1, 1]], np.int32)})["order"] == "no raw head outputs in the file"


def check_engine_rows(golden, tf_weights, tmp_path):
    from oracle.compare import assert_rows_match
    from oracle.detect import rows_as_array
    from conftest import make_engine
    from watsor_amd import engine
    from watsor_amd.runtime import ROW_DTYPE
    engine.save_engine(engine.build_engine(tf_weights), str(tmp_path / "mi355x.bin"))
    e = make_engine(str(tmp_path), max_batch=1)
    try:
        for i, f in enumerate(frames_of(golden)):
            rows = [np.zeros(100, ROW_DTYPE)]
            e.detect_batch([f], rows)
            ref = rows_as_array(f.shape, golden["f%d_detection_boxes" % i][0], golden["f%d_detection_classes" % i][0],
                                golden["f%d_detection_scores" % i][0])
            assert_rows_match(rows[0], ref, f.shape, min_score=0.05, what="frame %d" % i)   # scores 1e-3, stated box tolerance, every odd row explained
    finally:
        e.close()


@pytest.mark.gpu
@needs_tf_vectors
def test_engine_matches_tensorflow_within_the_north_star_tolerance(golden, tf_weights, tmp_path):
    check_engine_rows(golden, tf_weights, tmp_path)


@pytest.mark.gpu
def test_self_test_of_the_engine_comparison(self_made, tmp_path):
    check_engine_rows(*self_made, tmp_path)
