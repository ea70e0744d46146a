"""Oracle and engine against golden vectors of the REAL TensorFlow detector (tests/golden/make_tf_golden.py).

Skipped unless `tests/golden/tf_ssd_mobilenet_v2.npz` exists and WATSOR_TF_PB names the frozen graph it was made from
(the .pb supplies the weights through watsor_amd/frozen_graph.py).  With them present the oracle is PINNED: every stage
it restates (`watsor/detection/tensorflow_cpu.py:94-121`) is compared with what TensorFlow computed."""
import hashlib
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "tf_ssd_mobilenet_v2.npz")
PB = os.environ.get("WATSOR_TF_PB", "")

pytestmark = pytest.mark.skipif(not (os.path.isfile(GOLDEN) and os.path.isfile(PB)),
                                reason="no TensorFlow golden vectors / frozen graph here (see tests/golden/make_tf_golden.py)")


@pytest.fixture(scope="module")
def golden():
    g = dict(np.load(GOLDEN, allow_pickle=False))
    assert hashlib.sha256(open(PB, "rb").read()).hexdigest() == str(g["pb_sha256"]), "WATSOR_TF_PB is not the graph the vectors were made from"
    return g


@pytest.fixture(scope="module")
def tf_weights():
    from watsor_amd.frozen_graph import read_frozen_graph_variables
    return read_frozen_graph_variables(PB)


def frames_of(g):
    from watsor_amd.synth import synthetic_frame
    return [synthetic_frame(int(w), int(h), int(s)) for w, h, s in g["frames"]]


def test_oracle_matches_tensorflow_stage_by_stage(golden, tf_weights):
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


@pytest.mark.gpu
def test_engine_matches_tensorflow_within_the_north_star_tolerance(golden, tf_weights, tmp_path):
    import parity_utils as pu
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
            pairs, missing = pu.match_rows(rows[0], ref, min_score=0.05)
            assert len(missing) <= max(1, len(pairs) // 20)
            assert max(abs(p[3]) for p in pairs) <= 1e-3
    finally:
        e.close()
