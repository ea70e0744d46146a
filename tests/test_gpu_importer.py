"""SURVEY 8(f)-4 on the GPU: an engine built through the model-file path the reference uses
(`watsor/detection/tensorflow_cpu.py:50-62` loads `frozen_inference_graph.pb`; README.md:446-451) detects exactly like
the engine built from the same variables handed over as a dict."""
import os

import numpy as np
import pytest

pytest.importorskip("google.protobuf")
from conftest import make_engine                       # noqa: E402
from pb_writer import write_frozen_graph                # noqa: E402
from watsor_amd import engine                          # noqa: E402
from watsor_amd.runtime import ROW_DTYPE               # noqa: E402
from watsor_amd.synth import synthetic_frame           # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", [16, 32])
def test_engine_from_a_frozen_graph_detects_identically(model_dir_default, model_dir_fp32, synth_weights, tmp_path, precision):
    pb = tmp_path / "frozen_inference_graph.pb"
    write_frozen_graph(str(pb), synth_weights)
    out = tmp_path / "model" / "mi355x.bin"
    assert engine.main(["-i", str(pb), "-o", str(out), "-p", str(precision)]) == 0     # the CLI, like watsor/engine.py:61-107
    ref_dir = model_dir_default if precision == 16 else model_dir_fp32
    assert open(out, "rb").read() == open(os.path.join(ref_dir, "mi355x.bin"), "rb").read()
    frames = [synthetic_frame(640, 480, 50 + i) for i in range(3)] + [synthetic_frame(1280, 720, 60)]
    a, b = make_engine(str(tmp_path / "model"), max_batch=4), make_engine(ref_dir, max_batch=4)
    try:
        assert a.precision == precision and a.hp_blocks == (13 if precision == 16 else 0)
        ra = [np.zeros(100, ROW_DTYPE) for _ in frames]
        rb = [np.zeros(100, ROW_DTYPE) for _ in frames]
        a.detect_batch(frames, ra)
        b.detect_batch(frames, rb)
        for x, y in zip(ra, rb):
            assert x.tobytes() == y.tobytes() and x["confidence"][0] > 0
    finally:
        a.close()
        b.close()


def test_frozen_graph_with_spread_channels_gets_the_robust_program_and_holds_the_tolerance(synth_weights, tmp_path, capsys):
    """The path a user with a TRAINED model file takes: `python -m watsor_amd.engine -i frozen_inference_graph.pb` with nothing else
    said.  The builder measures the per-channel spread of the folded weights (1.36 decades here), packs the robust program for it,
    and the detector's scores stay within the north star's 1e-3 of the CPU detector on the plugin path (`HipObjectDetector.detect`)."""
    import parity_utils as pu
    from oracle import detect as odet
    from watsor_amd.detection.hip_gpu import HipObjectDetector
    from watsor_amd.share import DetectionArray
    from watsor_amd.synth import spread_channel_scales
    W = spread_channel_scales(synth_weights, 1.5)
    pb = tmp_path / "frozen_inference_graph.pb"
    write_frozen_graph(str(pb), W)
    out = tmp_path / "model" / "mi355x.bin"
    assert engine.main(["-i", str(pb), "-o", str(out)]) == 0
    cap = capsys.readouterr()
    assert "robust" in cap.out and "WARNING" not in cap.err
    oracle = odet.OracleObjectDetector(weights=W)
    worst, n = 0.0, 0
    with HipObjectDetector(str(tmp_path / "model"), 0, max_batch=2, max_width=1280, max_height=720) as det:
        assert det.engine.hp_blocks == 17
        for f in (synthetic_frame(640, 480, 71), synthetic_frame(1280, 720, 72)):
            rows = DetectionArray()
            det.detect(f.shape, f, rows)
            got = np.frombuffer(rows, dtype=ROW_DTYPE)
            b, c, s, _, _ = oracle.raw(f)
            pairs, missing = pu.match_rows(got, odet.rows_as_array(f.shape, b, c, s), min_score=0.1)
            assert len(pairs) >= 50 and len(missing) <= 2
            worst = max(worst, max(abs(p[3]) for p in pairs))
            n += len(pairs)
    assert worst <= 1e-3, worst
