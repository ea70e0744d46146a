"""Score tolerance on weights whose channels do NOT all live at one scale: `spread_channel_scales`
spreads the per-channel amplitudes the way a folded trained BatchNorm does.  What is asserted is what holds: the `-p 32` engine keeps
the north star's 1e-3 at any spread; the `-p 16` engine keeps it at the spread of He-initialised weights and at 0.3 decades, and is
MEASURED (printed, bounded loosely) beyond -- the builder warns there (tests/test_engine_pack.py), DESIGN.md section 4 has the table."""
import numpy as np
import pytest

import parity_utils as pu
from oracle import detect as odet
from oracle.compare import assert_rows_match
from watsor_amd import engine
from watsor_amd.runtime import ROW_DTYPE
from watsor_amd.synth import spread_channel_scales, synthetic_frame

pytestmark = pytest.mark.gpu


def worst_score_error(model_dir, weights, frames):
    from watsor_amd.detection.hip_gpu import HipObjectDetector
    from watsor_amd.share import DetectionArray
    oracle = odet.OracleObjectDetector(weights=weights)
    worst, n = 0.0, 0
    with HipObjectDetector(model_dir, 0, max_batch=1, max_width=640, max_height=480) as det:
        for f in frames:
            rows = DetectionArray()
            det.detect(f.shape, f, rows)
            got = np.frombuffer(rows, dtype=ROW_DTYPE)
            b, c, s, _, _ = oracle.raw(f)
            ref = odet.rows_as_array(f.shape, b, c, s)
            pairs, missing = pu.match_rows(got, ref, min_score=0.1)
            assert len(pairs) >= 50
            worst = max(worst, max(abs(p[3]) for p in pairs))
            n += len(pairs)
    return worst, n


@pytest.mark.parametrize("decades,bar16", [(0.3, 1e-3), (1.0, 1e-2), (1.5, 1e-2)])
def test_score_error_under_channel_spread(tmp_path, synth_weights, decades, bar16):
    W = spread_channel_scales(synth_weights, decades)
    frames = [synthetic_frame(640, 480, 5000 + i) for i in range(3)]
    out = {}
    for prec in (16, 32):
        d = tmp_path / ("p%d" % prec)
        engine.save_engine(engine.build_engine(W, precision=prec), str(d / "mi355x.bin"))
        out[prec] = worst_score_error(str(d), W, frames)
    print("\\nchannel spread %.1f decades (measured %.2f): max |dscore| -p 16 %.2e, -p 32 %.2e over %d rows"
          % (decades, engine.channel_spread_decades(W), out[16][0], out[32][0], out[16][1]))
    assert out[32][0] <= 1e-4                       # the fp32 engine (pair input, exact-fp32 matrix cores): far inside the tolerance at any spread
    assert out[16][0] <= bar16                      # the fp16 engine: the tolerance up to what was validated, a sanity bound beyond


@pytest.mark.parametrize("decades,bar", [(0.0, 5e-4), (1.0, 1e-3), (1.5, 1e-3), (2.0, 1e-3)])
def test_robust_program_holds_the_tolerance_under_channel_spread(tmp_path, synth_weights, decades, bar):
    """`build_engine(robust=True)` -- all 17 blocks on the split-operand kernel, expanded tensors in the 16-bit float form, Conv_1 with
    split weights: the north star's 1e-3 at 0 / 1.0 / 1.5 / 2.0 decades of per-channel spread (the default program is at 3e-3 at 1.5),
    on all 100 rows, boxes within the stated pixel tolerance and no unexplained row (oracle/compare.py).  BatchNorm's epsilon bounds
    what folding can do to a channel at ~30x, i.e. 1.5 decades.  The frames go through in ONE batch of mixed resolutions, so the lean
    builds behind the 19x19 maps run their multi-frame grids."""
    from watsor_amd.runtime import HipEngine
    from watsor_amd.share import DetectionArray
    W = spread_channel_scales(synth_weights, decades) if decades else synth_weights
    sizes = [(640, 480), (1280, 720), (320, 240), (640, 360), (640, 480)]
    frames = [synthetic_frame(w, h, 5100 + i) for i, (w, h) in enumerate(sizes)]
    path = str(tmp_path / "robust" / "mi355x.bin")
    engine.save_engine(engine.build_engine(W, robust=True), path)
    oracle = odet.OracleObjectDetector(weights=W)
    eng = HipEngine(path, 0, 8, 1280, 720)
    try:
        assert eng.hp_blocks == 17
        rows = [DetectionArray() for _ in frames]
        eng.detect_batch(frames, rows)
        worst, n = 0.0, 0
        for f, r in zip(frames, rows):
            got = np.frombuffer(r, dtype=ROW_DTYPE)
            b, c, s, _, _ = oracle.raw(f)
            ref = odet.rows_as_array(f.shape, b, c, s)
            r = assert_rows_match(got, ref, f.shape, tol=bar, what="spread %.1f, %dx%d" % (decades, f.shape[1], f.shape[0]))
            assert len(r["pairs"]) >= 90
            worst = max(worst, r["max_dscore"])
            n += len(r["pairs"])
    finally:
        eng.close()
    print("\\nrobust program, channel spread %.1f decades: max |dscore| %.2e over %d rows" % (decades, worst, n))
    assert worst <= bar


@pytest.mark.parametrize("batch", [1, 2, 3])
def test_robust_program_with_channel_groups_over_workgroups(tmp_path, synth_weights, batch):
    """Few frames in a batch: the 10x10 split blocks deal their chunks out over 2 - 4 WORKGROUPS per tile whose partial sums meet through
    the workspace (a ticket per tile, the last arriver adds the groups in order: `wz_k_mbconv_hp`'s CG builds; one frame: 3 groups for
    block 13, 4 for blocks 14 .. 16; two and three frames: 2 - 4).  Same tolerance as any other launch shape, and the same rows every time."""
    from watsor_amd.runtime import HipEngine
    from watsor_amd.share import DetectionArray
    W = spread_channel_scales(synth_weights, 1.0)
    frames = [synthetic_frame(640, 480, 5200 + i) for i in range(batch)]
    path = str(tmp_path / "robust" / "mi355x.bin")
    engine.save_engine(engine.build_engine(W, robust=True), path)
    oracle = odet.OracleObjectDetector(weights=W)
    eng = HipEngine(path, 0, 4, 640, 480)
    try:
        first = None
        for rep in range(3):
            rows = [DetectionArray() for _ in frames]
            eng.detect_batch(frames, rows)
            got = [np.frombuffer(r, dtype=ROW_DTYPE).copy() for r in rows]
            if first is None:
                first = got
                for f, g in zip(frames, got):
                    b, c, s, _, _ = oracle.raw(f)
                    r = assert_rows_match(g, odet.rows_as_array(f.shape, b, c, s), f.shape, tol=1e-3, what="batch of %d" % batch)
                    assert len(r["pairs"]) >= 90
            else:
                for a, g in zip(first, got):
                    assert a.tobytes() == g.tobytes()           # fixed summation order: bit for bit
    finally:
        eng.close()
