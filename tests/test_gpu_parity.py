"""HIP path vs the CPU oracle, stage by stage and end to end, all through the C ABI (ctypes).

Tolerances (stated here; measured values: DESIGN.md section 4, the pytest logs under profiles/*_pytest_gpu.txt):
  * resize+normalise ............ bit-exact fp16 (integer/byte stage of the oracle, fp32 ops in TF order)
  * row fill (int truncation) .... bit-exact
  * post-processing on identical fp32 head outputs: same (class, anchor) rows, |score| <= 1e-6,
    |box| <= 2e-6 (GPU expf vs numpy expf differ by <= 2 ulp)
  * network, default `-p 16` engine (fp16 MFMA everywhere; the stem and blocks 0 .. 12 with split hi + lo
    operands, csrc/k_mbconv_hp.hip) vs the fp32 oracle: every sigmoid score and every detection row's
    confidence within SCORE_TOL = 1e-3, the north star's bar ("box scores within 1e-3 of the CPU
    reference"); tools/err_budget.py predicts 5e-4 for this program
  * `--plain-fp16` engine (one fp16 rounding per operand everywhere): SCORE_TOL_PLAIN = 4e-3 (measured
    2.7e-3: the ~53-layer random-init network amplifies the 2^-11 roundings)
  * `-p 32` engine: 1e-4 on all scores, 1e-3 asserted end to end
"""
import os

import numpy as np
import pytest

import parity_utils as pu
import conftest


def make_engine(*args, **kwargs):
    """Stage-level entry points and WZ_* knobs live in the development library (include/watsor_hip.h, WZ_DEV_BUILD section)."""
    kwargs.setdefault("dev", True)
    return conftest.make_engine(*args, **kwargs)

from oracle import detect as odet
from oracle import preprocess as pre
from watsor_amd.runtime import ROW_DTYPE
from watsor_amd.synth import synthetic_frame

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-3        # |sigmoid(logit_gpu) - sigmoid(logit_oracle)| over all 1917*91 entries, default engine
SCORE_TOL_PLAIN = 4e-3  # the same for the --plain-fp16 engine
LOGIT_TOL = 0.05        # max abs logit error of the plain fp16 programs vs the fp32 oracle
BOXENC_TOL = 0.04


@pytest.fixture(scope="module")
def eng(model_dir):
    e = make_engine(model_dir)
    yield e
    e.close()


def _keep_engine(path, **env):
    env = dict(env, WZ_NO_BUFFER_REUSE="1")
    os.environ.update(env)
    try:
        return make_engine(path, max_batch=2)
    finally:
        for k in env:
            os.environ.pop(k)


@pytest.fixture(scope="module")
def eng_keep(model_dir_unfused):
    """One op per layer, activation tensors do not share buffers: every layer is readable after a forward."""
    e = _keep_engine(model_dir_unfused)
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng_keep_fused(model_dir):
    """The default program (fused inverted-residual blocks), tensors not sharing buffers."""
    e = _keep_engine(model_dir)
    yield e
    e.close()


@pytest.fixture(scope="module")
def head_outputs(oracle_net, frames_640):
    x_half = pu.oracle_input_half(frames_640[:2])
    rbe, rlg, T = pu.oracle_forward_from_half(oracle_net, x_half, keep=True)
    return x_half, rbe, rlg, T


@pytest.mark.parametrize("wh", [(640, 480), (1280, 720), (1920, 1080), (300, 300), (301, 299), (64, 48)])
def test_preprocess_bit_exact(eng, wh):
    f = synthetic_frame(wh[0], wh[1], 7 + wh[0])
    got = eng.stage_preprocess(f)
    ref = pre.preprocess_fp16(f)
    np.testing.assert_array_equal(got[..., :3].view(np.uint16), ref.view(np.uint16))
    assert not got[..., 3].any()
    # the default program's input tensor is a pair: lo = RN16(v - RN16(v)) of the same fp32 value, bit for bit
    assert got.shape[-1] == 8
    lo = (pre.preprocess(f) - ref.astype(np.float32)).astype(np.float16)
    np.testing.assert_array_equal(got[..., 4:7].view(np.uint16), lo.view(np.uint16))
    assert not got[..., 7].any()


def test_preprocess_plain_program(model_dir_plain):
    e = make_engine(model_dir_plain, max_batch=1)
    try:
        f = synthetic_frame(640, 480, 77)
        got = e.stage_preprocess(f)
        assert got.shape[-1] == 4 and e.hp_blocks == 0
        np.testing.assert_array_equal(got[..., :3].view(np.uint16), pre.preprocess_fp16(f).view(np.uint16))
    finally:
        e.close()


def test_every_layer_close_to_oracle(eng_keep, head_outputs):
    x_half, rbe, rlg, T = head_outputs
    be, lg = eng_keep.stage_forward(x_half)
    worst = 0.0
    for idx, (name, h, w, c) in enumerate(eng_keep.tensors()):
        if name == "input":
            continue
        got = np.stack([eng_keep.stage_read_tensor(idx, f) for f in range(2)]).astype(np.float32)
        ref = T[name]
        assert got.shape == ref.shape
        err = np.abs(got - ref).max()
        scale = np.abs(ref).max()
        worst = max(worst, err / scale)
        assert err <= 0.04 * scale + 0.02, "%s: max abs err %.4f (max|ref| %.3f)" % (name, err, scale)
    assert np.abs(be - rbe).max() <= BOXENC_TOL
    assert np.abs(lg - rlg).max() <= LOGIT_TOL


def _compare_programs(e_unf, e_fused, x_half, rel):
    a = e_unf.stage_forward(x_half)
    b = e_fused.stage_forward(x_half)
    unf = {t[0]: i for i, t in enumerate(e_unf.tensors())}
    fused = e_fused.tensors()
    assert len(fused) < len(unf) and any(k == 4 for k in [o["kind"] for o in e_fused.ops()])
    bad = []
    for idx, (name, h, w, c) in enumerate(fused):
        if name == "input":
            continue
        for f in range(2):
            x = e_fused.stage_read_tensor(idx, f).astype(np.float32)
            y = e_unf.stage_read_tensor(unf[name], f).astype(np.float32)
            d = np.abs(x - y)
            if d.max() > rel * np.abs(y).max():
                bad.append("%s[%d]: %d of %d differ, max %.4g (max|ref| %.3g) at %s" % (
                    name, f, int((d > 0).sum()), d.size, d.max(), np.abs(y).max(), np.unravel_index(d.argmax(), d.shape)))
    assert not bad, "\n".join(bad[:12])
    return a, b


def test_fused_blocks_equal_unfused_layers(model_dir_stem_separate, model_dir_unfused, head_outputs):
    """One launch per inverted-residual block (k_mbconv.hip) rounds at the same points and, when the
    expanded channels are not spread over workgroups (WZ_SPLITK=0), accumulates in the same order as the
    per-layer kernels: every tensor both programs hold is bit-identical."""
    e_unf = _keep_engine(model_dir_unfused, WZ_SPLITK="0")
    e_fus = _keep_engine(model_dir_stem_separate, WZ_SPLITK="0")
    try:
        a, b = _compare_programs(e_unf, e_fus, head_outputs[0], 0.0)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    finally:
        e_unf.close()
        e_fus.close()


def test_both_programs_close_to_oracle_tensor_by_tensor(eng_keep_fused, model_dir, head_outputs):
    """Every tensor a `-p 16` program holds vs the fp32 oracle, for BOTH programs (`model_dir` is parametrised).
    default -- stem inside the first block's launch, blocks 0 .. 12 with split operands: the pair tensors (outputs of blocks 0 .. 11)
    carry no fp16 rounding at all: 5e-4 of the tensor's range covers the unorm16 chunk buffer (step 9e-5) and fp32 summation order;
    block 12's output is one fp16 rounding of such a value; everything behind it is plain fp16 as before.
    robust -- all 17 blocks split: outputs of blocks 0 .. 15 are pairs (blocks 0 .. 9 through the 16-bit float-form chunk buffer,
    10 .. 16 through the linear one on the lean builds): the same pair bound; the first SSD feature map (block 13's expanded tensor,
    stored by the block as its second output) and block 16's output are ONE fp16 rounding of such values -- block 16's twice per pixel
    ([x | x], what Conv_1's split weights multiply); Conv_1, the extras and the heads are plain fp16 behind exact inputs, so they
    are held to a quarter of the plain bound."""
    x_half, rbe, rlg, T = head_outputs
    e = eng_keep_fused
    robust = model_dir.program == "robust"
    be, lg = e.stage_forward(x_half)
    names = [t[0] for t in e.tensors()]
    assert "Conv" not in names and "expanded_conv/output" in names and e.hp_blocks == (17 if robust else 13)
    one_rounding = ("expanded_conv_16/output", "expanded_conv_13/expand") if robust else ("expanded_conv_12/output",)
    n_pair, worst = 0, {}
    for idx, (name, h, w, c) in enumerate(e.tensors()):
        if name == "input":
            assert e.tensor_is_pair(idx)
            continue
        got = np.stack([e.stage_read_tensor(idx, f) for f in range(2)]).astype(np.float32)
        ref = T[name]
        if robust and name == "expanded_conv_16/output":
            assert c == 2 * ref.shape[-1] and not e.tensor_is_pair(idx)
            np.testing.assert_array_equal(got[..., :c // 2], got[..., c // 2:])        # the fp16 output, twice per pixel
            got = got[..., :c // 2]
        err, scale = np.abs(got - ref).max(), np.abs(ref).max()
        if e.tensor_is_pair(idx):
            n_pair += 1
            kind = "pair"
            assert err <= 5e-4 * scale + 1e-4, "%s (pair): max abs err %.3g (max|ref| %.3f)" % (name, err, scale)
        elif name in one_rounding:
            kind = "one fp16 rounding"
            assert err <= 1.5e-3 * scale + 1e-4, "%s: max abs err %.3g (max|ref| %.3f)" % (name, err, scale)
        else:
            kind = "plain fp16"
            k = 0.25 if robust else 1.0
            assert err <= k * (0.04 * scale + 0.02), "%s: max abs err %.4f (max|ref| %.3f)" % (name, err, scale)
        worst[kind] = max(worst.get(kind, 0.0), err / scale)
    assert n_pair == (16 if robust else 12)
    print("\n%s program, worst relative error per tensor kind: %s" % (model_dir.program, {k: "%.2e" % v for k, v in worst.items()}))
    from oracle.postprocess import sigmoid
    assert np.abs(be - rbe).max() <= BOXENC_TOL / 4 and np.abs(lg - rlg).max() <= LOGIT_TOL / 4
    assert np.abs(sigmoid(lg) - sigmoid(rlg)).max() <= SCORE_TOL


def test_plain_fp16_program_close_to_oracle(model_dir_plain, head_outputs):
    """`--plain-fp16` (stem inside the first block's launch, one fp16 rounding per operand): every tensor vs the fp32 oracle."""
    from oracle.postprocess import sigmoid
    x_half, rbe, rlg, T = head_outputs
    e = _keep_engine(model_dir_plain)
    try:
        be, lg = e.stage_forward(x_half)
        for idx, (name, h, w, c) in enumerate(e.tensors()):
            if name == "input":
                continue
            assert not e.tensor_is_pair(idx)
            got = np.stack([e.stage_read_tensor(idx, f) for f in range(2)]).astype(np.float32)
            err, scale = np.abs(got - T[name]).max(), np.abs(T[name]).max()
            assert err <= 0.04 * scale + 0.02, "%s: max abs err %.4f (max|ref| %.3f)" % (name, err, scale)
        assert np.abs(be - rbe).max() <= BOXENC_TOL and np.abs(lg - rlg).max() <= LOGIT_TOL
        assert np.abs(sigmoid(lg) - sigmoid(rlg)).max() <= SCORE_TOL_PLAIN
    finally:
        e.close()


def test_fused_blocks_with_channel_groups_close_to_unfused(eng_keep, model_dir_stem_separate, head_outputs):
    """Late blocks sum fp32 partials of channel groups in a fixed order -- same values as the per-layer program up
    to the fp32 summation order, i.e. an fp16 ulp here and there, never more than 1 % of a tensor's range.  (Stem as
    its own kernel in both programs: the stem-fused program has other stem numerics, see the test above.)"""
    e = _keep_engine(model_dir_stem_separate)
    try:
        a, b = _compare_programs(eng_keep, e, head_outputs[0], 0.01)
        assert np.abs(a[1] - b[1]).max() <= 0.02 and np.abs(a[0] - b[0]).max() <= 0.02
    finally:
        e.close()


@pytest.fixture(scope="module")
def nosplit_outputs(model_dir_stem_separate, head_outputs):
    os.environ["WZ_SPLITK"] = "0"
    try:
        e = make_engine(model_dir_stem_separate, max_batch=2)
    finally:
        os.environ.pop("WZ_SPLITK")
    try:
        return e.stage_forward(head_outputs[0])
    finally:
        e.close()


@pytest.mark.parametrize("tile", [(4, 4), (5, 7), (8, 16), (16, 8), (3, 19)])
def test_fused_blocks_any_tile_shape(model_dir_stem_separate, nosplit_outputs, head_outputs, tile):
    """The workgroup tile is a tuning knob: results may not depend on it (channel groups off, so that the
    fp32 summation order is the same)."""
    env = dict(WZ_MB_TH=str(tile[0]), WZ_MB_TW=str(tile[1]), WZ_SPLITK="0", WZ_MB_WAVE="2")   # workgroup-per-tile kernel everywhere
    os.environ.update(env)
    try:
        e = make_engine(model_dir_stem_separate, max_batch=2)
        try:
            got = e.stage_forward(head_outputs[0])
        finally:
            e.close()
    finally:
        for k in env:
            os.environ.pop(k)
    np.testing.assert_array_equal(got[0], nosplit_outputs[0])
    np.testing.assert_array_equal(got[1], nosplit_outputs[1])


def test_scores_within_tolerance(eng, head_outputs):
    from oracle.postprocess import sigmoid
    x_half, rbe, rlg, _ = head_outputs
    be, lg = eng.stage_forward(x_half)
    assert np.abs(sigmoid(lg) - sigmoid(rlg)).max() <= SCORE_TOL


SCORE_TOL_FP32 = 1e-3   # the north-star's bar ("box scores within 1e-3 of the CPU reference"); measured ~1e-5


def test_fp32_engine_meets_the_north_star_tolerance(model_dir_fp32, head_outputs, synth_weights, frames_640):
    """`-p 32` engine (fp32 storage, v_mfma_f32_16x16x4_f32): every layer, every score and the end-to-end rows."""
    from oracle.postprocess import sigmoid
    from watsor_amd.detection.hip_gpu import HipObjectDetector
    from watsor_amd.share import DetectionArray
    x_half, rbe, rlg, T = head_outputs
    e = _keep_engine(model_dir_fp32)
    try:
        assert e.precision == 32
        be, lg = e.stage_forward(x_half)
        for idx, (name, h, w, c) in enumerate(e.tensors()):
            if name == "input":
                continue
            got = np.stack([e.stage_read_tensor(idx, f) for f in range(2)])
            assert got.dtype == np.float32
            err, scale = np.abs(got - T[name]).max(), np.abs(T[name]).max()
            assert err <= 2e-4 * scale + 1e-5, "%s: max abs err %.3g (max|ref| %.3g)" % (name, err, scale)
        assert np.abs(sigmoid(lg) - sigmoid(rlg)).max() <= SCORE_TOL_FP32 / 10
        assert np.abs(be - rbe).max() <= 1e-3
    finally:
        e.close()
    oracle = odet.OracleObjectDetector(weights=synth_weights)
    with HipObjectDetector(model_dir_fp32, 0) as det:
        for f in frames_640[:2]:
            rows = DetectionArray()
            det.detect(f.shape, f, rows)
            got = np.frombuffer(rows, dtype=ROW_DTYPE)
            b, c, s, _, _ = oracle.raw(f)
            ref = odet.rows_as_array(f.shape, b, c, s)
            pairs, missing = pu.match_rows(got, ref, min_score=0.0)
            assert len(missing) <= 1, missing            # (a tie at the top-100 cut may swap one row)
            assert max(abs(p[3]) for p in pairs) <= SCORE_TOL_FP32


def test_buffer_sharing_changes_nothing(eng, eng_keep_fused, head_outputs):
    x_half = head_outputs[0]
    a = eng.stage_forward(x_half)
    b = eng_keep_fused.stage_forward(x_half)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_forward_is_deterministic_and_batch_invariant(eng, head_outputs):
    x_half = head_outputs[0]
    a = eng.stage_forward(x_half)
    b = eng.stage_forward(x_half)
    np.testing.assert_array_equal(a[1], b[1])
    x8 = np.concatenate([x_half] * 4)          # batch 8: other split-K / tiling choices, same math per frame
    c = eng.stage_forward(x8)
    np.testing.assert_allclose(c[1][:2], a[1], rtol=0, atol=2e-2)
    np.testing.assert_array_equal(c[1][0], c[1][2])


def _check_post(eng, be, lg):
    B, S, C, N = eng.stage_postprocess(be, lg)
    rB, rS, rC, rN = pu.oracle_postprocess(be, lg)
    np.testing.assert_array_equal(N, rN)
    np.testing.assert_array_equal(C, rC)
    np.testing.assert_allclose(S, rS, rtol=0, atol=1e-6)
    np.testing.assert_allclose(B, rB, rtol=0, atol=2e-6)
    return N


def test_postprocess_matches_literal_per_class_nms(eng, head_outputs):
    _, rbe, rlg, _ = head_outputs
    n = _check_post(eng, rbe, rlg)
    assert (n == 100).all()


def test_postprocess_dense_scores_force_suppression(eng, head_outputs):
    """Shift logits up so thousands of candidates tie for the top and NMS has to suppress a lot."""
    _, rbe, rlg, _ = head_outputs
    _check_post(eng, rbe[:1] * 0.2, rlg[:1] + 4.0)


def test_postprocess_all_equal_logits_takes_exact_slow_path(eng, head_outputs):
    _, rbe, rlg, _ = head_outputs
    _check_post(eng, np.zeros_like(rbe[:1]), np.zeros_like(rlg[:1]))


def test_postprocess_empty_and_sparse(eng, head_outputs):
    _, rbe, rlg, _ = head_outputs
    lg = np.full_like(rlg[:1], -60.0)           # sigmoid < 1e-8 everywhere -> nothing survives the score filter
    B, S, C, N = eng.stage_postprocess(rbe[:1], lg)
    assert N[0] == 0 and not S.any() and not B.any() and (C == 1).all()   # zero padding carries label offset 1
    lg[0, 5, 3] = 2.0
    lg[0, 900, 17] = 1.0
    lg[0, 901, 17] = 0.5
    n = _check_post(eng, rbe[:1], lg)
    assert 1 <= n[0] <= 3


def test_postprocess_zero_area_boxes_are_dropped(eng, head_outputs):
    _, rbe, rlg, _ = head_outputs
    be = rbe[:1].copy()
    be[0, :, 2] = -80.0                          # exp(-16) * ha: height collapses; after clipping area may be 0
    be[0, :, 0] = 200.0                          # centre pushed far below the window -> clipped to a line
    B, S, C, N = eng.stage_postprocess(be, rlg[:1])
    rB, rS, rC, rN = pu.oracle_postprocess(be, rlg[:1])
    assert N[0] == rN[0] == 0


@pytest.mark.parametrize("wh", [(640, 480), (1280, 720), (1920, 1080), (1, 1)])
def test_row_fill_bit_exact(eng, head_outputs, wh):
    _, rbe, rlg, _ = head_outputs
    rB, rS, rC, rN = pu.oracle_postprocess(rbe[:1], rlg[:1])
    rB = rB[0].copy()
    rB[7] = (0.0, 0.0, 1.0, 1.0)                 # exact corners
    rB[8] = np.nextafter(np.float32(1.0), np.float32(0.0))
    rows = eng.stage_rows(wh[0], wh[1], rB, rS[0], rC[0])
    ref = odet.rows_as_array((wh[1], wh[0], 3), rB, rC[0], rS[0])
    np.testing.assert_array_equal(rows["label"], ref["label"])
    np.testing.assert_array_equal(rows["confidence"], ref["confidence"])
    np.testing.assert_array_equal(np.stack([rows["x_min"], rows["y_min"], rows["x_max"], rows["y_max"]], 1), ref["box"])
    assert not rows["zones"].any() and not rows["_pad"].any()


def test_detect_end_to_end_matches_oracle_detector(model_dir, synth_weights, frames_640):
    """`detect()` through the plugin class on full-resolution frames vs the oracle plugin: the north star's tolerance
    (|dscore| <= 1e-3 on every matched detection row, boxes within `box_tolerance_px`, no unexplained row) on 9 frames over
    640x480 / 1280x720 / 1920x1080, for the engine `bench.py` times."""
    from watsor_amd.detection.hip_gpu import HipObjectDetector
    from watsor_amd.share import DetectionArray
    oracle = odet.OracleObjectDetector(weights=synth_weights)
    frames = list(frames_640[:3]) + [synthetic_frame(1280, 720, 2000 + i) for i in range(3)] + \
        [synthetic_frame(1920, 1080, 3000 + i) for i in range(3)]
    from oracle.compare import assert_rows_match
    worst_px = {}
    with HipObjectDetector(model_dir, 0) as det:
        assert "gfx950" in det.device_name or "MI3" in det.device_name
        for f in frames:
            rows = DetectionArray()
            ms = det.detect(f.shape, f, rows)
            assert ms > 0
            ref_rows = DetectionArray()
            oracle.detect(f.shape, f, ref_rows)
            got = np.frombuffer(rows, dtype=ROW_DTYPE)
            b, c, s, _, _ = oracle.raw(f)
            ref = odet.rows_as_array(f.shape, b, c, s)
            # all 100 rows: scores within 1e-3, every box coordinate within the stated pixel tolerance of the frame size, and every
            # row without a partner explained by the score tolerance itself (top-100 cut / NMS tie) -- oracle/compare.py
            r = assert_rows_match(got, ref, f.shape, tol=SCORE_TOL, what="%dx%d" % (f.shape[1], f.shape[0]))
            assert len(r["pairs"]) >= 90
            worst_px[f.shape[1]] = max(worst_px.get(f.shape[1], 0), r["max_dbox_px"])
            assert got["label"][0] == ref_rows[0].label
    print("\nmax |dbox| by frame width: %s px" % worst_px)


@pytest.mark.parametrize("robust", [False, True], ids=["default", "robust"])
@pytest.mark.parametrize("seed", [77, 4242])
def test_score_tolerance_holds_for_other_weights(tmp_path, seed, robust):
    """The 1e-3 on the scores is a property of the mixed-precision program, not of one set of weights: the error budget behind it
    (tools/err_budget.py) was worked out on the seed the other tests use.  Two more networks (same architecture, other seeded
    weights), three frames each: `detect()` vs the oracle on those weights."""
    from watsor_amd import engine
    from watsor_amd.detection.hip_gpu import HipObjectDetector
    from watsor_amd.share import DetectionArray
    from watsor_amd.synth import synthetic_weights
    w = synthetic_weights(seed)
    engine.save_engine(engine.build_engine(w, robust=robust), str(tmp_path / "mi355x.bin"))
    oracle = odet.OracleObjectDetector(weights=w)
    frames = [synthetic_frame(640, 480, seed + 1), synthetic_frame(1280, 720, seed + 2), synthetic_frame(1920, 1080, seed + 3)]
    worst = 0.0
    with HipObjectDetector(str(tmp_path), 0) as det:
        for f in frames:
            rows = DetectionArray()
            det.detect(f.shape, f, rows)
            got = np.frombuffer(rows, dtype=ROW_DTYPE)
            b, c, s, _, _ = oracle.raw(f)
            ref = odet.rows_as_array(f.shape, b, c, s)
            every, _ = pu.match_rows(got, ref, min_score=0.0)
            assert len(every) >= 90
            worst = max([worst] + [abs(p[3]) for p in every])
    assert worst <= SCORE_TOL, worst


def test_batch_of_mixed_resolutions_equals_single_calls(eng):
    frames = [synthetic_frame(640, 480, 1), synthetic_frame(1920, 1080, 2), synthetic_frame(1280, 720, 3),
              synthetic_frame(640, 480, 4)]
    single = []
    for f in frames:
        r = np.zeros(100, ROW_DTYPE)
        eng.detect_batch([f], [r])
        single.append(r)
    batch = [np.zeros(100, ROW_DTYPE) for _ in frames]
    eng.detect_batch(frames, batch)
    for a, b in zip(single, batch):
        pairs, missing = pu.match_rows(b, {"label": a["label"], "confidence": a["confidence"],
                                           "box": np.stack([a["x_min"], a["y_min"], a["x_max"], a["y_max"]], 1)},
                                       min_score=0.1)
        # a batch of 4 and a batch of 1 pick different channel-group / split-K counts (fp32 summation
        # order), i.e. scores move by ~1e-4: the rows at the top-100 cut may swap with the 101st candidate
        assert all(m >= 90 for m in missing) and len(missing) <= 2, missing
        assert max(abs(p[3]) for p in pairs) <= SCORE_TOL


def test_batch_of_sixteen_cameras_stays_within_the_tolerance(model_dir, synth_weights):
    """BASELINE configs[4]'s share of a GPU: 16 frames in one batch, 640x480 and 1920x1080 alternating.  Every frame's rows against
    the ORACLE (not against another batch size): the tolerance holds at the batch size the saturation configs use."""
    oracle = odet.OracleObjectDetector(weights=synth_weights)
    e = make_engine(model_dir, max_batch=16)
    try:
        frames = [synthetic_frame(*((640, 480) if i % 2 == 0 else (1920, 1080)), 8100 + i) for i in range(16)]
        rows = [np.zeros(100, ROW_DTYPE) for _ in frames]
        e.detect_batch(frames, rows)
        for i in (0, 1, 6, 11, 15):                                  # (the oracle takes ~2 s per frame)
            b, c, s, _, _ = oracle.raw(frames[i])
            ref = odet.rows_as_array(frames[i].shape, b, c, s)
            every, _ = pu.match_rows(rows[i], ref, min_score=0.0)
            assert len(every) >= 90 and max(abs(p[3]) for p in every) <= SCORE_TOL
    finally:
        e.close()


def test_submit_host_equals_detect_batch(eng):
    """Asynchronous host-frame path (per-lane staging, optional page-locking) == the synchronous plugin call."""
    frames = [synthetic_frame(640, 480, 11 + i) for i in range(4)]
    ref = [np.zeros(100, ROW_DTYPE) for _ in frames]
    eng.detect_batch(frames, ref)
    arena = np.stack(frames)                                   # one contiguous "FrameBuffer arena"
    eng.host_register(arena)
    try:
        views = [arena[i] for i in range(4)]
        for slot in range(min(3, eng.num_slots)):              # several lanes in flight
            eng.submit_host(slot, views)
        for slot in range(min(3, eng.num_slots)):
            got = [np.zeros(100, ROW_DTYPE) for _ in frames]
            eng.collect(slot, got)
            for a, b in zip(ref, got):
                np.testing.assert_array_equal(a, b)
    finally:
        eng.sync()
        eng.host_unregister(arena)


def _engine_with(model_dir, max_batch=8, **env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return make_engine(model_dir, max_batch=max_batch)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("knob", ["WZ_HEAD_INLINE=1", "WZ_DEFER_HEADS=0", "WZ_FUSE_DECODE=0"])
def test_grouped_head_launches_are_bit_identical_to_separate_ones(eng, model_dir, knob):
    """The launch-count work on the SSD heads must not change a bit: `WZ_HEAD_INLINE=1` moves the split-K reduction of the
    heads into the head convolutions (each tile's last K slice sums the slices itself, in the same order; measured slower,
    hence off by default), `WZ_DEFER_HEADS=0` runs every head and its split-K
    reduction as launches of their own (same split counts, same summation order), `WZ_FUSE_DECODE=0` keeps the box
    decode in wz_k_decode instead of the grouped reduce (same arithmetic, compiled without contraction).
    The first two are variants of the 128 x 128 tile kernel's launches (`WZ_CONV_WIDE=0`; the wide tile kernel of the default
    program cuts K into other slices, i.e. another fp32 summation order: see the test below)."""
    frames = [synthetic_frame(640, 480, 300 + i) for i in range(4)]
    name, value = knob.split("=")
    narrow = name != "WZ_FUSE_DECODE"
    base = _engine_with(model_dir, WZ_CONV_WIDE="0") if narrow else eng
    other = _engine_with(model_dir, **({name: value, "WZ_CONV_WIDE": "0"} if narrow else {name: value}))
    try:
        ref = [np.zeros(100, ROW_DTYPE) for _ in frames]
        base.detect_batch(frames, ref)
        got = [np.zeros(100, ROW_DTYPE) for _ in frames]
        other.detect_batch(frames, got)
        for a, b in zip(ref, got):
            assert a.tobytes() == b.tobytes()
        one = [np.zeros(100, ROW_DTYPE)]                       # batch of 1: other split counts, other group shapes
        other.detect_batch(frames[:1], one)
        ref1 = [np.zeros(100, ROW_DTYPE)]
        base.detect_batch(frames[:1], ref1)
        assert one[0].tobytes() == ref1[0].tobytes()
    finally:
        other.close()
        if narrow:
            base.close()


@pytest.mark.parametrize("batch", [1, 3, 8, 21])
def test_wide_head_kernel_agrees_with_the_tile_kernel(model_dir, batch):
    """The big SSD heads run on wz_k_conv_wide (128 pixels x 288 channels per workgroup, k_conv_wide.hip); `WZ_CONV_WIDE=0` gives
    them back to wz_k_conv_rs.  Same products, other K slices (fp32 summation order): the head outputs agree to fp32 rounding of a
    5 184- / 11 520-term sum, far inside the score tolerance, and the rows match."""
    frames = [synthetic_frame(640, 480, 700 + i) for i in range(batch)]
    cap = 8 if batch <= 8 else 32       # (21 frames: more workgroups than CUs, slices of unequal length, single-slice heads)
    wide = _engine_with(model_dir, cap, WZ_CONV_WIDE="1")
    narrow = _engine_with(model_dir, cap, WZ_CONV_WIDE="0")
    try:
        x = np.stack([wide.stage_preprocess(f) for f in frames])
        bw, lw = wide.stage_forward(x)
        bn, ln = narrow.stage_forward(x)
        assert np.isfinite(lw).all() and np.isfinite(bw).all()
        assert np.abs(lw - ln).max() <= 2e-4 * max(1.0, np.abs(ln).max())
        assert np.abs(bw - bn).max() <= 2e-4 * max(1.0, np.abs(bn).max())
        assert np.abs(lw - ln).max() > 0 or batch == 0            # (they ARE different programs)
        rw = [np.zeros(100, ROW_DTYPE) for _ in frames]
        rn = [np.zeros(100, ROW_DTYPE) for _ in frames]
        wide.detect_batch(frames, rw)
        narrow.detect_batch(frames, rn)
        for a, b in zip(rn, rw):
            pairs, missing = pu.match_rows(b, {"label": a["label"], "confidence": a["confidence"],
                                               "box": np.stack([a["x_min"], a["y_min"], a["x_max"], a["y_max"]], 1)},
                                           min_score=0.1)
            assert len(missing) <= 2, missing
            assert max(abs(p[3]) for p in pairs) <= 1e-4
    finally:
        wide.close()
        narrow.close()


@pytest.mark.parametrize("T", ["4", "5", "9", "200"])
def test_wide_head_kernel_with_any_slice_length(model_dir, T):
    """`WZ_WIDE_T` forces the K slice length of wz_k_conv_wide (normally chosen per launch by a cost model).  The kernel walks a
    slice two steps at a time: an odd slice ends with a phantom step whose activations are zeros (T = 5, 9, and the odd
    remainders that T = 4 leaves of BoxPredictor_0's 81 steps), a slice longer than K is the whole K loop in one workgroup per
    tile (T = 200: no split at all, partial sums still go through the grouped reduce).  All of them are the same sums in
    another order."""
    frames = [synthetic_frame(640, 480, 900 + i) for i in range(2)]
    ref = _engine_with(model_dir, WZ_CONV_WIDE="0")
    other = _engine_with(model_dir, WZ_WIDE_T=T)
    try:
        x = np.stack([ref.stage_preprocess(f) for f in frames])
        br, lr = ref.stage_forward(x)
        bo, lo = other.stage_forward(x)
        assert np.isfinite(lo).all() and np.isfinite(bo).all()
        assert np.abs(lo - lr).max() <= 2e-4 * max(1.0, np.abs(lr).max())
        assert np.abs(bo - br).max() <= 2e-4 * max(1.0, np.abs(br).max())
    finally:
        ref.close()
        other.close()


def test_product_and_development_libraries_write_the_same_rows(model_dir, eng):
    """Two builds of one source tree: libwatsor_hip.so (what ships) and libwatsor_hip_dev.so (what the stage tests above drive)."""
    prod = conftest.make_engine(model_dir, dev=False)
    try:
        assert not prod.dev and eng.dev
        with pytest.raises(AttributeError):
            prod.stage_forward(np.zeros((1, 300, 300, 4), np.float16))
        frames = [synthetic_frame(640, 480, 3100 + i) for i in range(8)] + [synthetic_frame(1280, 720, 3200)]
        for batch in (frames[:8], frames[8:], frames[2:5]):
            a = [np.zeros(100, ROW_DTYPE) for _ in batch]
            b = [np.zeros(100, ROW_DTYPE) for _ in batch]
            prod.detect_batch(batch, a)
            eng.detect_batch(batch, b)
            for x, y in zip(a, b):
                assert x.tobytes() == y.tobytes() and x["label"][0] >= 1
    finally:
        prod.close()


def test_limits_and_errors(model_dir):
    e = conftest.make_engine(model_dir, max_batch=2, max_width=640, max_height=480)
    try:
        f = synthetic_frame(640, 480, 5)
        rows = [np.zeros(100, ROW_DTYPE) for _ in range(3)]
        with pytest.raises(ValueError):
            e.detect_batch([f, f, f], rows)                    # batch > max_batch
        with pytest.raises(ValueError):
            e.detect_batch([synthetic_frame(1280, 720, 6)], rows[:1])   # frame larger than reserved
        e.detect_batch([f, f], rows[:2])
        assert rows[0]["label"][0] >= 1
    finally:
        e.close()
    with pytest.raises(FileNotFoundError):
        from watsor_amd.detection.hip_gpu import HipObjectDetector
        HipObjectDetector(os.path.join(model_dir, "nope"), 0)


def test_postprocess_with_binding_per_class_cap(synth_weights, tmp_path, head_outputs):
    """max_detections_per_class < max_total_detections: the per-class cap decides which rows survive."""
    from watsor_amd import engine
    engine.save_engine(engine.build_engine(synth_weights, post=dict(max_per_class=3)), str(tmp_path / "mi355x.bin"))
    e = make_engine(str(tmp_path), max_batch=2)
    try:
        _, rbe, rlg, _ = head_outputs
        B, S, C, N = e.stage_postprocess(rbe, rlg)
        rB, rS, rC, rN = pu.oracle_postprocess(rbe, rlg, max_per_class=3)
        np.testing.assert_array_equal(N, rN)
        np.testing.assert_array_equal(C, rC)
        np.testing.assert_allclose(S, rS, rtol=0, atol=1e-6)
        np.testing.assert_allclose(B, rB, rtol=0, atol=2e-6)
        assert np.bincount(C[0][:N[0]]).max() == 3
    finally:
        e.close()


def trained_like_head_outputs(seed, n_objects=14, saturated=3):
    """Head outputs shaped like a TRAINED SSD's rather than the random-init network's: every class logit well above the
    graph's 1e-8 score threshold (all 1917 x 90 candidates enter per-class NMS), a few hundred scores above 0.3 in tight
    clusters of overlapping same-class anchors around each "object", several saturated logits per object (sigmoid == 1.0
    exactly in fp32: ties that only the anchor index breaks) and duplicated logit values elsewhere."""
    rng = np.random.default_rng(seed)
    anchors = pu.anchors_cs()                                      # (ycenter, xcenter, h, w)
    A, C = anchors.shape[0], 91
    lg = rng.normal(-7.0, 1.5, (A, C)).astype(np.float32)
    lg[:, 0] = rng.normal(4.0, 1.0, A)                             # background column: dropped before NMS
    be = rng.normal(0.0, 0.6, (A, 4)).astype(np.float32)
    for _ in range(n_objects):
        cls = int(rng.integers(1, C))
        cy, cx = rng.random(2)
        d = np.hypot(anchors[:, 0] - cy, anchors[:, 1] - cx)
        near = d < 0.12
        lg[near, cls] = (5.0 - 45.0 * d[near] + rng.normal(0, 0.3, int(near.sum()))).astype(np.float32)
        lg[np.argsort(d)[:saturated], cls] = 20.0                  # sigmoid(20) == 1.0f: exact ties
        dup = np.nonzero(near)[0][:6]
        lg[dup, cls] = np.float32(1.25)                            # equal unsaturated scores, too
    return be[None], lg[None]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_postprocess_on_trained_like_score_distributions(eng, seed):
    """SURVEY App. B.5 on realistic inputs: candidate selection (hint-driven bands of wz_k_nms), ties, heavy same-class
    suppression -- same rows as the literal class-by-class algorithm."""
    from oracle.postprocess import sigmoid
    be, lg = trained_like_head_outputs(seed)
    s = sigmoid(lg)[..., 1:]
    assert (s > 1e-8).all() and 100 < (s > 0.3).sum() < 2000 and (s == 1.0).sum() >= 20
    n = _check_post(eng, be, lg)
    assert n[0] == 100


def test_postprocess_band_hint_survives_scene_cuts(eng, head_outputs):
    """The first score band of a frame slot starts where it started last time (the hint): a slot that sees a busy scene,
    then an almost empty one, then the busy one again must give the oracle's rows every time."""
    _, rbe, rlg, _ = head_outputs
    busy = trained_like_head_outputs(7)
    quiet_lg = np.full_like(rlg[:1], -12.0)
    quiet_lg[0, 10:14, 5] = (-3.0, -3.5, -4.0, -9.0)               # four low scores, everything else ~6e-6
    quiet = (rbe[:1], quiet_lg)
    dense = (rbe[:1] * 0.2, rlg[:1] + 4.0)                         # thousands of candidates tie near the top
    for be, lg in (busy, quiet, busy, dense, quiet, (rbe[:1], rlg[:1]), busy):
        _check_post(eng, be, lg)


def test_inline_head_reduction_under_load(eng, model_dir):
    """The in-launch reduction of the SSD heads is a cross-workgroup hand-off (write-through slabs, ticket, acquire): run it
    the way it fails when it is wrong -- four lanes in flight, two alternating scenes so that every slab and every cached
    line holds last step's values of the OTHER scene, hundreds of steps, every row of every frame compared with the rows of
    an engine that reduces in a launch of its own (the tile kernel's K slices: the wide kernel's are cut for half the chip when
    several lanes are in flight, another summation order)."""
    os.environ.update(WZ_HEAD_INLINE="1")
    try:
        inl = make_engine(model_dir)
        os.environ.pop("WZ_HEAD_INLINE")
        os.environ.update(WZ_CONV_WIDE="0")
        ref_eng = make_engine(model_dir)
    finally:
        os.environ.pop("WZ_HEAD_INLINE", None)
        os.environ.pop("WZ_CONV_WIDE", None)
    eng = inl
    scenes = [[synthetic_frame(640, 480, 5000 + 100 * s + i) for i in range(8)] for s in range(2)]
    try:
        refs = []
        for sc in scenes:
            r = [np.zeros(100, ROW_DTYPE) for _ in sc]
            ref_eng.detect_batch(sc, r)
            refs.append(np.stack(r))
    except Exception:
        inl.close()
        raise
    finally:
        ref_eng.close()
    assert refs[0].tobytes() != refs[1].tobytes()
    dev = [[eng.upload(f) for f in sc] for sc in scenes]
    ws, hs = [640] * 8, [480] * 8
    lanes = eng.num_slots
    pending = {}
    bad = 0
    for step in range(240):
        lane = step % lanes
        if lane in pending:
            eng.wait(lane)
            bad += int(eng.slot_rows(lane, 8).tobytes() != refs[pending[lane]].tobytes())
        which = (step // 3 + step) % 2 if step % 7 else 1 - (step % 2)       # an irregular alternation
        eng.submit_device(lane, dev[which], ws, hs)
        pending[lane] = which
    for lane, which in pending.items():
        eng.wait(lane)
        bad += int(eng.slot_rows(lane, 8).tobytes() != refs[which].tobytes())
    inl.close()
    assert bad == 0
