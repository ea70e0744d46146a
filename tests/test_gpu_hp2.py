"""The 10x10 split blocks of the robust program (expanded_conv_13 .. 16) as TWO launches per block (csrc/k_mbconv_hp2.hip: expand +
depthwise per band and chunk -> project fragments in the workspace; then a plain split-operand GEMM over the batch's pixels) -- the form
the engine picks from four frames up under the throughput schedule.  Reference anchor: these are layers inside `sess.run` of
watsor/detection/tensorflow_cpu.py:113-115; the checker is the fp32 oracle (oracle/ssd_mobilenet_v2.py), tensor by tensor and row by row.

Tolerances: as in tests/test_gpu_parity.py::test_both_programs_close_to_oracle_tensor_by_tensor -- pair tensors 5e-4 of the range,
the two one-rounding tensors 1.5e-3, scores 1e-3; rows against the oracle through oracle/compare.py (1e-3, stated pixel tolerance,
every unmatched row explained); run-to-run bit-identical.
"""
import os

import numpy as np
import pytest

import conftest
import parity_utils as pu
from oracle import detect as odet
from oracle.compare import assert_rows_match
from watsor_amd.runtime import ROW_DTYPE
from watsor_amd.synth import synthetic_frame

pytestmark = pytest.mark.gpu

LATE = ("expanded_conv_13/expand", "expanded_conv_13/output", "expanded_conv_14/output", "expanded_conv_15/output", "expanded_conv_16/output")


def _forward_tensors(model_dir, x_half, min_n, monkeypatch):
    """stage_forward on a buffer-keeping development engine with WZ_HP2_MIN_N = min_n: {tensor name: [2, h, w, c] float32}, (box enc, logits)."""
    monkeypatch.setenv("WZ_NO_BUFFER_REUSE", "1")
    monkeypatch.setenv("WZ_HP2_MIN_N", str(min_n))
    e = conftest.make_engine(model_dir, max_batch=2, dev=True)
    try:
        be, lg = e.stage_forward(x_half)
        out = {}
        for idx, (name, h, w, c) in enumerate(e.tensors()):
            if name in LATE:
                out[name] = (np.stack([e.stage_read_tensor(idx, f) for f in range(2)]).astype(np.float32), e.tensor_is_pair(idx))
        return out, be, lg
    finally:
        e.close()


def test_two_launch_blocks_close_to_oracle_tensor_by_tensor(model_dir_robust, oracle_net, frames_640, monkeypatch):
    """Blocks 13 .. 16 forced onto the two-launch form at batch 2 (WZ_HP2_MIN_N=1): their outputs and the first SSD feature map against
    the fp32 oracle at the bounds of the one-launch form, and against the one-launch form itself (same rounding points: the tensors
    agree to a few fp32 summation-order ulps of the pair, i.e. far inside the oracle bound)."""
    x_half = pu.oracle_input_half(frames_640[:2])
    rbe, rlg, T = pu.oracle_forward_from_half(oracle_net, x_half, keep=True)
    two, be2, lg2 = _forward_tensors(model_dir_robust, x_half, 1, monkeypatch)
    one, be1, lg1 = _forward_tensors(model_dir_robust, x_half, 0, monkeypatch)
    assert set(two) == set(LATE)
    for name in LATE:
        got, pair = two[name]
        ref = T[name]
        if name == "expanded_conv_16/output":
            c = got.shape[-1]
            np.testing.assert_array_equal(got[..., :c // 2], got[..., c // 2:])
            got = got[..., :c // 2]
        err, scale = np.abs(got - ref).max(), np.abs(ref).max()
        bound = (5e-4 if pair else 1.5e-3) * scale + 1e-4
        assert err <= bound, "%s (%s): max abs err %.3g (max|ref| %.3f)" % (name, "pair" if pair else "one rounding", err, scale)
        old = one[name][0][..., :got.shape[-1]]
        # (pair tensors: a few fp32 ulps of the sums; block 16's plain fp16 output: the roundings may land on neighbouring halves)
        assert np.abs(got - old).max() <= (2e-4 * scale + 1e-5 if pair else scale * 2.0 ** -10), "%s: two-launch vs one-launch %.3g" % (name, np.abs(got - old).max())
    # the expanded tensor of block 13 (the first SSD feature map) is computed by the same instructions in both forms
    np.testing.assert_array_equal(two["expanded_conv_13/expand"][0], one["expanded_conv_13/expand"][0])
    from oracle.postprocess import sigmoid
    assert np.abs(sigmoid(lg2) - sigmoid(rlg)).max() <= 1e-3
    assert np.abs(be2 - rbe).max() <= 0.01 and np.abs(lg2 - rlg).max() <= 0.0125


@pytest.mark.parametrize("batch", [4, 5, 7, 8])
def test_two_launch_blocks_end_to_end_any_batch(model_dir_robust, synth_weights, batch):
    """The PRODUCT library at the batch sizes where the form is picked by default, including pixel counts that are no multiple of 16
    (5 frames = 500 pixels = 31.25 tiles: the last project tile is part garbage, which must stay in its own columns): rows against the
    oracle, bit-identical on repeat, and the frames' rows do not depend on their position in the batch."""
    frames = [synthetic_frame(640, 480, 6100 + i) for i in range(batch)]
    oracle = odet.OracleObjectDetector(weights=synth_weights)
    eng = conftest.make_engine(model_dir_robust, max_batch=8, max_width=640, max_height=480)
    try:
        rows = [np.zeros(100, ROW_DTYPE) for _ in frames]
        eng.detect_batch(frames, rows)
        for i in (0, batch - 1):
            f = frames[i]
            b, c, s, _, _ = oracle.raw(f)
            r = assert_rows_match(rows[i], odet.rows_as_array(f.shape, b, c, s), f.shape, tol=1e-3, what="two-launch blocks, batch %d, frame %d" % (batch, i))
            assert len(r["pairs"]) >= 90
        again = [np.zeros(100, ROW_DTYPE) for _ in frames]
        eng.detect_batch(frames, again)
        for a, g in zip(rows, again):
            assert a.tobytes() == g.tobytes()
        rev = [np.zeros(100, ROW_DTYPE) for _ in frames]
        eng.detect_batch(frames[::-1], rev)
        for a, g in zip(rows, rev[::-1]):
            assert a.tobytes() == g.tobytes()           # a frame's pixels never mix with its neighbours' (partial tiles at frame seams)
    finally:
        eng.close()


def test_two_launch_form_is_what_runs_at_batch_8(model_dir_robust):
    """Graph of a batch of 8 on the development library: 4 more kernel nodes than with the form switched off (one extra launch per block)."""
    import json
    import subprocess
    import sys
    script = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from watsor_amd.runtime import HipEngine, ROW_DTYPE\n"
        "from watsor_amd.synth import synthetic_frame\n"
        "e = HipEngine(%r, 0, 8, 640, 480, dev=True)\n"
        "frames = [synthetic_frame(640, 480, 6200 + i) for i in range(8)]\n"
        "rows = [np.zeros(100, ROW_DTYPE) for _ in frames]\n"
        "e.detect_batch(frames, rows)\n"
        "print(json.dumps(dict(nodes=e.graph_nodes(0), conf=[r['confidence'].tolist() for r in rows])))\n"
        "e.close()\n" % (conftest.ROOT, os.path.join(model_dir_robust, "mi355x.bin")))
    out = {}
    for name, env in (("two", {}), ("one", {"WZ_HP2_MIN_N": "0"})):
        p = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, WZ_GRAPH="1", **env), capture_output=True, text=True, timeout=240)
        assert p.returncode == 0, p.stderr[-1500:]
        out[name] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["two"]["nodes"] == out["one"]["nodes"] + 4
    a, b = np.array(out["two"]["conf"]), np.array(out["one"]["conf"])
    assert np.abs(np.sort(a, axis=1) - np.sort(b, axis=1)).max() <= 5e-4
