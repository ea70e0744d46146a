"""The 10x10 split blocks of the robust program (expanded_conv_13 .. 16) as TWO launches per block (csrc/k_mbconv_hp2.hip: expand +
depthwise per band and chunk -> project fragments in the workspace; then a plain split-operand GEMM over the batch's pixels) -- the only
form of these blocks since late round 6 (faster than the one-launch lean builds at every batch size: profiles/r06_hp2_by_batch_size.txt).  Reference anchor: these are layers inside `sess.run` of
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


def _forward_tensors(model_dir, x_half, monkeypatch, **env):
    """stage_forward on a buffer-keeping development engine: {tensor name: ([2, h, w, c] float32, is pair)}, box encodings, logits."""
    monkeypatch.setenv("WZ_NO_BUFFER_REUSE", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
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
    """Blocks 13 .. 16 at batch 2: their outputs and the first SSD feature map against the fp32 oracle at the bounds of
    tests/test_gpu_parity.py::test_both_programs_close_to_oracle_tensor_by_tensor, with both tile shapes of launch B (one / two pixel
    tiles per workgroup) and both wave counts of launch A -- every shape sums the project stage's K in the same wave order, so the
    tensors are bit-identical across them."""
    x_half = pu.oracle_input_half(frames_640[:2])
    rbe, rlg, T = pu.oracle_forward_from_half(oracle_net, x_half, keep=True)
    two, be2, lg2 = _forward_tensors(model_dir_robust, x_half, monkeypatch)
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
    from oracle.postprocess import sigmoid
    assert np.abs(sigmoid(lg2) - sigmoid(rlg)).max() <= 1e-3
    assert np.abs(be2 - rbe).max() <= 0.01 and np.abs(lg2 - rlg).max() <= 0.0125


@pytest.mark.parametrize("batch", [1, 2, 3, 4, 5, 7, 8])     # (1, 2: the small-batch shapes of blocks 1 .. 5, csrc/k_mbconv_hp.hip)
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


def test_two_launch_form_is_what_runs(model_dir_robust):
    """The captured graph of a batch on the robust program: 31 kernel nodes -- the resize, 13 split blocks, 4 x 2 launches for blocks 13 .. 16,
    Conv_1, the extras (2 launches + 3 fused pairs, csrc/k_extras_pair.hip), the grouped heads + their reduce, the NMS -- at batch 8 and at batch 1 alike."""
    import json
    import subprocess
    import sys
    script = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from watsor_amd.runtime import HipEngine, ROW_DTYPE\n"
        "from watsor_amd.synth import synthetic_frame\n"
        "e = HipEngine(%r, 0, 8, 640, 480)\n"
        "out = {}\n"
        "for n in (8, 1):\n"
        "    frames = [synthetic_frame(640, 480, 6200 + i) for i in range(n)]\n"
        "    rows = [np.zeros(100, ROW_DTYPE) for _ in frames]\n"
        "    e.detect_batch(frames, rows)\n"
        "    out[str(n)] = e.graph_nodes(0)\n"
        "print(json.dumps(out))\n"
        "e.close()\n" % (conftest.ROOT, os.path.join(model_dir_robust, "mi355x.bin")))
    p = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, WZ_GRAPH="1"), capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-1500:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out == {"8": 31, "1": 31}, out


def test_small_batch_tile_shape_changes_no_bit(model_dir_robust, frames_640, monkeypatch):
    """One or two frames per batch: the robust program's stride-1 blocks of the 75x75 and 38x38 maps (blocks 2, 4, 5) run on 4 x 4 tiles instead of 4 x 8
    (csrc/k_mbconv_hp.hip, wz_launch_mbconv_hp_q: twice the workgroups on grids that fill a fifth of the chip).  A tile shape only decides which lane holds a
    pixel: every tensor of the network, the box encodings and the class logits are the same bit for bit with the shape forced off and on."""
    x_half = pu.oracle_input_half(frames_640[:2])
    monkeypatch.setenv("WZ_NO_BUFFER_REUSE", "1")

    def run(knob):
        monkeypatch.setenv("WZ_HP_TILES44", knob)
        e = conftest.make_engine(model_dir_robust, max_batch=2, dev=True)
        try:
            be, lg = e.stage_forward(x_half)
            tensors = {name: np.stack([e.stage_read_tensor(idx, f) for f in range(2)]) for idx, (name, h, w, c) in enumerate(e.tensors()) if name != "input"}
            return be, lg, tensors
        finally:
            e.close()

    be0, lg0, t0 = run("0")
    be1, lg1, t1 = run("1")
    assert "expanded_conv_2/output" in t0
    for k in t0:
        np.testing.assert_array_equal(t0[k].view(np.uint16), t1[k].view(np.uint16), err_msg=k)
    np.testing.assert_array_equal(be0, be1)
    np.testing.assert_array_equal(lg0, lg1)


def test_large_batch_shapes_of_the_early_blocks_tensor_by_tensor(model_dir_robust, oracle_net, frames_640, monkeypatch):
    """The tensor-by-tensor tests run two frames, and at one or two frames per batch blocks 1 .. 3 take the chunk-split shape and blocks 4 / 5 the 4 x 4 tiles
    (csrc/k_mbconv_hp.hip, wz_launch_mbconv_hp_q).  This test switches both rules off (WZ_HP_SMALL_CS=0, WZ_HP_TILES44=0: the shapes of batches >= 3, the
    benchmarked ones) and holds the outputs of blocks 0 .. 5 against the fp32 oracle at the pair-tensor bound of
    tests/test_gpu_parity.py::test_both_programs_close_to_oracle_tensor_by_tensor (5e-4 of the range), and against the small-batch shapes at a tenth of that
    (5e-5 of the range: the chunk-split shape sums a tile's chunks in wave order, an fp32 rounding that the hi + lo split and the next blocks carry on)."""
    x_half = pu.oracle_input_half(frames_640[:2])
    rbe, rlg, T = pu.oracle_forward_from_half(oracle_net, x_half, keep=True)
    monkeypatch.setenv("WZ_NO_BUFFER_REUSE", "1")
    want = ["expanded_conv/output"] + ["expanded_conv_%d/output" % i for i in range(1, 6)]

    def run(cs, t44):
        monkeypatch.setenv("WZ_HP_SMALL_CS", cs)
        monkeypatch.setenv("WZ_HP_TILES44", t44)
        e = conftest.make_engine(model_dir_robust, max_batch=2, dev=True)
        try:
            be, lg = e.stage_forward(x_half)
            out = {}
            for idx, (name, h, w, c) in enumerate(e.tensors()):
                if name in want:
                    assert e.tensor_is_pair(idx)
                    out[name] = (np.stack([e.stage_read_tensor(idx, f) for f in range(2)]).astype(np.float32), name)
            return out, be, lg
        finally:
            e.close()

    big, be_b, lg_b = run("0", "0")
    small, be_s, lg_s = run("2", "2")
    assert sorted(big) == sorted(want)
    for k in want:
        got, name = big[k]
        ref = T[name]
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 5e-4 * scale + 1e-4, k
        assert np.abs(got - small[k][0]).max() <= 5e-5 * scale, "%s: %.3g of %.3g" % (k, np.abs(got - small[k][0]).max(), scale)
    # (behind the pair tensors the network rounds to fp16 per layer: an fp32 rounding upstream moves single fp16 steps downstream)
    d_lg, d_be = np.abs(lg_b - lg_s).max(), np.abs(be_b - be_s).max()
    assert d_lg <= 6e-3 and d_be <= 6e-3, (d_lg, d_be)
    from oracle.postprocess import sigmoid
    assert np.abs(sigmoid(lg_b) - sigmoid(rlg)).max() <= 1e-3 and np.abs(sigmoid(lg_s) - sigmoid(rlg)).max() <= 1e-3
