"""The 1x1 + 3x3 stride-2 pairs of the SSD extras chain in ONE launch (csrc/k_extras_pair.hip; layer_19_1_Conv2d_{3,4,5}_1x1_* and
layer_19_2_Conv2d_{3,4,5}_3x3_s2_*: layers inside `sess.run` of watsor/detection/tensorflow_cpu.py:113-115) against the fp32 oracle and
against the two-launch form of the same layers (development knob WZ_EXTRAS_PAIR=0).

Tolerances: each tensor against the oracle at the bound of tests/test_gpu_parity.py::test_every_layer_close_to_oracle (0.04 of the
range + 0.02); fused against the two launches within one fp16 step of the range for the 1x1 (the two forms sum K in different orders: a
value on a rounding boundary may land on either side) and four steps for the 3x3 that reads it; scores 1e-3.
"""
import numpy as np
import pytest

import conftest
import parity_utils as pu

pytestmark = pytest.mark.gpu

PAIRS = [("layer_19_1_Conv2d_%d_1x1_%d" % (k, c), "layer_19_2_Conv2d_%d_3x3_s2_%d" % (k, 2 * c)) for k, c in ((3, 128), (4, 128), (5, 64))]


def _extras(model_dir, x_half, monkeypatch, pair):
    monkeypatch.setenv("WZ_NO_BUFFER_REUSE", "1")
    monkeypatch.setenv("WZ_EXTRAS_PAIR", pair)
    e = conftest.make_engine(model_dir, max_batch=len(x_half), dev=True)
    try:
        be, lg = e.stage_forward(x_half)
        out = {}
        for idx, (name, h, w, c) in enumerate(e.tensors()):
            short = name.split("/")[-1]
            if short.startswith("layer_19_"):
                out[short] = np.stack([e.stage_read_tensor(idx, f) for f in range(len(x_half))]).astype(np.float32)
        return out, be, lg, {t[0].split("/")[-1]: t[0] for t in e.tensors()}
    finally:
        e.close()


def test_fused_pairs_close_to_oracle_and_to_the_two_launch_form(model_dir, oracle_net, frames_640, monkeypatch):
    d = model_dir                                           # (both `-p 16` programs: the fixture is parametrised)
    x_half = pu.oracle_input_half(frames_640[:3])          # three frames: the workgroups of a frame never read a neighbour's pixels
    rbe, rlg, T = pu.oracle_forward_from_half(oracle_net, x_half, keep=True)
    fused, be1, lg1, names = _extras(d, x_half, monkeypatch, "1")
    plain, be0, lg0, _ = _extras(d, x_half, monkeypatch, "0")
    for a, b in PAIRS:
        assert a in fused and b in fused, sorted(fused)
        for n in (a, b):
            ref = T[names[n]]
            scale = np.abs(ref).max()
            assert fused[n].shape == ref.shape
            err = np.abs(fused[n] - ref).max()
            assert err <= 0.04 * scale + 0.02, "%s: max abs err %.4g (max|ref| %.3f)" % (n, err, scale)
            dif = np.abs(fused[n] - plain[n]).max()
            assert dif <= (2.0 ** -10 if n == a else 2.0 ** -8) * scale, "%s: fused vs two launches %.4g (max|ref| %.3f)" % (n, dif, scale)
    from oracle.postprocess import sigmoid
    assert np.abs(sigmoid(lg1) - sigmoid(rlg)).max() <= 1e-3
    assert np.abs(lg1 - lg0).max() <= 4e-3 and np.abs(be1 - be0).max() <= 4e-3
    # the frames in another order: every frame's tensors bit for bit (a frame's workgroups never read a neighbour's pixels; the launches above the
    # pairs pick their shapes by the batch's pixel count, so the comparison is at the same batch size)
    order = [2, 0, 1]
    moved, _, _, _ = _extras(d, x_half[order], monkeypatch, "1")
    for a, b in PAIRS:
        for n in (a, b):
            for k, src in enumerate(order):
                np.testing.assert_array_equal(moved[n][k], fused[n][src])
