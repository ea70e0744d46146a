"""`oracle.postprocess.multiclass_nms_global_order` (what bench.py's cpu_baseline leg times) == the literal class-by-class
`multiclass_nms` (what checks the GPU path): same boxes, scores, classes and count, bit for bit, on the network's own head
outputs, on trained-like score distributions with ties, and on the edge cases."""
import numpy as np
import pytest

import parity_utils as pu
from oracle import postprocess as post


def both(be, lg, **kw):
    a = post.postprocess(be, lg, pu.anchors_cs(), **kw)
    b = post.postprocess(be, lg, pu.anchors_cs(), fast=True, **kw)
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_array_equal(x, y)
    assert a[3] == b[3]
    return a[3]


def trained_like(seed, n_objects=14):
    rng = np.random.default_rng(seed)
    anchors = pu.anchors_cs()
    A, C = anchors.shape[0], 91
    lg = rng.normal(-7.0, 1.5, (A, C)).astype(np.float32)
    be = rng.normal(0.0, 0.6, (A, 4)).astype(np.float32)
    for _ in range(n_objects):
        cls = int(rng.integers(1, C))
        cy, cx = rng.random(2)
        d = np.hypot(anchors[:, 0] - cy, anchors[:, 1] - cx)
        near = d < 0.12
        lg[near, cls] = (5.0 - 45.0 * d[near] + rng.normal(0, 0.3, int(near.sum()))).astype(np.float32)
        lg[np.argsort(d)[:3], cls] = 20.0
        lg[np.nonzero(near)[0][:6], cls] = np.float32(1.25)
    return be, lg


def test_network_head_outputs(oracle_net, frames_640):
    x = pu.oracle_input_half(frames_640[:2])
    be, lg, _ = pu.oracle_forward_from_half(oracle_net, x)
    for i in range(2):
        assert both(be[i], lg[i]) == 100


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_trained_like_distributions_with_ties(seed):
    be, lg = trained_like(seed)
    assert both(be, lg) == 100
    assert both(be, lg, max_per_class=3) > 0


def test_edge_cases():
    A = pu.anchors_cs().shape[0]
    z = np.zeros((A, 91), np.float32)
    assert both(np.zeros((A, 4), np.float32), z) == 100                      # all scores equal: pure tie order
    assert both(np.zeros((A, 4), np.float32), np.full((A, 91), -60.0, np.float32)) == 0
    be = np.zeros((A, 4), np.float32)
    be[:, 2] = -80.0
    be[:, 0] = 200.0
    assert both(be, z) == 0                                                   # zero-area boxes
    lg = np.full((A, 91), -60.0, np.float32)
    lg[5, 3], lg[900, 17], lg[901, 17] = 2.0, 1.0, 0.5
    assert 1 <= both(np.zeros((A, 4), np.float32), lg) <= 3
