"""Independent witness for the oracle's backbone: the installed `transformers` ships a
MobileNetV2 written to load TF-slim checkpoints (TF "SAME" padding, BatchNorm eps 1e-3, ReLU6, expansion 6,
first layer without expansion).  It is not part of the reference and shares no code with oracle/: loaded
with the same seeded weights it must reproduce every inverted-residual block output, the first SSD feature
map (expanded_conv_13/expand) and Conv_1 of oracle/ssd_mobilenet_v2.py.  (The SSD extras, heads, anchors
and NMS have no such witness: oracle/postprocess.py is pinned only by its literal restatement of SURVEY
App. B.)"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

FE = "FeatureExtractor/MobilenetV2/"


def _load(module, W, scope, depthwise=False):
    """TF variables of one Conv+BatchNorm scope -> a transformers MobileNetV2ConvLayer."""
    w = W[scope + ("/depthwise_weights" if depthwise else "/weights")]
    w = w.transpose(2, 3, 0, 1) if depthwise else w.transpose(3, 2, 0, 1)        # -> OIHW
    sd = {"convolution.weight": torch.from_numpy(np.ascontiguousarray(w)),
          "normalization.weight": torch.from_numpy(W[scope + "/BatchNorm/gamma"]),
          "normalization.bias": torch.from_numpy(W[scope + "/BatchNorm/beta"]),
          "normalization.running_mean": torch.from_numpy(W[scope + "/BatchNorm/moving_mean"]),
          "normalization.running_var": torch.from_numpy(W[scope + "/BatchNorm/moving_variance"])}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)


def test_backbone_matches_transformers_mobilenet_v2(synth_weights, oracle_net):
    from transformers import MobileNetV2Config, MobileNetV2Model
    cfg = MobileNetV2Config(image_size=300, depth_multiplier=1.0, depth_divisible_by=8, min_depth=8, expand_ratio=6.0,
                            output_stride=32, first_layer_is_expansion=True, finegrained_output=True,
                            hidden_act="relu6", tf_padding=True, layer_norm_eps=0.001)
    torch.set_grad_enabled(False)
    m = MobileNetV2Model(cfg, add_pooling_layer=False).eval()
    W = synth_weights
    _load(m.conv_stem.first_conv, W, FE + "Conv")
    _load(m.conv_stem.conv_3x3, W, FE + "expanded_conv/depthwise", depthwise=True)
    _load(m.conv_stem.reduce_1x1, W, FE + "expanded_conv/project")
    assert len(m.layer) == 16
    for i, layer in enumerate(m.layer):
        s = FE + "expanded_conv_%d" % (i + 1)
        _load(layer.expand_1x1, W, s + "/expand")
        _load(layer.conv_3x3, W, s + "/depthwise", depthwise=True)
        _load(layer.reduce_1x1, W, s + "/project")
    _load(m.conv_1x1, W, FE + "Conv_1")
    tap0 = {}
    m.layer[12].expand_1x1.register_forward_hook(lambda mod, inp, out: tap0.setdefault("x", out))

    rng = np.random.Generator(np.random.PCG64(7))
    x = rng.uniform(-1.0, 1.0, (1, 300, 300, 3)).astype(np.float32)
    _, _, T = oracle_net.forward(x, keep=True)
    out = m(torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2))), output_hidden_states=True)

    def close(name, got):
        ref = T[name]
        got = got.permute(0, 2, 3, 1).numpy()
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        err = np.abs(got - ref).max()
        assert err <= 2e-4 * np.abs(ref).max() + 1e-5, "%s: max abs err %.3g (max|ref| %.3g)" % (name, err, np.abs(ref).max())

    hs = out.hidden_states                        # outputs of layer 0 .. layer 15 (the stem's is not exposed)
    assert len(hs) >= 16
    for i in range(16):
        close("expanded_conv_%d/output" % (i + 1), hs[i])
    close("expanded_conv_13/expand", tap0["x"])
    close("Conv_1", out.last_hidden_state)


def test_resize_matches_scipy_sampling_at_the_legacy_coordinates():
    """oracle/preprocess.py against scipy.ndimage.map_coordinates (an independent bilinear sampler) at the TF1
    legacy sample positions src = dst * (in / out) (align_corners=False, no half-pixel centres).  This checks
    the interpolation arithmetic and the edge clamp; that TF uses these positions is SURVEY App. B.1's claim."""
    ndi = pytest.importorskip("scipy.ndimage")
    from oracle import preprocess as pre
    rng = np.random.Generator(np.random.PCG64(3))
    for (w, h) in [(640, 480), (301, 299), (64, 48), (1280, 720)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = pre.resize_bilinear(img, 300, 300)
        ys = (np.arange(300, dtype=np.float32) * (np.float32(h) / np.float32(300))).astype(np.float64)
        xs = (np.arange(300, dtype=np.float32) * (np.float32(w) / np.float32(300))).astype(np.float64)
        yy, xx = np.meshgrid(ys, xs, indexing="ij")
        for c in range(3):
            ref = ndi.map_coordinates(img[..., c].astype(np.float64), [yy, xx], order=1, mode="nearest")
            assert np.abs(got[..., c] - ref).max() <= 2e-3      # fp32 lerps vs float64 spline evaluation on 0..255


def test_half_pixel_resize_matches_torch_interpolate():
    """The `half_pixel` variant of oracle/preprocess.py (graphs exported with `ResizeBilinear(half_pixel_centers=True)`: src =
    (dst + 0.5) * scale - 0.5, clamped at the edges) against `torch.nn.functional.interpolate(mode="bilinear", align_corners=False,
    antialias=False)` -- an independent implementation of the same coordinate rule that sits in this image (the variant
    was pinned by hand-worked known answers only).  Up- and down-scaling, odd sizes; fp32 lerps on 0 .. 255 on both sides."""
    import torch.nn.functional as F
    from oracle import preprocess as pre
    rng = np.random.Generator(np.random.PCG64(11))
    for (w, h) in [(640, 480), (301, 299), (64, 48), (1280, 720), (300, 300)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = pre.resize_bilinear(img, 300, 300, half_pixel_centers=True)
        x = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
        ref = F.interpolate(x, size=(300, 300), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        assert got.shape == ref.shape
        # (the two compute the sample position with different roundings -- TF's three fp32 roundings against torch's scale arithmetic -- which
        #  shows as a few 1e-3 on 0 .. 255 where neighbouring pixels differ by up to 255; the legacy rule is off by whole grey levels, below)
        assert np.abs(got - ref).max() <= 1e-2, (w, h, np.abs(got - ref).max())
    # ... and the legacy rule is a different function: the witness tells the two variants apart
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    ref = F.interpolate(torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None], size=(300, 300), mode="bilinear",
                        align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(pre.resize_bilinear(img, 300, 300) - ref).max() > 1.0
