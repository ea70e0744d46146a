"""Oracle (test infrastructure): SSD-MobileNet-v2 300x300 forward pass in fp32 on the CPU.

Restates what `sess.run` computes between `image_tensor:0` (after resize+normalise) and
the raw box encodings / class logits for the reference's CPU plugin
(`watsor/detection/tensorflow_cpu.py:94-121`).  The graph itself is not in the reference
(it is the model *file*), so this follows SURVEY.md Appendix A (layer inventory) and the
reference's own SSD config `watsor/test/model/prepare.py:19-150` (ReLU6, BN eps 1e-3,
convolutional box predictor).  PARITY UNPINNED against real TensorFlow (see oracle/__init__).

Weights are a dict keyed by the TF-slim / TF-OD-API variable names of the frozen graph
(`FeatureExtractor/MobilenetV2/expanded_conv_3/depthwise/depthwise_weights`, ...), in TF
layouts (HWIO for conv, HWC1 for depthwise), BatchNorm kept *unfolded* exactly like the
frozen graph's FusedBatchNorm nodes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

BN_EPS = 1e-3  # prepare.py:48 (epsilon: 0.0010000000475)
NUM_CLASSES_WITH_BG = 91  # 90 COCO ids + background column 0
FE = "FeatureExtractor/MobilenetV2/"

# (expansion t, out channels, stride) for expanded_conv .. expanded_conv_16  
BLOCKS: List[Tuple[int, int, int]] = [
    (1, 16, 1),
    (6, 24, 2), (6, 24, 1),
    (6, 32, 2), (6, 32, 1), (6, 32, 1),
    (6, 64, 2), (6, 64, 1), (6, 64, 1), (6, 64, 1),
    (6, 96, 1), (6, 96, 1), (6, 96, 1),
    (6, 160, 2), (6, 160, 1), (6, 160, 1),
    (6, 320, 1),
]
# SSD extras: (1x1 depth, 3x3 s2 depth), names layer_19_{1,2}_Conv2d_{i}_...
EXTRAS: List[Tuple[int, int]] = [(256, 512), (128, 256), (128, 256), (64, 128)]
ANCHORS_PER_LOC = [3, 6, 6, 6, 6, 6]
HEAD_KERNEL = 3  # ssd_mobilenet_v2_coco: convolutional_box_predictor kernel_size 3


def block_name(i: int) -> str:
    return "expanded_conv" if i == 0 else "expanded_conv_%d" % i


def extra_names(i: int) -> Tuple[str, str]:
    d1, d2 = EXTRAS[i]
    return ("layer_19_1_Conv2d_%d_1x1_%d" % (i + 2, d1), "layer_19_2_Conv2d_%d_3x3_s2_%d" % (i + 2, d2))


def same_pad(n_in: int, k: int, stride: int) -> Tuple[int, int, int]:
    """TensorFlow 'SAME' padding: returns (n_out, pad_before, pad_after)."""
    n_out = -(-n_in // stride)
    total = max((n_out - 1) * stride + k - n_in, 0)
    before = total // 2
    return n_out, before, total - before


@dataclass
class ConvSpec:
    """One convolution of the graph, in execution order (used by tests to walk layers)."""
    name: str            # TF scope, e.g. FeatureExtractor/MobilenetV2/expanded_conv_1/expand
    kind: str            # 'conv' | 'dw'
    cin: int
    cout: int
    k: int
    stride: int
    bn: bool             # conv followed by FusedBatchNorm (else `biases`)
    relu6: bool
    src: str             # tensor name consumed
    dst: str             # tensor name produced
    res: Optional[str] = None  # residual tensor added to the output (after BN, no activation)


def graph_spec() -> List[ConvSpec]:
    """Execution-ordered list of every convolution (70 ops) with tensor names."""
    ops: List[ConvSpec] = []
    ops.append(ConvSpec(FE + "Conv", "conv", 3, 32, 3, 2, True, True, "input", "Conv"))
    cur, cin = "Conv", 32
    for i, (t, cout, s) in enumerate(BLOCKS):
        bn_ = block_name(i)
        x_in = cur
        mid = cin * t
        if t != 1:
            ops.append(ConvSpec(FE + bn_ + "/expand", "conv", cin, mid, 1, 1, True, True, cur, bn_ + "/expand"))
            cur = bn_ + "/expand"
        ops.append(ConvSpec(FE + bn_ + "/depthwise", "dw", mid, mid, 3, s, True, True, cur, bn_ + "/depthwise"))
        res = x_in if (s == 1 and cin == cout) else None
        ops.append(ConvSpec(FE + bn_ + "/project", "conv", mid, cout, 1, 1, True, False,
                            bn_ + "/depthwise", bn_ + "/output", res))
        cur, cin = bn_ + "/output", cout
    ops.append(ConvSpec(FE + "Conv_1", "conv", 320, 1280, 1, 1, True, True, cur, "Conv_1"))
    cur, cin = "Conv_1", 1280
    for i, (d1, d2) in enumerate(EXTRAS):
        n1, n2 = extra_names(i)
        ops.append(ConvSpec(FE + n1, "conv", cin, d1, 1, 1, True, True, cur, n1))
        ops.append(ConvSpec(FE + n2, "conv", d1, d2, 3, 2, True, True, n1, n2))
        cur, cin = n2, d2
    taps = feature_map_names()
    tap_c = [576, 1280, 512, 256, 256, 128]
    for i, (tname, c) in enumerate(zip(taps, tap_c)):
        a = ANCHORS_PER_LOC[i]
        ops.append(ConvSpec("BoxPredictor_%d/BoxEncodingPredictor" % i, "conv", c, a * 4, HEAD_KERNEL, 1,
                            False, False, tname, "box_%d" % i))
        ops.append(ConvSpec("BoxPredictor_%d/ClassPredictor" % i, "conv", c, a * NUM_CLASSES_WITH_BG,
                            HEAD_KERNEL, 1, False, False, tname, "cls_%d" % i))
    return ops


def feature_map_names() -> List[str]:
    """SSD taps: expanded_conv_13/expand output (19x19x576), Conv_1 (10x10x1280), 4 extras."""
    return ["expanded_conv_13/expand", "Conv_1"] + [extra_names(i)[1] for i in range(4)]


def feature_map_sizes(size: int = 300) -> List[int]:
    s = size
    s = same_pad(s, 3, 2)[0]          # Conv            150
    s = same_pad(s, 3, 2)[0]          # expanded_conv_1  75
    s = same_pad(s, 3, 2)[0]          # expanded_conv_3  38
    s19 = same_pad(s, 3, 2)[0]        # expanded_conv_6  19
    s10 = same_pad(s19, 3, 2)[0]      # expanded_conv_13 10
    out = [s19, s10]
    s = s10
    for _ in range(4):
        s = same_pad(s, 3, 2)[0]
        out.append(s)
    return out                         # [19, 10, 5, 3, 2, 1]


def num_anchors(size: int = 300) -> int:
    return sum(f * f * a for f, a in zip(feature_map_sizes(size), ANCHORS_PER_LOC))  # 1917


# --------------------------------------------------------------------------------------
# forward pass (torch CPU fp32)
# --------------------------------------------------------------------------------------

class OracleNet:
    """The graph with weights converted to torch tensors once (layouts only; BatchNorm stays unfolded)."""

    def __init__(self, W: Dict[str, np.ndarray]):
        import torch

        torch.set_grad_enabled(False)
        self.spec = graph_spec()
        self.params = []
        for op in self.spec:
            if op.kind == "dw":
                w = torch.from_numpy(np.ascontiguousarray(W[op.name + "/depthwise_weights"].transpose(2, 3, 0, 1)))
            else:
                w = torch.from_numpy(np.ascontiguousarray(W[op.name + "/weights"].transpose(3, 2, 0, 1)))  # OIHW
            if op.bn:
                g = torch.from_numpy(W[op.name + "/BatchNorm/gamma"])
                b = torch.from_numpy(W[op.name + "/BatchNorm/beta"])
                m = torch.from_numpy(W[op.name + "/BatchNorm/moving_mean"])
                v = torch.from_numpy(W[op.name + "/BatchNorm/moving_variance"])
                # FusedBatchNorm (inference): (x - mean) * (gamma * rsqrt(var + eps)) + beta
                scale = (g * torch.rsqrt(v + BN_EPS))[None, :, None, None]
                self.params.append((w, m[None, :, None, None], scale, b[None, :, None, None]))
            else:
                self.params.append((w, None, None, torch.from_numpy(W[op.name + "/biases"])[None, :, None, None]))

    def forward(self, x_nhwc: np.ndarray, keep: bool = False):
        """x_nhwc: float32 [B,300,300,3] already resized + normalised.

        Returns (box_encodings [B,1917,4], class_logits [B,1917,91], tensors) where `tensors`
        (when keep=True) maps the tensor names of graph_spec() to NHWC float32 arrays.
        """
        import torch
        import torch.nn.functional as F

        T = {"input": torch.from_numpy(np.ascontiguousarray(x_nhwc.transpose(0, 3, 1, 2)))}
        boxes, logits = [], []
        for op, (w, mean, scale, beta) in zip(self.spec, self.params):
            x = T[op.src]
            _, pt, pb = same_pad(x.shape[2], op.k, op.stride)
            _, pl, pr = same_pad(x.shape[3], op.k, op.stride)
            if pt or pb or pl or pr:
                x = F.pad(x, (pl, pr, pt, pb))        # TF 'SAME': the extra pixel goes after
            y = F.conv2d(x, w, None, stride=op.stride, padding=0, groups=op.cin if op.kind == "dw" else 1)
            if op.bn:
                y = (y - mean) * scale + beta
            else:
                y = y + beta
            if op.relu6:
                y = torch.clamp(y, 0.0, 6.0)
            if op.res is not None:
                y = y + T[op.res]
            T[op.dst] = y
            if op.dst.startswith("box_"):
                boxes.append(y.permute(0, 2, 3, 1).reshape(y.shape[0], -1, 4))
            elif op.dst.startswith("cls_"):
                logits.append(y.permute(0, 2, 3, 1).reshape(y.shape[0], -1, NUM_CLASSES_WITH_BG))
        box_enc = torch.cat(boxes, 1).numpy()
        cls = torch.cat(logits, 1).numpy()
        tensors = None
        if keep:
            tensors = {k: v.permute(0, 2, 3, 1).contiguous().numpy() for k, v in T.items()}
        return box_enc, cls, tensors


def forward(W: Dict[str, np.ndarray], x_nhwc: np.ndarray, keep: bool = False):
    return OracleNet(W).forward(x_nhwc, keep)


def fold_bn(W: Dict[str, np.ndarray], op: ConvSpec) -> Tuple[np.ndarray, np.ndarray]:
    """Reference-side helper for tests: (folded float32 weights in TF layout, float32 bias)."""
    if op.kind == "dw":
        w = W[op.name + "/depthwise_weights"].astype(np.float64)
    else:
        w = W[op.name + "/weights"].astype(np.float64)
    if not op.bn:
        return w.astype(np.float32), W[op.name + "/biases"].astype(np.float32)
    g = W[op.name + "/BatchNorm/gamma"].astype(np.float64)
    b = W[op.name + "/BatchNorm/beta"].astype(np.float64)
    m = W[op.name + "/BatchNorm/moving_mean"].astype(np.float64)
    v = W[op.name + "/BatchNorm/moving_variance"].astype(np.float64)
    s = g / np.sqrt(v + BN_EPS)
    if op.kind == "dw":
        w = w * s[None, None, :, None]
    else:
        w = w * s[None, None, None, :]
    return w.astype(np.float32), (b - m * s).astype(np.float32)


def total_macs(size: int = 300) -> int:
    """MACs per frame (SURVEY.md Appendix A quotes 1 905 M)."""
    sizes = {"input": size}
    macs = 0
    for op in graph_spec():
        hin = sizes[op.src]
        hout = same_pad(hin, op.k, op.stride)[0]
        sizes[op.dst] = hout
        per = op.k * op.k * (1 if op.kind == "dw" else op.cin)
        macs += hout * hout * op.cout * per
    return macs
