"""Error budget of the fp16 engine on the CPU (diagnosis helper, not part of the product path).

Simulates in float64 where the -p 16 engine rounds: per convolution the weights (`w`), the activation
operand of the matrix instruction (`a`) and the stored output (`o`), each one of
    x  exact (fp32-or-better carried through)
    h  one fp16 rounding
    s  hi + lo pair of fp16 halves (what a split operand carries: ~22 significand bits)
and prints max |d sigmoid score| over all 1917 x 90 class scores of a few seeded frames against the
all-exact run.  Used to decide which operands of which layers need the split.

    python tools/err_budget.py baseline          # everything 'h'
    python tools/err_budget.py solo              # one group rounded at a time (contributions add in RSS)
    python tools/err_budget.py plan NAME         # a named mixed-precision plan (PLANS below)
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                    # noqa: E402
import torch.nn.functional as F                 # noqa: E402

from oracle import preprocess as opre           # noqa: E402
from oracle import ssd_mobilenet_v2 as onet     # noqa: E402
from watsor_amd.synth import synthetic_frame, synthetic_weights   # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(os.cpu_count() or 1)


def q16(x):
    return x.to(torch.float16).to(torch.float64)


def qsplit(x, shift=0.0):
    hi = q16(x)
    lo = q16((x - hi) * (2.0 ** shift)) * (2.0 ** -shift)
    return hi + lo


def q32(x):
    return x.to(torch.float32).to(torch.float64)


def quant(x, mode):
    if mode == "x":
        return x
    if mode == "h":
        return q16(x)
    if mode == "s":
        return qsplit(x)
    if mode == "f":
        return q32(x)
    if mode == "u":      # unorm16 of x/6 (relu6 outputs only)
        return torch.round(torch.clamp(x, 0.0, 6.0) * (65535.0 / 6.0)) * (6.0 / 65535.0)
    if mode == "q":      # unorm16 of sqrt(x/6): step 12 sqrt(x/6) / 65535 -- relative precision for small values
        u = torch.round(torch.sqrt(torch.clamp(x, 0.0, 6.0) / 6.0) * 65535.0) / 65535.0
        return u * u * 6.0
    if mode == "e":      # 16-bit float of x + 2^-12: 4 exponent bits (2^-12 .. 2^2), 12 mantissa bits, round to nearest -- relative
        c = 2.0 ** -12  # precision 2^-13 whatever the channel's scale is
        y = (torch.clamp(x, 0.0, 6.0) + c).to(torch.float32)
        b = y.view(torch.int32).to(torch.int64)
        b = ((b + (1 << 10)) >> 11) << 11
        return b.to(torch.int32).view(torch.float32).to(torch.float64) - c
    if mode == "t":      # e3m13 float of t = c + z*K (c = 2^-7, top 2 - 2^-13), TRUNCATED to 13 mantissa bits, mean bias taken out by one factor
        c = 2.0 ** -7
        K = (2.0 - 2.0 ** -13) - c
        t = (c + torch.clamp(x, 0.0, 6.0) / 6.0 * K).to(torch.float32)
        b = t.view(torch.int32).to(torch.int64)
        b = (b >> 10) << 10
        tq = b.to(torch.int32).view(torch.float32).to(torch.float64) * (1.0 + 0.72 * 2.0 ** -14)
        return (tq - c) / K * 6.0
    if mode == "T":      # the same, rounded to nearest
        c = 2.0 ** -7
        K = (2.0 - 2.0 ** -13) - c
        t = (c + torch.clamp(x, 0.0, 6.0) / 6.0 * K).to(torch.float32)
        b = t.view(torch.int32).to(torch.int64)
        b = ((b + 512) >> 10) << 10
        tq = b.to(torch.int32).view(torch.float32).to(torch.float64)
        return (tq - c) / K * 6.0
    if mode == "D":      # round 6's form of the same buffer (csrc/k_hp_ops.h): t = z * T, T = 2^-120 (2 - 2^-13), rounded to 13 mantissa bits; the code is
        T = 2.0 ** -120 * (2.0 - 2.0 ** -13)   # bits 10 .. 25 of t's fp32 pattern (subnormal below 2^-7 of full scale): what `code << 10` decodes to
        t = (torch.clamp(x, 0.0, 6.0) / 6.0 * T).to(torch.float32)
        b = t.view(torch.int32).to(torch.int64)
        b = ((b + 512) >> 10) << 10
        tq = b.to(torch.int32).view(torch.float32).to(torch.float64)
        return tq / T * 6.0
    raise ValueError(mode)


class Net:
    def __init__(self, W):
        self.spec = onet.graph_spec()
        self.w, self.b = [], []
        for op in self.spec:
            w, b = onet.fold_bn(W, op) if False else _fold64(W, op)
            if op.kind == "dw":
                wt = torch.from_numpy(np.ascontiguousarray(w.transpose(2, 3, 0, 1)))
            else:
                wt = torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1)))
            self.w.append(wt)
            self.b.append(torch.from_numpy(b)[None, :, None, None])

    def forward(self, x_nhwc, cfg):
        """cfg: op name -> (w, a, o) modes; missing = ('x','x','x')."""
        T = {"input": torch.from_numpy(np.ascontiguousarray(x_nhwc.transpose(0, 3, 1, 2))).to(torch.float64)}
        logits = []
        boxes = []
        for op, w, b in zip(self.spec, self.w, self.b):
            mw, ma, mo = cfg.get(op.name, ("x", "x", "x"))
            x = quant(T[op.src], ma)
            _, pt, pb = onet.same_pad(x.shape[2], op.k, op.stride)
            _, pl, pr = onet.same_pad(x.shape[3], op.k, op.stride)
            if pt or pb or pl or pr:
                x = F.pad(x, (pl, pr, pt, pb))
            y = F.conv2d(x, quant(w, mw), None, stride=op.stride, groups=op.cin if op.kind == "dw" else 1) + b
            if op.relu6:
                y = torch.clamp(y, 0.0, 6.0)
            if op.res is not None:
                y = y + T[op.res]
            T[op.dst] = quant(y, mo)
            if op.dst.startswith("cls_"):
                logits.append(y.permute(0, 2, 3, 1).reshape(y.shape[0], -1, onet.NUM_CLASSES_WITH_BG))
            elif op.dst.startswith("box_"):
                boxes.append(y.permute(0, 2, 3, 1).reshape(y.shape[0], -1, 4))
        return torch.cat(logits, 1), torch.cat(boxes, 1)


def _fold64(W, op):
    if op.kind == "dw":
        w = W[op.name + "/depthwise_weights"].astype(np.float64)
    else:
        w = W[op.name + "/weights"].astype(np.float64)
    if not op.bn:
        return w, W[op.name + "/biases"].astype(np.float64)
    g = W[op.name + "/BatchNorm/gamma"].astype(np.float64)
    b = W[op.name + "/BatchNorm/beta"].astype(np.float64)
    m = W[op.name + "/BatchNorm/moving_mean"].astype(np.float64)
    v = W[op.name + "/BatchNorm/moving_variance"].astype(np.float64)
    s = g / np.sqrt(v + onet.BN_EPS)
    w = w * (s[None, None, :, None] if op.kind == "dw" else s[None, None, None, :])
    return w, b - m * s


def groups(spec):
    """group label -> list of op names."""
    g = {}
    for op in spec:
        n = op.name
        if n.startswith("BoxPredictor"):
            lab = "heads_cls" if "Class" in n else "heads_box"
        elif "layer_19" in n:
            lab = "extras"
        elif n.endswith("/Conv"):
            lab = "stem"
        elif n.endswith("Conv_1"):
            lab = "Conv_1"
        else:
            blk = n.split("/")[-2]
            idx = 0 if blk == "expanded_conv" else int(blk.split("_")[-1])
            lab = "b%02d_%s" % (idx, n.split("/")[-1][:3])
        g.setdefault(lab, []).append(n)
    return g


def frames_input(n=4, size=(640, 480)):
    xs = [opre.preprocess(synthetic_frame(size[0], size[1], 1234 + i)) for i in range(n)]
    return np.stack(xs)


def score_err(l0, l1):
    s0 = torch.sigmoid(l0[..., 1:])
    s1 = torch.sigmoid(l1[..., 1:])
    d = (s0 - s1).abs()
    # the detections that reach the rows are the high scores: report the error over the top scores too
    top = s0.flatten(1).topk(300, dim=1)
    dtop = d.flatten(1).gather(1, top.indices)
    return float(d.max()), float(dtop.max()), float((l0 - l1).abs().max())


def all_cfg(spec, w, a, o):
    return {op.name: (w, a, o) for op in spec}


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "baseline"
    nfr = int(os.environ.get("NFRAMES", "4"))
    W = synthetic_weights(1234)
    if os.environ.get("SPREAD"):    # per-channel scales spread over that many decades (watsor_amd/synth.py: spread_channel_scales)
        from watsor_amd.synth import spread_channel_scales
        W = spread_channel_scales(W, float(os.environ["SPREAD"]), int(os.environ.get("SPREAD_SEED", "77")))
    net = Net(W)
    x = frames_input(nfr)
    ref, _ = net.forward(x, {})
    spec = net.spec

    def report(label, cfg):
        l, _ = net.forward(x, cfg)
        e = score_err(ref, l)
        print("%-46s max|ds| %.2e  top300 %.2e  max|dlogit| %.2e" % (label, *e), flush=True)
        return e

    if mode == "baseline":
        report("all h/h/h", all_cfg(spec, "h", "h", "h"))
        report("weights only (h/x/x)", all_cfg(spec, "h", "x", "x"))
        report("MFMA inputs only (x/h/x)", all_cfg(spec, "x", "h", "x"))
        report("stored outputs only (x/x/h)", all_cfg(spec, "x", "x", "h"))
        report("all f32 (f/f/f)", all_cfg(spec, "f", "f", "f"))
        report("split everywhere (s/s/s)", all_cfg(spec, "s", "s", "s"))
        report("split w, h acts (s/h/h)", all_cfg(spec, "s", "h", "h"))
    elif mode == "solo":
        g = groups(spec)
        for lab, names in g.items():
            for what, trip in (("w", ("h", "x", "x")), ("a", ("x", "h", "x"))):
                cfg = {n: trip for n in names}
                report("%s %s" % (lab, what), cfg)
    elif mode == "plan":
        import importlib
        plans = importlib.import_module("err_plans") if os.path.exists(os.path.join(os.path.dirname(__file__), "err_plans.py")) else None
        name = sys.argv[2]
        maker = plans.PLANS[name] if name in plans.PLANS else plans.ranges(name)
        cfg = maker(spec, groups(spec))
        report("plan " + sys.argv[2], cfg)


if __name__ == "__main__":
    main()
