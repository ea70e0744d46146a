"""Named mixed-precision plans for tools/err_budget.py (which operands of which layers stay plain fp16)."""


def _idx(lab):
    return int(lab[1:3]) if lab[0] == "b" and lab[1:3].isdigit() else None


def cut(n, hi=("x", "x", "x"), lo=("h", "h", "h"), tail=None):
    """blocks 0..n (and the stem) in `hi`, the rest in `lo`; `tail` overrides Conv_1/extras/heads."""
    def plan(spec, groups):
        cfg = {}
        for lab, names in groups.items():
            i = _idx(lab)
            if lab == "stem" or (i is not None and i <= n):
                m = hi
            elif i is None and tail is not None:
                m = tail
            else:
                m = lo
            for nm in names:
                cfg[nm] = m
        return cfg
    return plan


PLANS = {}
for n in range(3, 17):
    PLANS["cut%d" % n] = cut(n)
    PLANS["cuts%d" % n] = cut(n, hi=("s", "s", "s"))
    # split weights + split MFMA inputs but fp16... no: outputs carried as hi+lo pairs


def ranges(spec_str):
    """'0-6:xxx,7-12:xhh,13-16:hhh,tail:hhh,stem:xxx' -> plan.  Triple = (weights, MFMA input, stored output)."""
    table = {}
    for part in spec_str.split(","):
        key, trip = part.split(":")
        trip = tuple(trip)
        if key in ("tail", "stem", "heads", "conv1", "extras"):
            table[key] = trip
        else:
            lo, _, hi = key.partition("-")
            for i in range(int(lo), int(hi or lo) + 1):
                table[i] = trip

    def plan(spec, groups):
        cfg = {}
        for lab, names in groups.items():
            i = _idx(lab)
            if i is not None:
                m = table.get((i, lab[4:]), table.get(i, ("h", "h", "h")))
            elif lab == "stem":
                m = table.get("stem", ("h", "h", "h"))
            elif lab.startswith("heads"):
                m = table.get("heads", table.get("tail", ("h", "h", "h")))
            elif lab == "Conv_1":
                m = table.get("conv1", table.get("tail", ("h", "h", "h")))
            else:
                m = table.get("extras", table.get("tail", ("h", "h", "h")))
            for nm in names:
                cfg[nm] = m
        return cfg
    return plan


def hp(n, dw_in="u", out_last="h"):
    """The mixed-precision program of the -p 16 engine: stem + blocks 0..n with split MFMA operands, unorm16 depthwise
    input, fp32 depthwise weights, hi+lo block outputs (block n's output plain fp16); everything behind in plain fp16."""
    def plan(spec, groups):
        cfg = {}
        for lab, names in groups.items():
            i = _idx(lab)
            for nm in names:
                if lab == "stem":
                    cfg[nm] = ("s", "s", "x")
                elif i is not None and i <= n:
                    kind = lab[4:]
                    if kind == "exp":
                        cfg[nm] = ("s", "s", "x")
                    elif kind == "dep":
                        cfg[nm] = ("f", dw_in, "x")
                    else:
                        cfg[nm] = ("s", "s", out_last if i == n else "s")
                else:
                    cfg[nm] = ("h", "h", "h")
        return cfg
    return plan


for n in range(2, 17):
    PLANS["hp%d" % n] = hp(n)
    PLANS["hp%dh" % n] = hp(n, dw_in="h")
    PLANS["hp%dx" % n] = hp(n, dw_in="x")
    PLANS["hp%dq" % n] = hp(n, dw_in="q")
    PLANS["hp%de" % n] = hp(n, dw_in="e")
    for m in "tT":                                # round 4: the 16-bit float form, truncated / rounded (err_budget.py: quant)
        PLANS["hp%d%s" % (n, m)] = hp(n, dw_in=m)


def hp_tail(n, exp=("h", "h", "h"), dep=("h", "h", "h"), pro=("h", "h", "h"), dw_in="x", rest=("h", "h", "h")):
    """hp(n) in front (with an exact depthwise input by default), blocks n+1 .. 16 with the given (weights, MFMA input, stored output)
    triples per layer kind, `rest` for Conv_1 / extras / heads -- to find which operand of the plain-fp16 tail decides the error."""
    front = hp(n, dw_in=dw_in)

    def plan(spec, groups):
        cfg = front(spec, groups)
        for lab, names in groups.items():
            i = _idx(lab)
            for nm in names:
                if i is not None and i > n:
                    cfg[nm] = {"exp": exp, "dep": dep, "pro": pro}[lab[4:]]
                elif i is None and lab != "stem":
                    cfg[nm] = rest
        return cfg
    return plan


X, H, S = ("x", "x", "x"), ("h", "h", "h"), ("s", "s", "s")
PLANS["t_all_x"] = hp_tail(12, X, X, X)                                  # tail blocks exact: what is left is Conv_1 / extras / heads
PLANS["t_w"] = hp_tail(12, ("h", "x", "x"), ("h", "x", "x"), ("h", "x", "x"))   # only the tail's weights rounded
PLANS["t_a"] = hp_tail(12, ("x", "h", "x"), ("x", "h", "x"), ("x", "h", "x"))   # only its MFMA / depthwise inputs
PLANS["t_o"] = hp_tail(12, ("x", "x", "h"), ("x", "x", "h"), ("x", "x", "h"))   # only its stored outputs
PLANS["t_exp"] = hp_tail(12, H, X, X)
PLANS["t_dep"] = hp_tail(12, X, H, X)
PLANS["t_pro"] = hp_tail(12, X, X, H)
PLANS["t_rest_x"] = hp_tail(12, H, H, H, rest=X)
PLANS["t_dep_in"] = hp_tail(12, X, ("x", "h", "x"), X)                   # the expanded tensor as fp16 in front of the depthwise conv
PLANS["t_pro_in"] = hp_tail(12, X, X, ("x", "h", "x"))                   # the depthwise output as fp16 in front of the project conv


def hp_rest(n, dw_in="q", conv1=H, extras=H, heads=H, e13_out=None):
    """hp(n, dw_in) with the layers behind the blocks chosen separately (which of Conv_1 / extras / heads decides what is left);
    e13_out: storage mode of block 13's expand output (the first SSD feature map), None = as hp() has it."""
    front = hp(n, dw_in=dw_in)

    def plan(spec, groups):
        cfg = front(spec, groups)
        for lab, names in groups.items():
            for nm in names:
                if lab == "Conv_1":
                    cfg[nm] = conv1
                elif lab.startswith("heads"):
                    cfg[nm] = heads
                elif _idx(lab) is None and lab != "stem":
                    cfg[nm] = extras
                elif e13_out is not None and _idx(lab) == 13 and lab[4:] == "exp":
                    cfg[nm] = (cfg[nm][0], cfg[nm][1], e13_out)
        return cfg
    return plan


PLANS["r_rest_x"] = hp_rest(16, conv1=X, extras=X, heads=X)
PLANS["r_conv1_s"] = hp_rest(16, conv1=S)
PLANS["r_conv1_extras_s"] = hp_rest(16, conv1=S, extras=S)
PLANS["r_heads_s"] = hp_rest(16, heads=S)
PLANS["r_e13h"] = hp_rest(16, e13_out="h")


def hp_tail_terms(first=13, exp_in="h", pro_in="h", exp_w="s", pro_w="s"):
    """hp(16, 'q') with the MFMA inputs / weights of blocks first .. 16 chosen separately: which of the three MFMA terms of the robust
    program's late blocks are worth their time."""
    front = hp(16, dw_in="q")

    def plan(spec, groups):
        cfg = front(spec, groups)
        for lab, names in groups.items():
            i = _idx(lab)
            if i is None or i < first:
                continue
            for nm in names:
                kind = lab[4:]
                if kind == "exp":
                    cfg[nm] = (exp_w, exp_in, cfg[nm][2])
                elif kind == "pro":
                    cfg[nm] = (pro_w, pro_in, cfg[nm][2])
        return cfg
    return plan


PLANS["rt_in_h"] = hp_tail_terms()                                   # both GEMMs of blocks 13..16: activations hi only (2 terms)
PLANS["rt_exp_in_h"] = hp_tail_terms(pro_in="s")                     # only the expand GEMM's input hi only
PLANS["rt_pro_in_h"] = hp_tail_terms(exp_in="s")                     # only the project GEMM's input hi only
PLANS["rt_w_h"] = hp_tail_terms(exp_in="s", pro_in="s", exp_w="h", pro_w="h")   # weights hi only
PLANS["rt_in_h_from7"] = hp_tail_terms(first=7)


def hp_mixed_buffer(first_q):
    """hp(16): linear unorm16 chunk buffer in blocks 0 .. first_q - 1, square-root buffer from block first_q on (the square roots cost most
    on the large maps)."""
    lin, sq = hp(16, dw_in="u"), hp(16, dw_in="q")

    def plan(spec, groups):
        a, b = lin(spec, groups), sq(spec, groups)
        cfg = {}
        for lab, names in groups.items():
            i = _idx(lab)
            for nm in names:
                cfg[nm] = a[nm] if (i is not None and i < first_q) else b[nm]
        return cfg
    return plan


for n in (1, 2, 4, 7):
    PLANS["mixq%d" % n] = hp_mixed_buffer(n)


# ---- round 4: which blocks need the float form, and what Conv_1's split weights are worth (profiles/r04_err_budget_float_form.txt)
PLANS["x_rest_x"] = hp_rest(16, dw_in="x", conv1=X, extras=X, heads=X)
PLANS["x_conv1_s"] = hp_rest(16, dw_in="x", conv1=S)
for tag, c1 in (("s", S), ("hss", ("h", "s", "s")), ("shs", ("s", "h", "s"))):
    PLANS["T_conv1_" + tag] = hp_rest(16, dw_in="T", conv1=c1)


def hp_mixed2(spec, conv1=("s", "h", "s")):
    """spec: list of (last_block, mode) in order, e.g. [(12,'T'),(16,'q')]"""
    plans_ = [(last, hp_rest(16, dw_in=m, conv1=conv1)) for last, m in spec]

    def plan(spec_, groups):
        cfgs = [(last, p(spec_, groups)) for last, p in plans_]
        cfg = {}
        for lab, names in groups.items():
            i = _idx(lab)
            for nm in names:
                chosen = cfgs[-1][1]
                if i is not None:
                    for last, c in cfgs:
                        if i <= last:
                            chosen = c
                            break
                cfg[nm] = chosen[nm]
        return cfg
    return plan


PLANS["mx_T12_q16"] = hp_mixed2([(12, "T"), (16, "q")])
PLANS["mx_T12_u16"] = hp_mixed2([(12, "T"), (16, "u")])
PLANS["mx_T6_q16"] = hp_mixed2([(6, "T"), (16, "q")])
PLANS["mx_T3_q16"] = hp_mixed2([(3, "T"), (16, "q")])
PLANS["mx_q3_T16"] = hp_mixed2([(3, "q"), (16, "T")])
PLANS["mx_q6_T16"] = hp_mixed2([(6, "q"), (16, "T")])
PLANS["mx_T12_q16_c1h"] = hp_mixed2([(12, "T"), (16, "q")], conv1=("h", "h", "h"))
PLANS["mx_T16_c1h"] = hp_mixed2([(16, "T")], conv1=("h", "h", "h"))
PLANS["mx_T12_u16_c1h"] = hp_mixed2([(12, "T"), (16, "u")], conv1=("h", "h", "h"))
PLANS["mx_T3_u16"] = hp_mixed2([(3, "T"), (16, "u")])
PLANS["mx_T6_u16"] = hp_mixed2([(6, "T"), (16, "u")])
PLANS["mx_T9_u16"] = hp_mixed2([(9, "T"), (16, "u")])
PLANS["mx_T10_u16"] = hp_mixed2([(10, "T"), (16, "u")])
for k in (8, 9, 10, 11, 12):
    PLANS["mx_T%d_u16_c1h" % k] = hp_mixed2([(k, "T"), (16, "u")], conv1=("h", "h", "h"))
    PLANS["mx_T%d_u16_c1s" % k] = hp_mixed2([(k, "T"), (16, "u")])
# the float form only where it is cheap (the large maps of blocks 0 .. 3 pay most for its encoder): linear in front as well
for k0 in (0, 1, 2, 3):
    PLANS["mx_u%d_T9_u16" % k0] = hp_mixed2([(k0, "u"), (9, "T"), (16, "u")])
    PLANS["mx_u%d_T12_u16" % k0] = hp_mixed2([(k0, "u"), (12, "T"), (16, "u")])
