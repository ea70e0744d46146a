"""SSD-MobileNet-v2 300x300 as an *op program* for the MI355X engine.

The reference never describes the network: it is whatever model file the user drops into
`model/` (`watsor/detection/tensorflow_cpu.py:14-18,50-53`; README.md:446-451 names
`ssd_mobilenet_v2_coco_2018_03_29`).  The engine builder (`watsor_amd/engine.py`, the
analogue of `watsor/engine.py:17-58`) needs the topology to fold BatchNorm, pad channels and
lay weights out for the HIP kernels, so it is spelled out here.

A program is a list of `Op`s over named activation tensors (NHWC fp16 in HBM).  Variable
names are the TF-slim / TF-OD-API names found in the frozen graph so that a real checkpoint
(dict name -> array) can be packed by the same code path as the seeded synthetic weights.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

INPUT_SIZE = 300
NUM_CLASSES = 91                      # class-head columns per anchor (column 0 = background)
FE = "FeatureExtractor/MobilenetV2/"

# op kinds understood by the runtime (keep in sync with csrc/wz_program.h)
OP_STEM, OP_DW, OP_CONV, OP_MBCONV = 1, 2, 3, 4   # OP_MBCONV: fused inverted-residual block (csrc/k_mbconv.hip)
# output modes of OP_CONV
OUT_ACT, OUT_BOX, OUT_CLS, OUT_HEAD = 0, 1, 2, 3   # OUT_HEAD: box columns then class columns, one launch
ACT_NONE, ACT_RELU6 = 0, 1

_INVERTED_RESIDUAL = [  # (t, c, n, s) rows of the MobileNetV2 paper, depth multiplier 1.0
    (1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1),
]
_EXTRA_DEPTHS = [(256, 512), (128, 256), (128, 256), (64, 128)]
_ANCHORS_PER_LOCATION = [3, 6, 6, 6, 6, 6]


def tf_same(n_in: int, k: int, s: int) -> Tuple[int, int]:
    """(n_out, pad_before) of TensorFlow 'SAME' padding."""
    n_out = (n_in + s - 1) // s
    total = max((n_out - 1) * s + k - n_in, 0)
    return n_out, total // 2


@dataclass
class Tensor:
    name: str
    h: int
    w: int
    c: int
    hp: bool = False           # stored as a pair of fp16 planes per pixel: c "hi" halves then c "lo" halves (x = hi + lo)


@dataclass
class Op:
    kind: int
    scope: str                 # TF variable scope holding weights / BatchNorm / biases
    src: str
    dst: str
    cin: int
    cout: int
    k: int
    stride: int
    act: int
    has_bn: bool
    res: Optional[str] = None
    out_mode: int = OUT_ACT
    head_index: int = -1       # which SSD feature map (for OUT_BOX / OUT_CLS)
    anchors_per_loc: int = 0
    # filled in by build():
    hin: int = 0
    win: int = 0
    hout: int = 0
    wout: int = 0
    pad_t: int = 0
    pad_l: int = 0
    anchor_offset: int = 0     # first anchor index of this head's feature map
    n_box: int = 0             # OUT_HEAD: leading output columns that are box encodings (anchors_per_loc * 4)
    # OP_MBCONV: [1x1 expand ->] depthwise 3x3 -> 1x1 project (+ res) in one launch; cin/cout above describe
    # the PROJECT conv (cin = cmid); `parts` are the unfused ops it replaces (weights are folded per part)
    cmid: int = 0              # depthwise (= expanded) channels
    cin0: int = 0              # block input channels feeding the expand conv; 0 = no expand stage
    parts: Optional[list] = None
    stem: bool = False         # OP_MBCONV whose expand stage is the stem conv (3x3 s2 on the network input, K 27 -> 32)
    stem_pad: Tuple[int, int] = (0, 0)
    block: int = -1            # OP_MBCONV: index of the inverted-residual block (expanded_conv_<block>)
    hp: bool = False           # OP_MBCONV on the split-operand kernel (csrc/k_mbconv_hp.hip): both matrix operands as
                               # hi + lo fp16 pairs, input (and residual) tensor stored as such a pair
    dup_out: bool = False      # OP_MBCONV (split-operand kernel) whose output tensor holds the fp16 output TWICE per pixel (2 * cout plain
                               # channels, [x | x]): the consumer below splits its WEIGHTS instead (the robust program's block 16 -> Conv_1)
    split_w: bool = False      # OP_CONV reading such a tensor: cin = 2 * (the variable's input channels); K = [hi halves of the folded
                               # weights over the first copy | lo halves over the second] -- W.x with ~22 significant bits of W on the plain kernels
    dst2: Optional[str] = None  # OP_MBCONV that also WRITES its expanded tensor (hin x win x cmid, plain fp16): block 13, whose expand
                                # output is the first SSD feature map -- the block stores what it computes anyway, no launch of its own


@dataclass
class Program:
    size: int
    tensors: Dict[str, Tensor] = field(default_factory=dict)
    ops: List[Op] = field(default_factory=list)
    feature_maps: List[Tuple[str, int, int]] = field(default_factory=list)   # (tensor, grid, anchors/loc)
    num_anchors: int = 0

    def variable_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """TF variable name -> shape, for every variable the program consumes."""
        out: Dict[str, Tuple[int, ...]] = {}
        flat: List[Op] = []
        for op in self.ops:
            flat.extend(op.parts if op.kind == OP_MBCONV else [op])
        for op in flat:
            if op.out_mode == OUT_HEAD:
                for sub, cols in (("BoxEncodingPredictor", op.n_box), ("ClassPredictor", op.cout - op.n_box)):
                    out["%s/%s/weights" % (op.scope, sub)] = (op.k, op.k, op.cin, cols)
                    out["%s/%s/biases" % (op.scope, sub)] = (cols,)
                continue
            if op.kind == OP_DW:
                out[op.scope + "/depthwise_weights"] = (op.k, op.k, op.cin, 1)
            else:
                out[op.scope + "/weights"] = (op.k, op.k, op.cin // 2 if op.split_w else op.cin, op.cout)
            if op.has_bn:
                for v in ("gamma", "beta", "moving_mean", "moving_variance"):
                    out[op.scope + "/BatchNorm/" + v] = (op.cout,)
            else:
                out[op.scope + "/biases"] = (op.cout,)
        return out


def _blk(i: int) -> str:
    return "expanded_conv" if i == 0 else "expanded_conv_%d" % i


HP_LAST_BLOCK = 12   # the blocks in front of the first SSD feature map (the 150x150 ... 19x19 maps): the `-p 16` program
HP_ALL_BLOCKS = 16   # every inverted-residual block: the robust program (engine.py --robust)


def build(size: int = INPUT_SIZE, fuse: bool = True, fuse_stem: bool = True, hp_upto: int = -1, input_pair: bool = False,
          tap_in_block: bool = True, conv1_split: bool = False) -> Program:
    """input_pair: the network input is stored as a hi + lo pair of halves even without split-operand blocks (the `-p 32` program:
    one fp16 rounding of the resized image is otherwise the largest error of an engine that computes in fp32).
    hp_upto >= 0 (needs fuse and fuse_stem): blocks 0 .. hp_upto run on the split-operand kernel and the tensors
    between them (the network input included) are hi + lo fp16 pairs; block hp_upto's output is plain fp16 again.
    With hp_upto >= 13 block 13 reads block 12's pair output (and stores the feature map as plain fp16, like the other programs).
    fuse=True: every inverted-residual block is ONE op (OP_MBCONV).  Block 13's expand output is the first SSD feature map:
    in the stem-folded programs the block writes it itself, next to its own output (`dst2`; tap_in_block=False and the
    fuse_stem=False program keep the expand conv as a separate op, which block 13 then reads).  fuse=False: one op per layer (the
    program the per-layer parity tests walk); both programs compute bit-identical tensors.
    conv1_split (needs hp_upto = 16, the robust program): Conv_1's folded weights travel as hi + lo halves too -- block 16 stores its
    fp16 output twice per pixel (`dup_out`) and Conv_1 is a plain 1x1 conv with K = 640 over [hi weights | lo weights] (`split_w`): at two
    decades of channel spread the single fp16 rounding of those 410 k weights is the largest error left (tools/err_budget.py).
    fuse_stem (with fuse): the stem conv becomes the expand stage of the first block -- input image to block
    output in one launch; the stem then runs on the matrix cores with fp16 weights, so this program matches
    the others to fp16 rounding, not bit for bit."""
    p = Program(size=size)
    ops: List[Op] = []
    ops.append(Op(OP_STEM, FE + "Conv", "input", "Conv", 3, 32, 3, 2, ACT_RELU6, True))
    cur, cin, idx = "Conv", 32, 0
    tap0 = None
    for t, c, n, s in _INVERTED_RESIDUAL:
        for j in range(n):
            stride = s if j == 0 else 1
            name = _blk(idx)
            x_in, mid = cur, cin * t
            block: List[Op] = []
            if t != 1:
                block.append(Op(OP_CONV, FE + name + "/expand", cur, name + "/expand", cin, mid, 1, 1, ACT_RELU6, True))
                cur = name + "/expand"
                if idx == 13:
                    tap0 = cur
            block.append(Op(OP_DW, FE + name + "/depthwise", cur, name + "/depthwise", mid, mid, 3, stride,
                            ACT_RELU6, True))
            res = x_in if (stride == 1 and cin == c) else None
            block.append(Op(OP_CONV, FE + name + "/project", name + "/depthwise", name + "/output", mid, c, 1, 1,
                            ACT_NONE, True, res=res))
            if fuse:
                keep_expand = t != 1 and idx == 13          # its output is an SSD feature map
                dst2 = None
                if keep_expand and fuse_stem and (tap_in_block or hp_upto >= idx):   # the block expands for itself and stores the feature map as well
                    dst2 = block[0].dst
                    keep_expand = False
                elif keep_expand:
                    ops.append(block.pop(0))
                has_expand = t != 1 and not keep_expand
                ops.append(Op(OP_MBCONV, FE + name, block[0].src, name + "/output", mid, c, 3, stride, ACT_NONE, True,
                              res=res, cmid=mid, cin0=cin if has_expand else 0, parts=block, block=idx, dst2=dst2))
            else:
                ops.extend(block)
            cur, cin = name + "/output", c
            idx += 1
    if fuse and fuse_stem:
        stem_op, blk0 = ops[0], ops[1]
        assert stem_op.kind == OP_STEM and blk0.kind == OP_MBCONV and blk0.cin0 == 0 and blk0.src == stem_op.dst
        blk0.src, blk0.cin0, blk0.stem = "input", 32, True
        blk0.parts = [stem_op] + blk0.parts
        ops.pop(0)
    if conv1_split:
        if hp_upto != HP_ALL_BLOCKS or not (fuse and fuse_stem):
            raise ValueError("conv1_split needs every block on the split-operand kernel (hp_upto = %d)" % HP_ALL_BLOCKS)
        ops[-1].dup_out = True
    ops.append(Op(OP_CONV, FE + "Conv_1", cur, "Conv_1", 2 * cin if conv1_split else cin, 1280, 1, 1, ACT_RELU6, True, split_w=conv1_split))
    cur, cin = "Conv_1", 1280
    taps = [tap0, "Conv_1"]
    for i, (d1, d2) in enumerate(_EXTRA_DEPTHS):
        n1 = "layer_19_1_Conv2d_%d_1x1_%d" % (i + 2, d1)
        n2 = "layer_19_2_Conv2d_%d_3x3_s2_%d" % (i + 2, d2)
        ops.append(Op(OP_CONV, FE + n1, cur, n1, cin, d1, 1, 1, ACT_RELU6, True))
        ops.append(Op(OP_CONV, FE + n2, n1, n2, d1, d2, 3, 2, ACT_RELU6, True))
        cur, cin = n2, d2
        taps.append(n2)

    if hp_upto >= 0:
        if not (fuse and fuse_stem) or hp_upto > HP_ALL_BLOCKS:
            raise ValueError("split-operand blocks need the fused program with the stem folded in, and end at block %d"
                             % HP_ALL_BLOCKS)
        for op in ops:
            if op.kind == OP_MBCONV and op.block <= hp_upto:
                op.hp = True

    # shape inference
    p.tensors["input"] = Tensor("input", size, size, 3, hp=hp_upto >= 0 or input_pair)
    for op in ops:
        src = p.tensors[op.src]
        op.hin, op.win = src.h, src.w
        if op.stem:                                        # the block sees the stem's output map
            op.hin, st = tf_same(src.h, 3, 2)
            op.win, sl = tf_same(src.w, 3, 2)
            op.stem_pad = (st, sl)
        op.hout, op.pad_t = tf_same(op.hin, op.k, op.stride)
        op.wout, op.pad_l = tf_same(op.win, op.k, op.stride)
        p.tensors[op.dst] = Tensor(op.dst, op.hout, op.wout, 2 * op.cout if op.dup_out else op.cout, hp=op.hp and op.block < hp_upto)
        if op.dst2:
            p.tensors[op.dst2] = Tensor(op.dst2, op.hin, op.win, op.cmid)

    # heads: BoxEncodingPredictor and ClassPredictor of a feature map read the same input, so they run
    # as ONE 3x3 conv whose output columns are [a*4 box encodings | a*91 class logits] (biases, no activation)
    off = 0
    for i, (tname, a) in enumerate(zip(taps, _ANCHORS_PER_LOCATION)):
        tt = p.tensors[tname]
        op = Op(OP_CONV, "BoxPredictor_%d" % i, tname, "head_%d" % i, tt.c, a * 4 + a * NUM_CLASSES,
                3, 1, ACT_NONE, False, out_mode=OUT_HEAD, head_index=i, anchors_per_loc=a)
        op.n_box = a * 4
        op.hin, op.win = tt.h, tt.w
        op.hout, op.pad_t = tf_same(tt.h, 3, 1)
        op.wout, op.pad_l = tf_same(tt.w, 3, 1)
        op.anchor_offset = off
        ops.append(op)
        p.feature_maps.append((tname, tt.h, a))
        off += tt.h * tt.w * a
    p.num_anchors = off
    p.ops = ops
    return p
