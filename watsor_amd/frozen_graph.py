"""Read the variables of a TensorFlow frozen graph (`frozen_inference_graph.pb`) without TensorFlow.

The reference loads the detector from such a file (`watsor/detection/tensorflow_cpu.py:14-18,50-62`;
README.md:446-451 names `ssd_mobilenet_v2_coco_2018_03_29`).  In a frozen graph every variable is a `Const`
node whose name is the variable's name (`FeatureExtractor/MobilenetV2/Conv/weights`,
`.../BatchNorm/gamma`, `BoxPredictor_0/ClassPredictor/biases`, ...) and whose `value` attribute carries the
tensor.  The engine builder only needs those arrays, so this module walks the protobuf wire format directly
(GraphDef.node = 1; NodeDef.name = 1, op = 2, attr = 5; AttrValue.tensor = 8; TensorProto.dtype = 1,
tensor_shape = 2, tensor_content = 4, float_val = 5; TensorShapeProto.dim = 2, Dim.size = 1) -- neither
TensorFlow nor its .proto files are required.

Besides the weights the graph carries its own SETTINGS -- the resize mode (`ResizeBilinear`'s `align_corners` /
`half_pixel_centers` attributes and its size input), the NMS thresholds and output sizes (the inputs of the
`NonMaxSuppressionV2/V3/...` nodes), the score filter (`Greater` against a constant in the `FilterGreaterThan` scopes),
the box coder's scale factors (the divisors in `Postprocessor/Decode`), the anchor generator's scales and aspect ratios
(`MultipleGridAnchorGenerator` constants) -- which the reference gets for free by running the graph
(`tensorflow_cpu.py:94-121`).  `read_frozen_graph` keeps the node list (op, inputs, attributes) and `graph_settings`
finds those values BY DATAFLOW (which constant feeds which input of which op), not by node name, so that the engine
builder can adopt them and refuse what it cannot honour instead of silently assuming its defaults.

No model file exists in the reference tree, so this reader is tested against graphs encoded
by an independent writer in tests/test_frozen_graph.py, not against the real checkpoint.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, Tuple

import numpy as np

DT_FLOAT, DT_HALF, DT_INT32, DT_INT64, DT_BOOL = 1, 19, 3, 9, 10


def _varint(buf: memoryview, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf: memoryview) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) of one message; value = int for varint/fixed, memoryview for bytes."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
        elif wire == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wire == 2:
            n, pos = _varint(buf, pos)
            if pos + n > end:
                raise ValueError("truncated length-delimited field")
            val = buf[pos:pos + n]
            pos += n
        elif wire == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wire)
        yield field, wire, val


def _shape(buf: memoryview) -> Tuple[int, ...]:
    dims = []
    for f, w, v in _fields(buf):
        if f == 2 and w == 2:                                  # TensorShapeProto.Dim
            size = 0
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == 0:
                    size = v2 - (1 << 64) if v2 >= (1 << 63) else v2
            dims.append(int(size))
    return tuple(dims)


def _tensor(buf: memoryview):
    dtype, shape, content, floats = 0, (), None, []
    for f, w, v in _fields(buf):
        if f == 1 and w == 0:
            dtype = v
        elif f == 2 and w == 2:
            shape = _shape(v)
        elif f == 4 and w == 2:
            content = bytes(v)
        elif f == 5:
            if w == 2:                                         # packed repeated float
                floats.extend(np.frombuffer(bytes(v), "<f4").tolist())
            elif w == 5:
                floats.append(struct.unpack("<f", struct.pack("<I", v))[0])
    if dtype not in (DT_FLOAT, DT_HALF):
        return None
    n = int(np.prod(shape)) if shape else 1
    if content is not None:
        arr = np.frombuffer(content, "<f4" if dtype == DT_FLOAT else "<f2").astype(np.float32)
    elif floats:
        arr = np.asarray(floats, np.float32)
        if arr.size == 1 and n > 1:                            # TF stores a splat as one value
            arr = np.full(n, arr[0], np.float32)
    else:
        arr = np.zeros(n, np.float32)
    if arr.size != n:
        raise ValueError("tensor has %d values, shape %s needs %d" % (arr.size, shape, n))
    return arr.reshape(shape)


def _node(buf: memoryview):
    name = op = None
    value = None
    for f, w, v in _fields(buf):
        if f == 1 and w == 2:
            name = bytes(v).decode()
        elif f == 2 and w == 2:
            op = bytes(v).decode()
        elif f == 5 and w == 2:                                # map<string, AttrValue> entry
            key, attr = None, None
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == 2:
                    key = bytes(v2).decode()
                elif f2 == 2 and w2 == 2:
                    attr = v2
            if key == "value" and attr is not None:
                for f3, w3, v3 in _fields(attr):
                    if f3 == 8 and w3 == 2:                    # AttrValue.tensor
                        value = v3
    return name, op, value


def read_frozen_graph_variables(path: str) -> Dict[str, np.ndarray]:
    """name -> float32 array of every floating-point `Const` node of the GraphDef in `path`."""
    with open(path, "rb") as f:
        data = memoryview(f.read())
    out: Dict[str, np.ndarray] = {}
    nodes = 0
    try:
        for f_, w, v in _fields(data):
            if f_ != 1 or w != 2:
                continue
            nodes += 1
            name, op, value = _node(v)
            if op == "Const" and name and value is not None:
                arr = _tensor(value)
                if arr is not None:
                    out[name] = arr
    except (IndexError, struct.error, UnicodeDecodeError) as exc:
        raise ValueError("%s is not a well-formed TensorFlow GraphDef (%s)" % (path, exc)) from None
    if nodes == 0:
        raise ValueError("%s does not look like a TensorFlow GraphDef" % path)
    return out


# ------------------------------------------------------------------------------------------------
# the graph itself: nodes, inputs, attributes -- and the settings the post-processing sub-graph carries
# ------------------------------------------------------------------------------------------------
def _int_tensor(buf: memoryview):
    """int32 / int64 / bool TensorProto -> int64 array, else None (small constants: sizes, counts)."""
    dtype, shape, content, ints = 0, (), None, []
    for f, w, v in _fields(buf):
        if f == 1 and w == 0:
            dtype = v
        elif f == 2 and w == 2:
            shape = _shape(v)
        elif f == 4 and w == 2:
            content = bytes(v)
        elif f in (7, 10, 11):                                 # int_val, int64_val, bool_val
            if w == 2:
                pos, vals = 0, []
                while pos < len(v):
                    x, pos = _varint(v, pos)
                    vals.append(x)
                ints.extend(vals)
            elif w == 0:
                ints.append(v)
    if dtype not in (DT_INT32, DT_INT64, DT_BOOL):
        return None
    n = int(np.prod(shape)) if shape else 1
    if content is not None:
        arr = np.frombuffer(content, {DT_INT32: "<i4", DT_INT64: "<i8", DT_BOOL: "u1"}[dtype]).astype(np.int64)
    else:
        arr = np.asarray([x - (1 << 64) if x >= (1 << 63) else x for x in ints], np.int64)
        if arr.size == 1 and n > 1:
            arr = np.full(n, arr[0], np.int64)
        if arr.size == 0:
            arr = np.zeros(n, np.int64)
    return arr.reshape(shape) if arr.size == n else None


class Node:
    __slots__ = ("name", "op", "inputs", "attrs", "value")

    def __init__(self, name, op, inputs, attrs, value):
        self.name, self.op, self.inputs, self.attrs, self.value = name, op, inputs, attrs, value

    def __repr__(self):
        return "Node(%r, %r, inputs=%r)" % (self.name, self.op, self.inputs)


def _full_node(buf: memoryview) -> Node:
    name = op = ""
    inputs, attrs, value = [], {}, None
    for f, w, v in _fields(buf):
        if f == 1 and w == 2:
            name = bytes(v).decode()
        elif f == 2 and w == 2:
            op = bytes(v).decode()
        elif f == 3 and w == 2:
            inputs.append(bytes(v).decode())
        elif f == 5 and w == 2:                                # map<string, AttrValue> entry
            key, attr = None, None
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == 2:
                    key = bytes(v2).decode()
                elif f2 == 2 and w2 == 2:
                    attr = v2
            if key is None or attr is None:
                continue
            for f3, w3, v3 in _fields(attr):                   # AttrValue: s = 2, i = 3, f = 4, b = 5, type = 6, tensor = 8
                if f3 == 5 and w3 == 0:
                    attrs[key] = bool(v3)
                elif f3 == 3 and w3 == 0:
                    attrs[key] = v3 - (1 << 64) if v3 >= (1 << 63) else v3
                elif f3 == 4 and w3 == 5:
                    attrs[key] = struct.unpack("<f", struct.pack("<I", v3))[0]
                elif f3 == 2 and w3 == 2:
                    attrs[key] = bytes(v3)
                elif f3 == 8 and w3 == 2 and key == "value":
                    t = _tensor(v3)
                    value = t if t is not None else _int_tensor(v3)
    return Node(name, op, inputs, attrs, value)


class Graph:
    """The nodes of a GraphDef by name, in file order."""

    def __init__(self, nodes):
        self.nodes = {n.name: n for n in nodes}

    def resolve(self, ref: str):
        """The node an input string names ("^ctrl" excluded, ":k" output suffix dropped), looking through Identity nodes."""
        seen = 0
        while ref and not ref.startswith("^") and seen < 16:
            node = self.nodes.get(ref.split(":")[0])
            if node is None or node.op not in ("Identity", "StopGradient"):
                return node
            ref = node.inputs[0] if node.inputs else ""
            seen += 1
        return None

    def constant(self, ref: str):
        """The value of the Const node behind an input reference (through Identity / Cast of a constant), else None."""
        node = self.resolve(ref)
        if node is not None and node.op == "Cast" and node.inputs:
            node = self.resolve(node.inputs[0])
        return node.value if node is not None and node.op == "Const" else None

    def variables(self) -> Dict[str, np.ndarray]:
        return {n.name: n.value for n in self.nodes.values()
                if n.op == "Const" and n.value is not None and n.value.dtype == np.float32}


def read_frozen_graph(path: str) -> Graph:
    with open(path, "rb") as f:
        data = memoryview(f.read())
    nodes = []
    try:
        for f_, w, v in _fields(data):
            if f_ == 1 and w == 2:
                nodes.append(_full_node(v))
    except (IndexError, struct.error, UnicodeDecodeError) as exc:
        raise ValueError("%s is not a well-formed TensorFlow GraphDef (%s)" % (path, exc)) from None
    if not nodes:
        raise ValueError("%s does not look like a TensorFlow GraphDef" % path)
    return Graph(nodes)


def _one(values, what):
    """The single distinct value of a list (the 90 per-class NMS nodes of an exported graph all carry the same constants)."""
    uniq = sorted(set(values))
    if len(uniq) > 1:
        raise ValueError("the graph's %s differ from node to node: %s" % (what, uniq[:6]))
    return uniq[0] if uniq else None


def _index_of(name: str) -> int:
    """TensorFlow's uniquifying suffix of the LAST path component: truediv -> 0, truediv_3 -> 3."""
    tail = name.rsplit("/", 1)[-1]
    head, _, num = tail.rpartition("_")
    return int(num) if head and num.isdigit() else 0


def graph_settings(g: Graph) -> dict:
    """What the graph itself says about the stages around the network; keys are present only for what was found:
      input_size (h, w), resize_align_corners, resize_half_pixel_centers     <- the ResizeBilinear node
      iou_threshold, max_per_class, score_threshold                          <- NonMaxSuppressionV2/V3/V4/V5 inputs, FilterGreaterThan
      max_total                                                             <- the TopKV2 / Minimum constant behind the per-class NMS
      box_scales (ty, tx, th, tw divisors)                                  <- RealDiv / Mul by a constant in a .../Decode scope
      anchor_vectors                                                        <- 1-D float constants of the anchor generator's scope"""
    out = {}
    for n in g.nodes.values():
        if n.op == "ResizeBilinear":
            out["resize_align_corners"] = bool(n.attrs.get("align_corners", False))
            out["resize_half_pixel_centers"] = bool(n.attrs.get("half_pixel_centers", False))
            size = g.constant(n.inputs[1]) if len(n.inputs) > 1 else None
            if size is not None and size.size == 2:
                out["input_size"] = (int(size.reshape(-1)[0]), int(size.reshape(-1)[1]))
            break
    iou, per_class, score = [], [], []
    for n in g.nodes.values():
        if n.op.startswith("NonMaxSuppression") and len(n.inputs) >= 4:
            m, t = g.constant(n.inputs[2]), g.constant(n.inputs[3])
            if m is None:                                        # `tf.minimum(max_size_per_class, boxlist.num_boxes())`: the constant side
                mn = g.resolve(n.inputs[2])
                if mn is not None and mn.op == "Minimum":
                    for ref in mn.inputs:
                        m = g.constant(ref)
                        if m is not None:
                            break
            if m is not None and m.size == 1:
                per_class.append(int(m.reshape(-1)[0]))
            if t is not None and t.size == 1:
                iou.append(float(np.float32(t.reshape(-1)[0])))
            if len(n.inputs) >= 5:
                s_ = g.constant(n.inputs[4])
                if s_ is not None and s_.size == 1 and np.isfinite(s_.reshape(-1)[0]):
                    score.append(float(np.float32(s_.reshape(-1)[0])))
        elif n.op == "Greater" and "FilterGreaterThan" in n.name and len(n.inputs) == 2:
            s_ = g.constant(n.inputs[1])
            if s_ is not None and s_.size == 1 and s_.dtype == np.float32:
                score.append(float(s_.reshape(-1)[0]))
    if iou:
        out["iou_threshold"] = _one(iou, "NMS IoU thresholds")
    if per_class:
        out["max_per_class"] = _one(per_class, "NMS output sizes")
    score = [s_ for s_ in score if s_ > -1e30]                  # (NonMaxSuppressionV3's default "-inf" filter is no filter)
    if score:
        out["score_threshold"] = _one(score, "score thresholds")
    # max_total: the constant of the top-k / Minimum that consumes the CONCATENATED boxlist (sort_by_field + pad-or-clip behind the
    # per-class loop).  The per-class `Minimum(max_size_per_class, num_boxes)` nodes sit in the same scope: constants that are an
    # NMS node's max_output_size input are those and are left out, so a graph with max_detections_per_class != max_total_detections
    # yields the total instead of "differ from node to node".
    feeds_nms = set()                                            # nodes between an NMS node and the Const behind its max_output_size
    for n in g.nodes.values():
        if n.op.startswith("NonMaxSuppression") and len(n.inputs) >= 3:
            todo, hops = [n.inputs[2]], 0
            while todo and hops < 64:
                node = g.nodes.get(todo.pop().split(":")[0].lstrip("^"))
                hops += 1
                if node is None or node.name in feeds_nms:
                    continue
                feeds_nms.add(node.name)
                if node.op in ("Identity", "StopGradient", "Minimum", "Cast"):
                    todo.extend(node.inputs)
    totals = []
    for n in g.nodes.values():
        if "MultiClassNonMaxSuppression" in n.name and n.op in ("TopKV2", "Minimum") and len(n.inputs) == 2 and n.name not in feeds_nms:
            for ref in (n.inputs[1], n.inputs[0]):
                k = g.constant(ref)
                if k is not None and k.size == 1 and k.dtype == np.int64:
                    totals.append(int(k.reshape(-1)[0]))
                    break
    if totals:
        out["max_total"] = _one(totals, "maximum total detections")
    # box_scales: FasterRcnnBoxCoder._decode divides the four rows of the transposed, unstacked encodings by its scale factors
    # (`ty /= 10.` ...) -- RealDiv nodes whose FIRST input is an output of that Unpack node; the output index says which of ty, tx,
    # th, tw it is.  The same scope also divides by 2. (`h / 2.`, `w / 2.`: Decode/truediv_4 .. _7, and
    # get_center_coordinates_and_sizes/truediv{,_1}): those read products / differences, not the unstack, and are no scale factors.
    by_unstack = {}
    for n in g.nodes.values():
        if "/Decode/" in "/" + n.name and n.op in ("RealDiv", "Div") and len(n.inputs) == 2:
            c = g.constant(n.inputs[1])
            src = g.resolve(n.inputs[0])
            if c is not None and c.size == 1 and c.dtype == np.float32 and src is not None and src.op == "Unpack" \
                    and "get_center_coordinates_and_sizes" not in n.name:
                ref = n.inputs[0]
                hops = 0
                while hops < 16:                                  # the output index survives Identity nodes in between
                    node = g.nodes.get(ref.split(":")[0])
                    if node is None or node.op not in ("Identity", "StopGradient"):
                        break
                    ref = node.inputs[0]
                    hops += 1
                idx = int(ref.split(":")[1]) if ":" in ref else 0
                by_unstack.setdefault(idx, []).append(float(c.reshape(-1)[0]))
    if by_unstack:
        if sorted(by_unstack) == [0, 1, 2, 3] and all(len(set(v)) == 1 for v in by_unstack.values()):
            out["box_scales"] = tuple(round(by_unstack[i][0], 6) for i in range(4))
        else:
            out["box_scales_ambiguous"] = {k: sorted(set(v)) for k, v in by_unstack.items()}
    else:
        # graphs whose decoder does not read an Unpack node (other exporters): by name, leaving out what cannot be a scale factor
        scales = []
        for n in g.nodes.values():
            if "/Decode/" in "/" + n.name and n.op in ("RealDiv", "Div", "Mul") and len(n.inputs) == 2 \
                    and "get_center_coordinates_and_sizes" not in n.name:
                c = g.constant(n.inputs[1])
                if c is not None and c.size == 1 and c.dtype == np.float32 and g.constant(n.inputs[0]) is None:
                    v = float(c.reshape(-1)[0])
                    if v == 0.0 or v in (0.5, 2.0):                # (half-extent arithmetic of the decoder, not a scale factor)
                        continue
                    if n.op == "Mul":
                        v = 1.0 / v
                    scales.append((n.op != "Mul", _index_of(n.name), n.name, v))
        if scales:
            divs = sorted(x for x in scales if x[0]) or sorted(scales)
            vals = tuple(round(x[3], 6) for x in sorted(divs, key=lambda x: (x[1], x[2])))
            if len(vals) == 4:
                out["box_scales"] = vals
            else:
                out["box_scales_ambiguous"] = {"by_name": list(vals)}
    vecs = [n.value.astype(np.float32) for n in g.nodes.values()
            if n.op == "Const" and "AnchorGenerator" in n.name and n.value is not None and n.value.dtype == np.float32
            and n.value.ndim == 1 and 2 <= n.value.size <= 16]
    if vecs:
        out["anchor_vectors"] = vecs
    return out


def read_frozen_graph_model(path: str):
    """(variables, settings) of a frozen detection graph: what `python -m watsor_amd.engine -i model.pb` builds from."""
    g = read_frozen_graph(path)
    return g.variables(), graph_settings(g)
