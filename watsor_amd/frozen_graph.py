"""Read the variables of a TensorFlow frozen graph (`frozen_inference_graph.pb`) without TensorFlow.

The reference loads the detector from such a file (`watsor/detection/tensorflow_cpu.py:14-18,50-62`;
README.md:446-451 names `ssd_mobilenet_v2_coco_2018_03_29`).  In a frozen graph every variable is a `Const`
node whose name is the variable's name (`FeatureExtractor/MobilenetV2/Conv/weights`,
`.../BatchNorm/gamma`, `BoxPredictor_0/ClassPredictor/biases`, ...) and whose `value` attribute carries the
tensor.  The engine builder only needs those arrays, so this module walks the protobuf wire format directly
(GraphDef.node = 1; NodeDef.name = 1, op = 2, attr = 5; AttrValue.tensor = 8; TensorProto.dtype = 1,
tensor_shape = 2, tensor_content = 4, float_val = 5; TensorShapeProto.dim = 2, Dim.size = 1) -- neither
TensorFlow nor its .proto files are required.

No model file exists in the reference tree, so this reader is tested against graphs encoded
by an independent writer in tests/test_frozen_graph.py, not against the real checkpoint.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, Tuple

import numpy as np

DT_FLOAT, DT_HALF = 1, 19


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
