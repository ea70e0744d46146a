"""watsor_amd/frozen_graph.py (TensorFlow-free GraphDef reader) against graphs serialised by the protobuf
library from message types declared here with the field numbers of TensorFlow's public .proto files
(graph.proto, node_def.proto, attr_value.proto, tensor.proto, tensor_shape.proto)."""
import numpy as np
import pytest

pb = pytest.importorskip("google.protobuf")
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory   # noqa: E402

from watsor_amd import arch, engine                                           # noqa: E402
from watsor_amd.frozen_graph import read_frozen_graph_variables                # noqa: E402

T = descriptor_pb2.FieldDescriptorProto


def _messages():
    fd = descriptor_pb2.FileDescriptorProto(name="tf_min.proto", package="tfmin", syntax="proto3")

    def msg(name, fields, nested=None):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = tname
        return m

    R, O = T.LABEL_REPEATED, T.LABEL_OPTIONAL
    dim = msg("Dim", [("size", 1, T.TYPE_INT64, O, None), ("name", 2, T.TYPE_STRING, O, None)])
    msg("TensorShapeProto", [("dim", 2, T.TYPE_MESSAGE, R, ".tfmin.Dim"), ("unknown_rank", 3, T.TYPE_BOOL, O, None)])
    msg("TensorProto", [("dtype", 1, T.TYPE_INT32, O, None), ("tensor_shape", 2, T.TYPE_MESSAGE, O, ".tfmin.TensorShapeProto"),
                        ("version_number", 3, T.TYPE_INT32, O, None), ("tensor_content", 4, T.TYPE_BYTES, O, None),
                        ("float_val", 5, T.TYPE_FLOAT, R, None), ("int_val", 7, T.TYPE_INT32, R, None)])
    msg("AttrValue", [("s", 2, T.TYPE_BYTES, O, None), ("i", 3, T.TYPE_INT64, O, None), ("type", 6, T.TYPE_INT32, O, None),
                      ("shape", 7, T.TYPE_MESSAGE, O, ".tfmin.TensorShapeProto"), ("tensor", 8, T.TYPE_MESSAGE, O, ".tfmin.TensorProto")])
    msg("AttrEntry", [("key", 1, T.TYPE_STRING, O, None), ("value", 2, T.TYPE_MESSAGE, O, ".tfmin.AttrValue")])
    msg("NodeDef", [("name", 1, T.TYPE_STRING, O, None), ("op", 2, T.TYPE_STRING, O, None), ("input", 3, T.TYPE_STRING, R, None),
                    ("device", 4, T.TYPE_STRING, O, None), ("attr", 5, T.TYPE_MESSAGE, R, ".tfmin.AttrEntry")])
    msg("GraphDef", [("node", 1, T.TYPE_MESSAGE, R, ".tfmin.NodeDef"), ("version", 3, T.TYPE_INT32, O, None)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("tfmin." + n))
            for n in ("GraphDef", "NodeDef", "AttrEntry", "AttrValue", "TensorProto", "TensorShapeProto", "Dim")}


def _const(M, g, name, arr=None, dtype=1, splat=None, shape=None, as_floats=False):
    n = g.node.add(name=name, op="Const")
    a = n.attr.add(key="dtype")
    a.value.type = dtype
    v = n.attr.add(key="value")
    t = v.value.tensor
    t.dtype = dtype
    for d in (shape if shape is not None else arr.shape):
        t.tensor_shape.dim.add(size=int(d))
    if splat is not None:
        t.float_val.append(float(splat))
    elif as_floats:
        t.float_val.extend(arr.reshape(-1).tolist())
    else:
        t.tensor_content = (arr.astype("<f2") if dtype == 19 else arr.astype("<f4")).tobytes()


def test_reads_every_float_const_and_nothing_else(tmp_path):
    M = _messages()
    g = M["GraphDef"](version=27)
    rng = np.random.default_rng(5)
    w = rng.standard_normal((3, 3, 3, 32)).astype(np.float32)
    gamma = rng.standard_normal(32).astype(np.float32)
    half = rng.standard_normal((2, 5)).astype(np.float16)
    ph = g.node.add(name="image_tensor", op="Placeholder")
    ph.attr.add(key="dtype").value.type = 4
    _const(M, g, "FeatureExtractor/MobilenetV2/Conv/weights", w)
    _const(M, g, "FeatureExtractor/MobilenetV2/Conv/BatchNorm/gamma", gamma, as_floats=True)
    _const(M, g, "ones", splat=1.0, shape=(4, 2))
    _const(M, g, "halfs", half, dtype=19)
    ints = g.node.add(name="Postprocessor/shape", op="Const")
    it = ints.attr.add(key="value").value.tensor
    it.dtype = 3
    it.int_val.extend([1, 2, 3])
    it.tensor_shape.dim.add(size=3)
    cv = g.node.add(name="FeatureExtractor/MobilenetV2/Conv/Conv2D", op="Conv2D")
    cv.input.extend(["Preprocessor/sub", "FeatureExtractor/MobilenetV2/Conv/weights/read"])
    path = tmp_path / "frozen_inference_graph.pb"
    path.write_bytes(g.SerializeToString())
    got = read_frozen_graph_variables(str(path))
    assert sorted(got) == ["FeatureExtractor/MobilenetV2/Conv/BatchNorm/gamma", "FeatureExtractor/MobilenetV2/Conv/weights",
                           "halfs", "ones"]
    np.testing.assert_array_equal(got["FeatureExtractor/MobilenetV2/Conv/weights"], w)
    np.testing.assert_array_equal(got["FeatureExtractor/MobilenetV2/Conv/BatchNorm/gamma"], gamma)
    np.testing.assert_array_equal(got["ones"], np.ones((4, 2), np.float32))
    np.testing.assert_array_equal(got["halfs"], half.astype(np.float32))
    with pytest.raises(ValueError):
        (tmp_path / "junk.pb").write_bytes(b"\x00\x01\x02")
        read_frozen_graph_variables(str(tmp_path / "junk.pb"))


def test_engine_builder_accepts_a_frozen_graph(tmp_path, synth_weights):
    """`python -m watsor_amd.engine -i frozen_inference_graph.pb`: the whole variable set through the .pb path gives
    the same engine image as the dict it was made from."""
    M = _messages()
    g = M["GraphDef"]()
    for name, arr in synth_weights.items():
        _const(M, g, name, arr)
        rd = g.node.add(name=name + "/read", op="Identity")
        rd.input.append(name)
    path = tmp_path / "frozen_inference_graph.pb"
    path.write_bytes(g.SerializeToString())
    W = engine.load_weights(str(path))
    assert set(arch.build(fuse=False).variable_shapes()) <= set(W)
    assert engine.build_engine(W) == engine.build_engine(synth_weights)
