"""Independent writer of TensorFlow GraphDef files for the tests of watsor_amd/frozen_graph.py: message types declared
here with the field numbers of TensorFlow's public .proto files (graph.proto, node_def.proto, attr_value.proto,
tensor.proto, tensor_shape.proto), serialised by the protobuf library."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

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


def write_frozen_graph(path, variables):
    """A `frozen_inference_graph.pb` holding every variable as a Const node + its `/read` Identity, like a real one."""
    M = _messages()
    g = M["GraphDef"]()
    for name, arr in variables.items():
        _const(M, g, name, arr)
        rd = g.node.add(name=name + "/read", op="Identity")
        rd.input.append(name)
    with open(path, "wb") as f:
        f.write(g.SerializeToString())
