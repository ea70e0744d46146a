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
    msg("AttrValue", [("s", 2, T.TYPE_BYTES, O, None), ("i", 3, T.TYPE_INT64, O, None), ("f", 4, T.TYPE_FLOAT, O, None),
                      ("b", 5, T.TYPE_BOOL, O, None), ("type", 6, T.TYPE_INT32, O, None),
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


def _scalar(M, g, name, value, dtype):
    n = g.node.add(name=name, op="Const")
    n.attr.add(key="dtype").value.type = dtype
    t = n.attr.add(key="value").value.tensor
    t.dtype = dtype
    if dtype == 1:
        t.float_val.append(float(value))
    else:
        t.int_val.append(int(value))
    return name


def _int_vector(M, g, name, values):
    n = g.node.add(name=name, op="Const")
    n.attr.add(key="dtype").value.type = 3
    t = n.attr.add(key="value").value.tensor
    t.dtype = 3
    t.tensor_shape.dim.add(size=len(values))
    import numpy as np
    t.tensor_content = np.asarray(values, "<i4").tobytes()
    return name


def write_detection_graph(path, variables, input_size=(300, 300), align_corners=False, half_pixel_centers=None, iou=0.6,
                          score=1e-8, max_per_class=100, max_total=100, box_scales=(10.0, 10.0, 5.0, 5.0), anchor_scales=None,
                          nms_op="NonMaxSuppressionV3", classes=3, faithful=True):
    """A frozen detection graph in miniature: the variables, and around them the nodes that carry the graph's own settings the way an
    Object Detection API export does -- Preprocessor ResizeBilinear (attributes + size input), per-class NonMaxSuppression nodes
    with constant inputs behind Identity nodes, the FilterGreaterThan comparisons, the box coder's divisions, the anchor
    generator's scale / aspect-ratio constants, the top-k behind the NMS.  `faithful` (default) adds what a real export also has
    in those scopes and what an extractor must NOT mistake for settings: the `tf.minimum(max_size_per_class, num_boxes)` node in front of
    every NMS node, the `tf.minimum(max_total_size, num_boxes)` behind the sort, the decoder's `h / 2.` / `w / 2.` divisions
    (Decode/truediv_4 .. _7) and those of get_center_coordinates_and_sizes."""
    import numpy as np
    M = _messages()
    g = M["GraphDef"]()
    for name, arr in variables.items():
        _const(M, g, name, arr)
        g.node.add(name=name + "/read", op="Identity").input.append(name)
    g.node.add(name="image_tensor", op="Placeholder")
    rs = g.node.add(name="Preprocessor/map/while/ResizeImage/resize/ResizeBilinear", op="ResizeBilinear")
    rs.input.extend(["Preprocessor/map/while/ResizeImage/resize/ExpandDims",
                     _int_vector(M, g, "Preprocessor/map/while/ResizeImage/resize/size", list(input_size))])
    rs.attr.add(key="align_corners").value.b = bool(align_corners)
    if half_pixel_centers is not None:                       # (graphs older than TF 1.14 have no such attribute)
        rs.attr.add(key="half_pixel_centers").value.b = bool(half_pixel_centers)
    scope = "Postprocessor/BatchMultiClassNonMaxSuppression/map/while/MultiClassNonMaxSuppression/"
    for c in range(classes):
        sfx = "" if c == 0 else "_%d" % c
        gt = g.node.add(name=scope + "FilterGreaterThan%s/Greater" % sfx, op="Greater")
        gt.input.extend([scope + "Reshape%s" % sfx, _scalar(M, g, scope + "FilterGreaterThan%s/Greater/y" % sfx, score, 1)])
        nms = g.node.add(name=scope + "non_max_suppression%s/%s" % (sfx, nms_op), op=nms_op)
        size_c = _scalar(M, g, scope + "Minimum%s/x" % sfx, max_per_class, 3)
        ident = g.node.add(name=scope + "non_max_suppression%s/max_output_size" % sfx, op="Identity")
        if faithful:
            mn = g.node.add(name=scope + "Minimum%s" % sfx, op="Minimum")
            mn.input.extend([size_c, scope + "strided_slice%s" % sfx])
            ident.input.append(mn.name)
        else:
            ident.input.append(size_c)
        ins = [scope + "boxes%s" % sfx, scope + "scores%s" % sfx, ident.name,
               _scalar(M, g, scope + "non_max_suppression%s/iou_threshold" % sfx, iou, 1)]
        if nms_op != "NonMaxSuppressionV2":
            ins.append(_scalar(M, g, scope + "non_max_suppression%s/score_threshold" % sfx, float("-inf"), 1))
        nms.input.extend(ins)
    tk = g.node.add(name=scope + "SortByField/TopKV2", op="TopKV2")
    if faithful:                                               # sort_by_field sorts ALL boxes (k = their number), the clip follows
        tk.input.extend([scope + "concat", scope + "SortByField/Size"])
        mt = g.node.add(name=scope + "Minimum_%d" % classes, op="Minimum")
        mt.input.extend([_scalar(M, g, scope + "Minimum_%d/x" % classes, max_total, 3), scope + "SortByField/strided_slice"])
    else:
        tk.input.extend([scope + "concat", _scalar(M, g, scope + "SortByField/k", max_total, 3)])
    g.node.add(name="Postprocessor/Decode/transpose", op="Transpose").input.extend(["Postprocessor/Reshape_1", "Postprocessor/Decode/transpose/perm"])
    g.node.add(name="Postprocessor/Decode/unstack", op="Unpack").input.append("Postprocessor/Decode/transpose")
    if faithful:
        for i, src in ((4, "mul_1"), (5, "mul"), (6, "mul_1"), (7, "mul")):      # ymin = ycenter - h / 2. ...
            d = g.node.add(name="Postprocessor/Decode/truediv_%d" % i, op="RealDiv")
            d.input.extend(["Postprocessor/Decode/" + src, _scalar(M, g, "Postprocessor/Decode/truediv_%d/y" % i, 2.0, 1)])
        for i, src in ((0, "sub_1"), (1, "sub")):                                # ycenter = ymin + height / 2. ...
            nm = "Postprocessor/Decode/get_center_coordinates_and_sizes/truediv" + ("" if i == 0 else "_%d" % i)
            d = g.node.add(name=nm, op="RealDiv")
            d.input.extend(["Postprocessor/Decode/get_center_coordinates_and_sizes/" + src, _scalar(M, g, nm + "/y", 2.0, 1)])
    for i, v in enumerate(box_scales):
        d = g.node.add(name="Postprocessor/Decode/truediv" + ("" if i == 0 else "_%d" % i), op="RealDiv")
        d.input.extend(["Postprocessor/Decode/unstack:%d" % i, _scalar(M, g, "Postprocessor/Decode/truediv%s/y" % ("" if i == 0 else "_%d" % i), v, 1)])
    half = g.node.add(name="Postprocessor/Decode/mul_2", op="Mul")          # (not a scale factor: half of the box extent)
    half.input.extend(["Postprocessor/Decode/Exp", _scalar(M, g, "Postprocessor/Decode/mul_2/y", 0.5, 1)])
    from watsor_amd.anchors import ssd_box_specs
    specs = ssd_box_specs()
    for k, layer in enumerate(specs):
        sc = np.array([s for s, _ in layer], np.float32)
        if anchor_scales is not None and k in anchor_scales:
            sc = np.asarray(anchor_scales[k], np.float32)
        _const(M, g, "MultipleGridAnchorGenerator/scales_%d" % k, sc)
        _const(M, g, "MultipleGridAnchorGenerator/aspect_ratios_%d" % k, np.array([r for _, r in layer], np.float32))
    _const(M, g, "MultipleGridAnchorGenerator/base_anchor_size", np.array([1.0, 1.0], np.float32))
    with open(path, "wb") as f:
        f.write(g.SerializeToString())
