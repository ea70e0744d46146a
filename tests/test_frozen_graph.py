"""watsor_amd/frozen_graph.py (TensorFlow-free GraphDef reader) against graphs serialised by the protobuf
library from message types declared here with the field numbers of TensorFlow's public .proto files
(graph.proto, node_def.proto, attr_value.proto, tensor.proto, tensor_shape.proto)."""
import numpy as np
import pytest

pb = pytest.importorskip("google.protobuf")

from watsor_amd import arch, engine                                           # noqa: E402
from watsor_amd.frozen_graph import read_frozen_graph_variables                # noqa: E402

from pb_writer import _const, _messages, write_frozen_graph                 # noqa: E402


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
    path = tmp_path / "frozen_inference_graph.pb"
    write_frozen_graph(str(path), synth_weights)
    W = engine.load_weights(str(path))
    assert set(arch.build(fuse=False).variable_shapes()) <= set(W)
    assert engine.build_engine(W) == engine.build_engine(synth_weights)


# ---- the graph's own settings: read by dataflow, adopted by the engine builder, refused when not reproducible
def _header_fields(blob):
    import struct
    v = struct.unpack_from("<10I6f6Q12I", blob, 0)
    return dict(max_total=v[8], max_per_class=v[9], score=v[10], iou=v[11], scales=v[12:16], resize_mode=v[24], post_flags=v[25])


def test_settings_are_read_by_dataflow_and_reach_the_engine_file(tmp_path, synth_weights):
    from pb_writer import write_detection_graph
    from watsor_amd.frozen_graph import read_frozen_graph_model
    path = str(tmp_path / "frozen_inference_graph.pb")
    write_detection_graph(path, synth_weights, iou=0.5, score=0.3, max_per_class=50, max_total=80, box_scales=(10.0, 10.0, 4.0, 4.0),
                          half_pixel_centers=True)
    W, settings = read_frozen_graph_model(path)
    assert set(arch.build(fuse=False).variable_shapes()) <= set(W)
    assert settings["input_size"] == (300, 300) and settings["resize_half_pixel_centers"] and not settings["resize_align_corners"]
    assert abs(settings["iou_threshold"] - 0.5) < 1e-7 and abs(settings["score_threshold"] - 0.3) < 1e-7
    assert settings["max_per_class"] == 50 and settings["max_total"] == 80 and settings["box_scales"] == (10.0, 10.0, 4.0, 4.0)
    assert len(settings["anchor_vectors"]) == 12 + 1
    post, options = engine.apply_graph_settings(settings, 300, 300, None, None)
    h = _header_fields(engine.build_engine(W, post=post, options=options))
    assert (h["max_total"], h["max_per_class"], h["resize_mode"], h["post_flags"]) == (80, 50, 1, 0)
    assert abs(h["iou"] - 0.5) < 1e-7 and abs(h["score"] - 0.3) < 1e-7 and h["scales"] == (10.0, 10.0, 4.0, 4.0)
    # the defaults of the 2018 graph, and an old graph without the half_pixel_centers attribute: legacy resize
    write_detection_graph(path, synth_weights, nms_op="NonMaxSuppressionV2")
    W2, s2 = read_frozen_graph_model(path)
    post, options = engine.apply_graph_settings(s2, 300, 300, None, None)
    assert options == {"resize": "legacy"} and post["iou_threshold"] == float(np.float32(0.6)) and post["max_total"] == 100
    assert engine.build_engine(W2, post=post, options=options) == engine.build_engine(synth_weights)


def test_a_graph_the_engine_cannot_reproduce_is_refused(tmp_path, synth_weights):
    from pb_writer import write_detection_graph
    from watsor_amd.frozen_graph import read_frozen_graph_model
    path = str(tmp_path / "g.pb")
    small = {k: v for k, v in list(synth_weights.items())[:3]}          # (the settings are what is under test)

    def settings_of(**kw):
        write_detection_graph(path, small, **kw)
        return read_frozen_graph_model(path)[1]

    with pytest.raises(ValueError, match="align_corners"):
        engine.apply_graph_settings(settings_of(align_corners=True), 300, 300, None, None)
    with pytest.raises(ValueError, match="320x320"):
        engine.apply_graph_settings(settings_of(input_size=(320, 320)), 300, 300, None, None)
    with pytest.raises(ValueError, match="anchor"):
        engine.apply_graph_settings(settings_of(anchor_scales={2: [0.3] * 5 + [0.4]}), 300, 300, None, None)
    with pytest.raises(ValueError, match="command line"):
        engine.apply_graph_settings(settings_of(half_pixel_centers=True), 300, 300, None, {"resize": "legacy"})
    with pytest.raises(ValueError, match="100 rows"):
        engine.apply_graph_settings(settings_of(max_total=300), 300, 300, None, None)
    with pytest.raises(ValueError, match="differ"):
        write_detection_graph(path, small)
        from watsor_amd import frozen_graph as fg
        g = fg.read_frozen_graph(path)
        node = [n for n in g.nodes.values() if n.name.endswith("non_max_suppression_1/iou_threshold")][0]
        node.value = np.array([0.45], np.float32)
        fg.graph_settings(g)


def test_engine_cli_adopts_the_graphs_settings(tmp_path, synth_weights, capsys):
    from pb_writer import write_detection_graph
    path = str(tmp_path / "frozen_inference_graph.pb")
    write_detection_graph(path, synth_weights, iou=0.45, half_pixel_centers=True)
    out = str(tmp_path / "model" / "mi355x.bin")
    assert engine.main(["-i", path, "-o", out, "--clip-after-nms"]) == 0
    h = _header_fields(open(out, "rb").read())
    assert abs(h["iou"] - 0.45) < 1e-7 and h["resize_mode"] == 1 and h["post_flags"] == 1
    assert "iou_threshold=0.4" in capsys.readouterr().out          # (0.45 as the float32 the graph holds)
    with pytest.raises(ValueError):
        engine.main(["-i", path, "-o", out, "--resize", "legacy"])


def test_decoder_halvings_are_not_scale_factors_and_ambiguity_only_warns(tmp_path, synth_weights):
    """a real Object Detection API export divides by 2. in the Decode scope as well (`h / 2.`, `w / 2.`,
    get_center_coordinates_and_sizes): the scale factors are the four divisions that read the unstacked encodings, by dataflow.
    The simplified graph of earlier rounds (no Unpack node, no halvings) still reads by name; a decoder nobody can sort out
    warns and keeps the defaults instead of refusing the graph."""
    from pb_writer import write_detection_graph
    from watsor_amd import frozen_graph as fg
    small = {k: v for k, v in list(synth_weights.items())[:3]}
    path = str(tmp_path / "g.pb")
    write_detection_graph(path, small, box_scales=(10.0, 10.0, 5.0, 4.0), max_per_class=30, max_total=70)
    g = fg.read_frozen_graph(path)
    halvings = [n for n in g.nodes.values() if n.op == "RealDiv" and g.constant(n.inputs[1]) is not None and float(g.constant(n.inputs[1]).reshape(-1)[0]) == 2.0]
    assert len(halvings) == 6                                   # (they are in the graph ...)
    s = fg.graph_settings(g)
    assert s["box_scales"] == (10.0, 10.0, 5.0, 4.0)            # (... and not among the scale factors)
    assert s["max_per_class"] == 30 and s["max_total"] == 70    # per-class Minimum nodes are not the total
    write_detection_graph(path, small, faithful=False)
    assert fg.read_frozen_graph_model(path)[1]["box_scales"] == (10.0, 10.0, 5.0, 5.0)
    # a fifth division reading the unstack: ambiguous -> a warning, the defaults, no exception
    g = fg.read_frozen_graph(path)
    extra = fg.Node("Postprocessor/Decode/truediv_9", "RealDiv", ["Postprocessor/Decode/unstack:1", "Postprocessor/Decode/truediv_9/y"], {}, None)
    g.nodes[extra.name] = extra
    g.nodes[extra.name + "/y"] = fg.Node(extra.name + "/y", "Const", [], {}, np.array([7.0], np.float32))
    s = fg.graph_settings(g)
    assert "box_scales" not in s and "box_scales_ambiguous" in s
    with pytest.warns(UserWarning, match="scale factors"):
        post, _ = engine.apply_graph_settings(s, 300, 300, None, None)
    assert "scales" not in post
