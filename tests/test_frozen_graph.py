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
