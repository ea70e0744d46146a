import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"          # present in the build container only, never on the GPU box
HAVE_REFERENCE = os.path.isdir(os.path.join(REFERENCE, "watsor"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    skip_ref = pytest.mark.skip(reason="/root/reference not present")
    for item in items:
        if "reference" in item.keywords and not HAVE_REFERENCE:
            item.add_marker(skip_ref)


@pytest.fixture(scope="session")
def reference_on_path():
    if not HAVE_REFERENCE:
        pytest.skip("/root/reference not present")
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    return REFERENCE


SEED = 1234


@pytest.fixture(scope="session")
def synth_weights():
    from watsor_amd.synth import synthetic_weights
    return synthetic_weights(SEED)


class ModelDir(str):
    """A model directory path that remembers which `-p 16` program its mi355x.bin holds (`.program`: "default" | "robust")."""
    program = "default"


def _packed(tmp_path_factory, name, image, program="default"):
    from watsor_amd import engine
    d = tmp_path_factory.mktemp(name)
    engine.save_engine(image, str(d / "mi355x.bin"))
    out = ModelDir(str(d))
    out.program = program
    return out


@pytest.fixture(scope="session")
def model_dir_default(synth_weights, tmp_path_factory):
    """A model directory holding mi355x.bin built from the seeded synthetic weights: the default `-p 16` program
    (fused blocks, stem folded in, blocks 0 .. 12 with split matrix operands)."""
    from watsor_amd import engine
    return _packed(tmp_path_factory, "model", engine.build_engine(synth_weights))


@pytest.fixture(scope="session")
def model_dir_robust(synth_weights, tmp_path_factory):
    """The ROBUST `-p 16` program on the same weights (`build_engine(robust=True)`: all 17 blocks on the split-operand kernel, the
    16-bit float-form chunk buffer on blocks 0 .. 9, Conv_1 with split weights) -- what `--robust auto` packs for a trained checkpoint
    and what bench.py's headline is timed on."""
    from watsor_amd import engine
    return _packed(tmp_path_factory, "model_robust", engine.build_engine(synth_weights, robust=True), "robust")


@pytest.fixture(scope="session", params=["default", "robust"])
def model_dir(request):
    """BOTH `-p 16` programs: every test that asks for `model_dir` -- directly or through an engine fixture built
    on it -- runs once on the default program and once on the robust one, the program the benchmark's headline and a trained
    checkpoint get.  Tests that are about one program's own launch shapes ask for `model_dir_default` / `model_dir_robust`."""
    return request.getfixturevalue("model_dir_" + request.param)


@pytest.fixture(scope="session")
def model_dir_plain(synth_weights, tmp_path_factory):
    """`--plain-fp16`: the same fused program with one fp16 rounding per operand everywhere (no split-operand blocks)."""
    from watsor_amd import engine
    d = tmp_path_factory.mktemp("model_plain")
    engine.save_engine(engine.build_engine(synth_weights, hp_upto=-1), str(d / "mi355x.bin"))
    return str(d)


@pytest.fixture(scope="session")
def model_dir_unfused(synth_weights, tmp_path_factory):
    """The same model with one op per layer (no fused inverted-residual blocks): every activation tensor exists."""
    from watsor_amd import engine
    d = tmp_path_factory.mktemp("model_unfused")
    engine.save_engine(engine.build_engine(synth_weights, fuse=False), str(d / "mi355x.bin"))
    return str(d)


@pytest.fixture(scope="session")
def model_dir_stem_separate(synth_weights, tmp_path_factory):
    """Fused blocks with the stem as its own kernel: the program that is bit-identical to the per-layer one."""
    from watsor_amd import engine
    d = tmp_path_factory.mktemp("model_stem_separate")
    engine.save_engine(engine.build_engine(synth_weights, fuse_stem=False), str(d / "mi355x.bin"))
    return str(d)


@pytest.fixture(scope="session")
def model_dir_fp32(synth_weights, tmp_path_factory):
    """The `-p 32` engine: fp32 storage, exact-fp32 matrix cores."""
    from watsor_amd import engine
    d = tmp_path_factory.mktemp("model_fp32")
    engine.save_engine(engine.build_engine(synth_weights, precision=32), str(d / "mi355x.bin"))
    return str(d)


@pytest.fixture(scope="session")
def oracle_net(synth_weights):
    from oracle.ssd_mobilenet_v2 import OracleNet
    return OracleNet(synth_weights)


@pytest.fixture(scope="session")
def frames_640():
    from watsor_amd.synth import synthetic_frame
    return [synthetic_frame(640, 480, SEED + i) for i in range(4)]


def make_engine(model_dir, max_batch=8, max_width=1920, max_height=1080, device=0, dev=False):
    """dev=True: on libwatsor_hip_dev.so (stage-level entry points, WZ_* knobs) -- the stage-by-stage parity tests; the product
    path's tests (plugin, worker, filters, bound frames) run on libwatsor_hip.so."""
    from watsor_amd.runtime import HipEngine
    return HipEngine(os.path.join(model_dir, "mi355x.bin"), device, max_batch, max_width, max_height, dev=dev)
