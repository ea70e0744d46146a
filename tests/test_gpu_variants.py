"""The engine-file options for the two steps that differ between exporter generations,
on the GPU against their oracle twins: `resize` = legacy | half_pixel (ResizeBilinear's coordinate rule) and `clip_after_nms`
(per-class NMS on the boxes as decoded, clipping afterwards).  Both settings of both options, stage by stage and end to end."""
import numpy as np
import pytest

import conftest
import parity_utils as pu
from oracle import detect as odet
from oracle import postprocess as post
from oracle import preprocess as pre
from watsor_amd import engine
from watsor_amd.runtime import ROW_DTYPE
from watsor_amd.synth import synthetic_frame

pytestmark = pytest.mark.gpu


def build(tmp_path_factory, synth_weights, name, **options):
    d = tmp_path_factory.mktemp(name)
    engine.save_engine(engine.build_engine(synth_weights, options=options), str(d / "mi355x.bin"))
    return str(d)


@pytest.fixture(scope="module")
def dir_half_pixel(tmp_path_factory, synth_weights):
    return build(tmp_path_factory, synth_weights, "half_pixel", resize="half_pixel")


@pytest.fixture(scope="module")
def dir_clip_after(tmp_path_factory, synth_weights):
    return build(tmp_path_factory, synth_weights, "clip_after", clip_after_nms=True)


@pytest.mark.parametrize("wh", [(640, 480), (1920, 1080), (300, 300), (301, 299), (150, 100), (64, 48)])
def test_half_pixel_resize_bit_exact(dir_half_pixel, wh):
    e = conftest.make_engine(dir_half_pixel, max_batch=1, dev=True)
    try:
        f = synthetic_frame(wh[0], wh[1], 11 + wh[0])
        got = e.stage_preprocess(f)
        ref32 = pre.preprocess(f, half_pixel_centers=True)
        ref = ref32.astype(np.float16)
        np.testing.assert_array_equal(got[..., :3].view(np.uint16), ref.view(np.uint16))
        lo = (ref32 - ref.astype(np.float32)).astype(np.float16)
        np.testing.assert_array_equal(got[..., 4:7].view(np.uint16), lo.view(np.uint16))
        assert np.abs(ref32 - pre.preprocess(f)).max() > 1e-3 or wh == (300, 300)     # the two rules really differ (except at scale 1)
    finally:
        e.close()


def outside_heavy_head_outputs(seed, n_objects=40, per_object=6):
    """Head outputs shaped like a detector's on a scene whose objects straddle the image border: per object a handful of nearby
    anchors of one class, each decoding to a jittered copy of the object's box -- boxes that overlap each other by more than
    the NMS threshold as decoded and by less (or not at all) once clipped, and the other way round: clipping first and
    clipping last keep different rows (checked below on the oracle twins)."""
    rng = np.random.default_rng(seed)
    anchors = pu.anchors_cs()
    be = np.zeros((2, 1917, 4), np.float32)
    be[:, :, 2:] = -3.0
    lg = (rng.standard_normal((2, 1917, 91)) * 1.0 - 7.0).astype(np.float32)
    for f in range(2):
        for _ in range(n_objects):
            cy, cx = rng.uniform(-0.15, 1.15, 2)
            h, w = rng.uniform(0.15, 0.6, 2)
            cls = int(rng.integers(1, 91))
            near = np.argsort((anchors[:, 0] - cy) ** 2 + (anchors[:, 1] - cx) ** 2)[:per_object * 3]
            for a in rng.choice(near, per_object, replace=False):
                jy, jx = rng.normal(0, 0.02, 2)
                jh, jw = np.exp(rng.normal(0, 0.08, 2))
                ay, ax, ah, aw = anchors[a]
                be[f, a] = [((cy + jy) - ay) / ah * 10, ((cx + jx) - ax) / aw * 10, np.log(h * jh / ah) * 5, np.log(w * jw / aw) * 5]
                lg[f, a, cls] = rng.uniform(0.5, 4.0)
    return be, lg


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_clip_after_nms_matches_its_oracle_twin(dir_clip_after, model_dir_default, seed):
    be, lg = outside_heavy_head_outputs(seed)
    after = conftest.make_engine(dir_clip_after, max_batch=2, dev=True)
    before = conftest.make_engine(model_dir_default, max_batch=2, dev=True)
    try:
        for eng, kw in ((after, dict(clip_after_nms=True)), (before, {})):
            B, S, C, N = eng.stage_postprocess(be, lg)
            rB, rS, rC, rN = pu.oracle_postprocess(be, lg, **kw)
            np.testing.assert_array_equal(N, rN)
            np.testing.assert_array_equal(C, rC)
            np.testing.assert_allclose(S, rS, rtol=0, atol=1e-6)
            np.testing.assert_allclose(B, rB, rtol=0, atol=2e-6)
            assert (B >= 0).all() and (B <= 1).all()
        a = pu.oracle_postprocess(be, lg, clip_after_nms=True)
        b = pu.oracle_postprocess(be, lg)
        assert (a[1] != b[1]).sum() > 50                              # the inputs do tell the two orders apart
    finally:
        after.close()
        before.close()


def test_clip_after_known_answers_on_the_gpu(dir_clip_after):
    """tests/test_oracle_variants.py's hand-made cases through the kernel: encodings chosen so that anchor a decodes to box a."""
    anchors = pu.anchors_cs()                                           # (yc, xc, h, w)
    e = conftest.make_engine(dir_clip_after, max_batch=1, dev=True)
    try:
        def encode(boxes):
            be = np.zeros((1, 1917, 4), np.float32)
            be[0, :, 2:] = -40.0                                        # every other anchor: a box of no size
            lg = np.full((1, 1917, 91), -30.0, np.float32)
            for i, (bx, sc) in enumerate(boxes):
                a = 1000 + 7 * i
                yc, xc, h, w = (bx[0] + bx[2]) / 2, (bx[1] + bx[3]) / 2, bx[2] - bx[0], bx[3] - bx[1]
                ay, ax, ah, aw = anchors[a]
                be[0, a] = [(yc - ay) / ah * 10, (xc - ax) / aw * 10, np.log(h / ah) * 5, np.log(w / aw) * 5]
                lg[0, a, 1] = np.log(sc / (1 - sc))
            return be, lg

        cases = [([([-1.0, -1.0, 1.0, 1.0], 0.9), ([0.0, 0.0, 1.0, 1.0], 0.8)], 2),
                 ([([1.2, 0.0, 1.6, 0.4], 0.9), ([1.15, 0.0, 1.6, 0.4], 0.8), ([0.9, 0.0, 1.6, 0.4], 0.7)], 1)]
        for boxes, want in cases:
            be, lg = encode(boxes)
            B, S, C, N = e.stage_postprocess(be, lg)
            rB, rS, rC, rN = pu.oracle_postprocess(be, lg, clip_after_nms=True)
            assert int(N[0]) == int(rN[0]) == want
            np.testing.assert_array_equal(C, rC)
            np.testing.assert_allclose(S, rS, rtol=0, atol=1e-6)
            np.testing.assert_allclose(B, rB, rtol=0, atol=2e-6)
    finally:
        e.close()


@pytest.mark.parametrize("which", ["half_pixel", "clip_after"])
def test_end_to_end_with_each_option(request, synth_weights, which):
    from watsor_amd.detection.hip_gpu import HipObjectDetector
    from watsor_amd.share import DetectionArray
    d = request.getfixturevalue("dir_" + which)
    oracle = odet.OracleObjectDetector(weights=synth_weights, half_pixel_centers=which == "half_pixel",
                                       clip_after_nms=which == "clip_after")
    with HipObjectDetector(d, 0, max_batch=1, max_width=1280, max_height=720) as det:
        for f in (synthetic_frame(640, 480, 2024), synthetic_frame(1280, 720, 2025)):
            rows = DetectionArray()
            det.detect(f.shape, f, rows)
            got = np.frombuffer(rows, dtype=ROW_DTYPE)
            b, c, s, _, _ = oracle.raw(f)
            ref = odet.rows_as_array(f.shape, b, c, s)
            pairs, missing = pu.match_rows(got, ref, min_score=0.1)
            n_ref = int((ref["confidence"] > 0.1).sum())
            assert n_ref > 0 and len(missing) <= max(1, n_ref // 20)
            assert max(abs(p[3]) for p in pairs) <= 1e-3


def test_latency_schedule_detects_the_same_objects(model_dir, tmp_path):
    """The LATENCY schedule (include/watsor_hip.h: wz_set_schedule; plugin option `schedule`, what the factory picks for up to 4 cameras)
    selects other launch shapes -- other summation orders -- for the same network: the rows of a child process that asks for it through
    the OPTION (not the environment) agree with those of one on the throughput schedule to rounding, on the product library; and a
    process whose shapes are fixed refuses the other schedule."""
    import json
    import os
    import subprocess
    import sys
    script = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from watsor_amd.runtime import HipEngine, ROW_DTYPE\n"
        "from watsor_amd.synth import synthetic_frame\n"
        "e = HipEngine(%r, 0, 8, 640, 480, schedule=sys.argv[1])\n"
        "assert e.schedule == sys.argv[1]\n"
        "try:\n"
        "    HipEngine(%r, 0, 1, 640, 480, schedule='latency' if sys.argv[1] == 'throughput' else 'throughput')\n"
        "    raise SystemExit('the other schedule was accepted')\n"
        "except ValueError:\n"
        "    pass\n"
        "frames = [synthetic_frame(640, 480, 8800 + i) for i in range(8)]\n"
        "rows = [np.zeros(100, ROW_DTYPE) for _ in frames]\n"
        "e.detect_batch(frames, rows)\n"
        "print(json.dumps(dict(nodes=e.graph_nodes(0), label=[r['label'].tolist() for r in rows], conf=[r['confidence'].tolist() for r in rows],"
        " box=[np.stack([r['x_min'], r['y_min'], r['x_max'], r['y_max']], 1).tolist() for r in rows])))\n"
        "e.close()\n" % (conftest.ROOT, os.path.join(model_dir, "mi355x.bin"), os.path.join(model_dir, "mi355x.bin")))
    out = {}
    env = {k: v for k, v in os.environ.items() if k not in ("WZ_SCHEDULE", "WZ_GRAPH")}
    for sched in ("throughput", "latency"):     # (WZ_GRAPH=1: both on captured graphs, so that their node counts can be compared)
        p = subprocess.run([sys.executable, "-c", script, sched], env=dict(env, WZ_GRAPH="1"), capture_output=True, text=True, timeout=240)
        assert p.returncode == 0, p.stderr[-1500:]
        out[sched] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    # left to itself the latency schedule launches kernel by kernel (no graph: the GPU starts on the first kernel while the host issues the
    # rest) -- the same launches, the same rows bit for bit
    p = subprocess.run([sys.executable, "-c", script, "latency"], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-1500:]
    eager = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert eager["nodes"] == 0 and out["latency"]["nodes"] > 0
    assert eager["label"] == out["latency"]["label"] and eager["conf"] == out["latency"]["conf"] and eager["box"] == out["latency"]["box"]
    a, b = out["throughput"], out["latency"]
    if model_dir.program == "default":
        assert b["nodes"] > a["nodes"]                               # the latency schedule keeps the reduce launches of blocks 13 .. 16
    else:
        assert b["nodes"] == a["nodes"]                              # (robust: blocks 13 .. 16 are two launches each under both schedules, k_mbconv_hp2.hip)
    for f in range(8):
        ref = dict(label=np.array(a["label"][f], np.int32), confidence=np.array(a["conf"][f]), box=np.array(a["box"][f], np.int32))
        got = np.zeros(100, ROW_DTYPE)
        got["label"], got["confidence"] = b["label"][f], b["conf"][f]
        bx = np.array(b["box"][f], np.int32)
        got["x_min"], got["y_min"], got["x_max"], got["y_max"] = bx[:, 0], bx[:, 1], bx[:, 2], bx[:, 3]
        pairs, missing = pu.match_rows(got, ref, min_score=0.1)
        n_ref = int((ref["confidence"] > 0.1).sum())
        assert n_ref > 50 and len(missing) <= max(1, n_ref // 20)
        assert max(abs(p[3]) for p in pairs) <= 5e-4


@pytest.mark.parametrize("knobs", [
    dict(WZ_MB_CS_SPLIT16="0"),                                                     # block 16 on the channel-group kernel + reduce launch
    dict(WZ_HP_CS19_LEAN4="0", WZ_HP_CS_OCC4="0", WZ_HP_CS6_LEAN4="0"),             # the 256-register builds of the chunk-split blocks
    dict(WZ_HP_CS19_LEAN4="0", WZ_HP_CS19_NW="8", WZ_MB_CS_MIN_W="11"),             # round 2's shapes
])
def test_earlier_launch_shapes_detect_the_same_objects(model_dir_default, knobs):
    model_dir = model_dir_default
    """The launch shapes of the split blocks and of blocks 13 .. 16 changed several times this round (DESIGN.md section 5): every one of
    them is the same network summed in another order.  A child process on the development library with the earlier shapes selected
    reports more graph nodes where a reduce launch comes back, and rows that agree with the default's to rounding."""
    import json
    import os
    import subprocess
    import sys
    script = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from watsor_amd.runtime import HipEngine, ROW_DTYPE\n"
        "from watsor_amd.synth import synthetic_frame\n"
        "e = HipEngine(%r, 0, 8, 640, 480, dev=True)\n"
        "frames = [synthetic_frame(640, 480, 8900 + i) for i in range(5)]\n"
        "rows = [np.zeros(100, ROW_DTYPE) for _ in frames]\n"
        "e.detect_batch(frames, rows)\n"
        "print(json.dumps(dict(nodes=e.graph_nodes(0), label=[r['label'].tolist() for r in rows], conf=[r['confidence'].tolist() for r in rows],"
        " box=[np.stack([r['x_min'], r['y_min'], r['x_max'], r['y_max']], 1).tolist() for r in rows])))\n"
        "e.close()\n" % (conftest.ROOT, os.path.join(model_dir, "mi355x.bin")))
    out = {}
    for name, env in (("default", {}), ("earlier", knobs)):     # (WZ_GRAPH=1: a lone batch would otherwise go kernel by kernel and report no graph)
        p = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, WZ_GRAPH="1", **env), capture_output=True, text=True, timeout=240)
        assert p.returncode == 0, p.stderr[-1500:]
        out[name] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    a, b = out["default"], out["earlier"]
    assert b["nodes"] >= a["nodes"] and (b["nodes"] > a["nodes"]) == ("WZ_MB_CS_SPLIT16" in knobs or "WZ_MB_CS_MIN_W" in knobs)
    for f in range(5):
        ref = dict(label=np.array(a["label"][f], np.int32), confidence=np.array(a["conf"][f]), box=np.array(a["box"][f], np.int32))
        got = np.zeros(100, ROW_DTYPE)
        got["label"], got["confidence"] = b["label"][f], b["conf"][f]
        bx = np.array(b["box"][f], np.int32)
        got["x_min"], got["y_min"], got["x_max"], got["y_max"] = bx[:, 0], bx[:, 1], bx[:, 2], bx[:, 3]
        pairs, missing = pu.match_rows(got, ref, min_score=0.1)
        n_ref = int((ref["confidence"] > 0.1).sum())
        assert n_ref > 50 and len(missing) <= max(1, n_ref // 20)
        assert max(abs(p[3]) for p in pairs) <= 5e-4


def test_stamps_build_writes_the_same_rows_and_sees_the_lanes_overlap(model_dir_robust, tmp_path):
    """`libwatsor_hip_stamps.so` (make stamps: every kernel records its entry / exit, tools/lane_overlap.py) is a MEASUREMENT build -- what it
    measures must be the product's behaviour: in a child process it writes bit for bit the rows the product library writes for the same
    batch, reports one (entry < exit) pair per launch of the batch, and with the four lanes busy finds more than two kernels in flight."""
    import json
    import os
    import subprocess
    import sys
    stamps = os.path.join(conftest.ROOT, "watsor_amd", "libwatsor_hip_stamps.so")
    if not os.path.isfile(stamps):
        pytest.skip("no stamps build here (make -C watsor_amd/csrc stamps)")
    script = (
        "import sys, json, ctypes as C, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from watsor_amd.runtime import HipEngine, ROW_DTYPE\n"
        "from watsor_amd.synth import synthetic_frame\n"
        "import lane_overlap as lo\n"
        "path = %r\n"
        "frames = [synthetic_frame(640, 480, 9100 + i) for i in range(8)]\n"
        "out = {}\n"
        "for name, dev in (('product', False), ('stamps', True)):\n"
        "    e = HipEngine(path, 0, 8, 640, 480, dev=dev)\n"
        "    rows = [np.zeros(100, ROW_DTYPE) for _ in frames]\n"
        "    e.detect_batch(frames, rows)\n"
        "    out[name] = np.stack(rows).tobytes().hex()\n"
        "    if dev:\n"
        "        buf = (C.c_uint64 * 320)()\n"
        "        k = e._lib.wz_debug_lane_stamps(e._h, 0, buf, 160)\n"
        "        out['launches'] = k\n"
        "        out['ordered'] = all(0 < buf[2 * i] < buf[2 * i + 1] for i in range(k))\n"
        "        out['chain'] = all(buf[2 * i + 1] <= buf[2 * i + 3] for i in range(k - 1))\n"
        "        out['nodes'] = e.graph_nodes(0)\n"
        "    e.close()\n"
        "iv, launches, fps, ms, lanes = lo.collect(path, 120, 20, 8)\n"
        "res = lo.analyse(iv, launches, fps, ms, lanes, 120, 8, None)\n"
        "out['in_flight'] = res['kernels_in_flight_mean_while_busy']; out['dropped'] = res['intervals_dropped']; out['per_step'] = res['launches_per_step']\n"
        "out['named'] = sum(1 for p in res['per_launch'] if p['workgroups'] > 0 and p['wg_per_cu'] > 0)\n"
        "print(json.dumps(out))\n" % (conftest.ROOT, os.path.join(conftest.ROOT, "tools"), os.path.join(model_dir_robust, "mi355x.bin")))
    env = dict(os.environ, WATSOR_HIP_DEV_LIBRARY=stamps, WZ_GRAPH="1")      # (the launch notes are taken when the graph is captured)
    p = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["product"] == out["stamps"]                              # the measurement build computes what the product computes
    assert out["launches"] == out["nodes"] == out["per_step"] == out["named"] and out["launches"] >= 25
    assert out["ordered"] and out["chain"]                              # every launch entered before it left, and left before its successor did
    assert out["dropped"] == 0 and out["in_flight"] > 2.0, out["in_flight"]


def test_kernel_by_kernel_launches_write_the_same_rows_as_the_captured_graph(model_dir, tmp_path):
    """`WZ_GRAPH=0` (one of the four operator settings of the product library, include/watsor_hip.h): the batch is enqueued kernel by kernel
    instead of replaying the captured hipGraph -- the same launches with the same arguments, so the rows are bit-identical, batch after batch
    (the frame descriptors travel as kernel arguments either way; with the graph they are rewritten in the captured node)."""
    import json
    import os
    import subprocess
    import sys
    script = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from watsor_amd.runtime import HipEngine, ROW_DTYPE\n"
        "from watsor_amd.synth import synthetic_frame\n"
        "e = HipEngine(%r, 0, 8, 1280, 720)\n"
        "out = []\n"
        "for k, n in enumerate((8, 3, 8, 1)):\n"
        "    frames = [synthetic_frame(*((640, 480) if (i + k) %% 2 else (1280, 720)), 9300 + 10 * k + i) for i in range(n)]\n"
        "    rows = [np.zeros(100, ROW_DTYPE) for _ in frames]\n"
        "    e.detect_batch(frames, rows)\n"
        "    out.append(np.stack(rows).tobytes().hex())\n"
        "print(json.dumps(dict(nodes=e.graph_nodes(0), rows=out)))\n"
        "e.close()\n" % (conftest.ROOT, os.path.join(model_dir, "mi355x.bin")))
    res = {}
    base = {k: v for k, v in os.environ.items() if k not in ("WZ_GRAPH", "WZ_SCHEDULE")}
    for g in ("1", "0", None):
        env = dict(base, WZ_GRAPH=g) if g is not None else base
        p = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=240)
        assert p.returncode == 0, p.stderr[-1500:]
        res[g] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert res["1"]["nodes"] >= 25 and res["0"]["nodes"] == 0          # a captured graph / none
    assert res["1"]["rows"] == res["0"]["rows"]
    # nothing said (throughput schedule): a batch that finds the other lanes idle -- every batch of this child -- goes kernel by kernel as well; the
    # graph of its size is captured at that first use all the same (the capture stall belongs to the first batch of a size, not to the first
    # moment of contention: ADVICE r5), it is only not replayed
    assert res[None]["nodes"] >= 25 and res[None]["rows"] == res["1"]["rows"]
