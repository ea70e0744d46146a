"""One-shot GPU diagnostic: every stage of the HIP path against the oracle, plus stage timings.

Not a test (nothing asserts): it prints / writes what the parity tests would need to know so that
one `gpurun` call localises a broken kernel.  Output: gpurun_out/diag.txt
"""
from __future__ import annotations

import json
import os
os.environ.setdefault("WATSOR_HIP_DEV", "1")   # tools run on the development library (stage entry points, knobs, profiling)
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from watsor_amd import engine as eng_builder            # noqa: E402
from watsor_amd.synth import synthetic_frame, synthetic_weights   # noqa: E402
from watsor_amd.runtime import HipEngine, ROW_DTYPE, device_count, device_name   # noqa: E402
import parity_utils as pu                              # noqa: E402
from oracle.ssd_mobilenet_v2 import OracleNet, graph_spec   # noqa: E402
from oracle import preprocess as pre, detect as odet   # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "diag.txt"), "w")


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()


def section(name, fn):
    say("\n==== %s ====" % name)
    try:
        fn()
    except Exception:
        say("!! EXCEPTION in", name)
        say(traceback.format_exc())


def main():
    say("devices:", device_count(), [device_name(i) for i in range(device_count())])
    W = synthetic_weights(1234)
    model_dir = "/tmp/wz_diag_model"
    os.makedirs(model_dir, exist_ok=True)
    path = os.path.join(model_dir, "mi355x.bin")
    eng_builder.save_engine(eng_builder.build_engine(W), path)
    net = OracleNet(W)
    frames = [synthetic_frame(640, 480, 1234 + i) for i in range(4)]

    os.environ["WZ_NO_BUFFER_REUSE"] = "1"
    os.environ["WZ_GRAPH"] = "0"
    e = HipEngine(path, 0, 8, 1920, 1080)
    say("engine on", e.device_name)

    def t_pre():
        for (w, h) in ((640, 480), (1280, 720), (1920, 1080), (300, 300), (333, 77)):
            f = synthetic_frame(w, h, 99 + w)
            got = e.stage_preprocess(f)
            ref = pre.preprocess_fp16(f)
            neq = int((got[..., :3].view(np.uint16) != ref.view(np.uint16)).sum())
            d = np.abs(got[..., :3].astype(np.float32) - ref.astype(np.float32)).max()
            say("preprocess %4dx%-4d mismatching halves: %d / %d  max abs diff %.3e  pad-channel nonzero: %d"
                % (w, h, neq, ref.size, d, int((got[..., 3] != 0).sum())))
            if got.shape[-1] == 8:
                lo = (pre.preprocess(f) - ref.astype(np.float32)).astype(np.float16)
                say("      pair input: mismatching lo halves %d  lo pad nonzero %d  max|lo| %.3e (subnormal lo values: %d)"
                    % (int((got[..., 4:7].view(np.uint16) != lo.view(np.uint16)).sum()), int((got[..., 7] != 0).sum()),
                       float(np.abs(lo.astype(np.float32)).max()), int(((np.abs(lo.astype(np.float32)) < 6.1e-5) & (lo != 0)).sum())))
    section("preprocess (bit-exact expected)", t_pre)

    x_half = pu.oracle_input_half(frames[:2])
    state = {}

    def t_fwd():
        be, lg = e.stage_forward(x_half)
        rbe, rlg, T = pu.oracle_forward_from_half(net, x_half, keep=True)
        state.update(be=be, lg=lg, rbe=rbe, rlg=rlg)
        names = [t[0] for t in e.tensors()]
        say("%-44s %-16s %10s %10s %10s" % ("tensor", "shape", "max|ref|", "max abs err", "rel"))
        for idx, (name, h, w, c) in enumerate(e.tensors()):
            if name == "input":
                continue
            got = np.stack([e.stage_read_tensor(idx, f) for f in range(2)]).astype(np.float32)
            ref = T[name]
            err = np.abs(got - ref).max()
            pair = e.tensor_is_pair(idx)
            lim = (5e-4 * np.abs(ref).max() + 1e-4) if pair else (0.05 * np.abs(ref).max() + 0.05)
            say("%-44s %-16s %10.4f %10.6f %10.2e%s%s" % (name, (h, w, c), np.abs(ref).max(), err,
                                                         err / (np.abs(ref).max() + 1e-9), " pair" if pair else "",
                                                         "   <<<<<< BAD" if err > lim else ""))
            if err > lim and pair:
                d = np.abs(got - ref)
                say("      worst at %s; frac of elements over the limit %.4f; per-channel max err (first 16): %s"
                    % (np.unravel_index(d.argmax(), d.shape), float((d > lim).mean()),
                       np.array2string(d.reshape(-1, c).max(0)[:16], precision=4)))
        say("box_enc  max|ref| %.4f  max abs err %.5f" % (np.abs(rbe).max(), np.abs(be - rbe).max()))
        say("logits   max|ref| %.4f  max abs err %.5f  mean abs err %.6f" % (np.abs(rlg).max(), np.abs(lg - rlg).max(),
                                                                             np.abs(lg - rlg).mean()))
        from oracle.postprocess import sigmoid
        ds = np.abs(sigmoid(lg) - sigmoid(rlg))
        say("scores   max abs err %.6f   (over scores > 0.1: %.6f)" % (ds.max(), ds[sigmoid(rlg) > 0.1].max()))
        for i in range(6):
            a = [0, 1083, 1683, 1833, 1887, 1911, 1917]
            sl = slice(a[i], a[i + 1])
            say("  head %d: logits err %.5f  box err %.5f" % (i, np.abs(lg[:, sl] - rlg[:, sl]).max(),
                                                              np.abs(be[:, sl] - rbe[:, sl]).max()))
    section("network forward per tensor (fp16 engine vs fp32 oracle on identical fp16 input)", t_fwd)

    def t_post():
        rbe, rlg = state["rbe"], state["rlg"]
        t0 = time.time()
        B, S, Cc, N = e.stage_postprocess(rbe, rlg)
        t1 = time.time()
        rB, rS, rC, rN = pu.oracle_postprocess(rbe, rlg)
        say("gpu post %.1f ms (incl. copies); num gpu %s ref %s" % ((t1 - t0) * 1e3, N.tolist(), rN.tolist()))
        for f in range(rbe.shape[0]):
            same_cls = int((Cc[f] == rC[f]).sum())
            say("frame %d: classes equal %d/100  max|dscore| %.3e  max|dbox| %.3e" %
                (f, same_cls, np.abs(S[f] - rS[f]).max(), np.abs(B[f] - rB[f]).max()))
            if same_cls != 100:
                bad = np.nonzero(Cc[f] != rC[f])[0][:5]
                for j in bad:
                    say("   row %d gpu (c%d s%.6f) ref (c%d s%.6f)" % (j, Cc[f][j], S[f][j], rC[f][j], rS[f][j]))
        # edge cases
        z = np.zeros_like(rlg[:1]); zb = np.zeros_like(rbe[:1])
        B, S, Cc, N = e.stage_postprocess(zb, z)
        rB, rS, rC, rN = pu.oracle_postprocess(zb, z)
        say("all-equal logits: num gpu %d ref %d  classes equal %d  max|dbox| %.3e max|ds| %.3e" %
            (N[0], rN[0], int((Cc[0] == rC[0]).sum()), np.abs(B - rB).max(), np.abs(S - rS).max()))
        z2 = np.full_like(rlg[:1], -50.0)
        B, S, Cc, N = e.stage_postprocess(zb, z2)
        say("all -50 logits: num gpu %d (expect 0), classes %s scores max %.3g" % (N[0], np.unique(Cc[0]).tolist(), S.max()))
    section("postprocess on the oracle's own fp32 head outputs", t_post)

    def t_rows():
        rbe, rlg = state["rbe"], state["rlg"]
        rB, rS, rC, rN = pu.oracle_postprocess(rbe[:1], rlg[:1])
        for (w, h) in ((640, 480), (1920, 1080), (1, 1)):
            rows = e.stage_rows(w, h, rB[0], rS[0], rC[0])
            ref = odet.rows_as_array((h, w, 3), rB[0], rC[0], rS[0])
            ok = (np.array_equal(rows["label"], ref["label"]) and np.array_equal(rows["confidence"], ref["confidence"])
                  and np.array_equal(np.stack([rows["x_min"], rows["y_min"], rows["x_max"], rows["y_max"]], 1), ref["box"])
                  and not rows["zones"].any())
            say("rows %dx%d bit-exact: %s" % (w, h, ok))
    section("row fill", t_rows)
    e.close()

    os.environ.pop("WZ_NO_BUFFER_REUSE")

    def t_tol():
        # the north star's tolerance, end to end, default program vs --plain-fp16, 9 frames over three resolutions
        many = frames[:3] + [synthetic_frame(1280, 720, 2000 + i) for i in range(3)] + [synthetic_frame(1920, 1080, 3000 + i) for i in range(3)]
        det = odet.OracleObjectDetector(weights=W)
        refs = []
        for f in many:
            b, c, s, _, _ = det.raw(f)
            refs.append(odet.rows_as_array(f.shape, b, c, s))
        ppath = os.path.join(model_dir, "plain", "mi355x.bin")
        eng_builder.save_engine(eng_builder.build_engine(W, hp_upto=-1), ppath)
        for label, pth in (("default (split operands)", path), ("--plain-fp16", ppath)):
            ex = HipEngine(pth, 0, 8, 1920, 1080)
            worst, worst_hi, matched = 0.0, 0.0, 0
            for f, ref in zip(many, refs):
                r = [np.zeros(100, ROW_DTYPE)]
                ex.detect_batch([f], r)
                pairs, missing = pu.match_rows(r[0], ref, min_score=0.0)
                matched += len(pairs)
                worst = max(worst, max(abs(p[3]) for p in pairs))
                worst_hi = max([worst_hi] + [abs(p[3]) for p in pairs if ref["confidence"][p[0]] > 0.1])
            say("%-28s hp_blocks %2d: 9 frames, %d rows matched, max|dscore| %.6f (rows with score > 0.1: %.6f)"
                % (label, ex.hp_blocks, matched, worst, worst_hi))
            ex.close()
    section("score tolerance end to end (north star: 1e-3)", t_tol)

    for graph in ("0", "1"):
        os.environ["WZ_GRAPH"] = graph
        e2 = HipEngine(path, 0, 8, 1920, 1080)

        def t_e2e():
            rows = [np.zeros(100, ROW_DTYPE) for _ in frames]
            ms = e2.detect_batch(frames, rows)
            say("detect_batch(4 frames, host) %.2f ms" % ms)
            det = odet.OracleObjectDetector(weights=W)
            for i, f in enumerate(frames):
                b, c, s, _, _ = det.raw(f)
                ref = odet.rows_as_array(f.shape, b, c, s)
                pairs, missing = pu.match_rows(rows[i], ref, min_score=0.05)
                nref = int((ref["confidence"] > 0.05).sum())
                ds = max([abs(p[3]) for p in pairs], default=0.0)
                exact = int(sum(1 for j in range(100) if rows[i]["label"][j] == ref["label"][j] and
                                abs(rows[i]["confidence"][j] - ref["confidence"][j]) < 2e-3))
                say("frame %d: ref dets(>0.05) %d matched %d missing %d  max|dscore| %.5f  rows equal-in-place %d/100  top: gpu (%d,%.4f) ref (%d,%.4f)"
                    % (i, nref, len(pairs), len(missing), ds, exact, rows[i]["label"][0], rows[i]["confidence"][0],
                       ref["label"][0], ref["confidence"][0]))
        section("end to end, WZ_GRAPH=%s" % graph, t_e2e)

        def t_time():
            d = [e2.upload(f) for f in (frames * 2)]
            ws, hs = [640] * 8, [480] * 8
            for n in (1, 8):
                e2.submit_device(0, d[:n], ws[:n], hs[:n]); e2.wait(0)
                t0 = time.time()
                K = 50
                for k in range(K):
                    e2.submit_device(k % 4, d[:n], ws[:n], hs[:n])
                e2.sync()
                dt = (time.time() - t0) / K
                say("graph=%s batch %d: %.3f ms/step  %.0f frames/s" % (graph, n, dt * 1e3, n / dt))
            if graph == "0":
                st = e2.profile_device(d[:8], ws, hs, reps=10)
                tot = sum(m for _, m in st)
                say("per-stage (batch 8, eager, event-bracketed) total %.3f ms" % tot)
                for name, m in st:
                    say("   %-70s %8.4f ms" % (name, m))
                json.dump(st, open(os.path.join(OUT, "stages_b8.json"), "w"))
        section("timing, WZ_GRAPH=%s" % graph, t_time)
        e2.close()


if __name__ == "__main__":
    main()
