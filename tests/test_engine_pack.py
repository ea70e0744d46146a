"""Engine builder (watsor_amd/engine.py): file format, BN folding, MFMA fragment layout, buffer slots."""
import struct

import numpy as np
import pytest

from watsor_amd import arch, engine


@pytest.fixture(scope="module")
def blob(synth_weights):
    """one op per layer"""
    return engine.build_engine(synth_weights, fuse=False)


@pytest.fixture(scope="module")
def fused_blob(synth_weights):
    """one op per inverted-residual block, the stem still a kernel of its own"""
    return engine.build_engine(synth_weights, fuse_stem=False)


@pytest.fixture(scope="module")
def default_blob(synth_weights):
    """fused blocks, the stem folded into the first one, plain fp16 everywhere (`--plain-fp16`)"""
    return engine.build_engine(synth_weights, hp_upto=-1)


@pytest.fixture(scope="module")
def hp_blob(synth_weights):
    """the default `-p 16` program: the same ops, blocks 0 .. 12 with split (hi + lo) matrix operands"""
    return engine.build_engine(synth_weights)


def parse(blob):
    h = struct.unpack_from("<10I6f6Q12I", blob, 0)
    hdr = dict(magic=h[0], version=h[1], precision=h[2], size=h[3], classes=h[4], anchors=h[5], n_tensors=h[6],
               n_ops=h[7], max_total=h[8], max_per_class=h[9], score_thr=h[10], iou_thr=h[11], scales=h[12:16],
               tensors_off=h[16], ops_off=h[17], anchors_off=h[18], weights_off=h[19], weights_bytes=h[20],
               total=h[21], n_slots=h[22], hp_blocks=h[23])
    tensors = []
    for i in range(hdr["n_tensors"]):
        t = struct.unpack_from("<5i44s", blob, hdr["tensors_off"] + 64 * i)
        tensors.append(dict(h=t[0], w=t[1], c=t[2], slot=t[3], flags=t[4], name=t[5].split(b"\0")[0].decode()))
    ops = []
    keys = ("kind src dst res cin cout ksize stride hin win hout wout pad_t pad_l act out_mode anchor_off "
            "anchors_per_loc n_pad kc").split()
    for i in range(hdr["n_ops"]):
        o = struct.unpack_from("<20i2q8i8q64s", blob, hdr["ops_off"] + engine.OP_RECORD_BYTES * i)
        d = dict(zip(keys, o[:20]))
        d.update(w_off=o[20], b_off=o[21], n_box=o[22], cmid=o[23], cin0=o[24], kc0=o[25], cmid_pad=o[26],
                 nmid_pad=o[27], stem=o[28], stem_pad=o[29], we_off=o[30], be_off=o[31], wd_off=o[32], bd_off=o[33],
                 we_lo_off=o[34], w_lo_off=o[35], flags=o[36], dst2=o[37], name=o[38].split(b"\0")[0].decode())
        ops.append(d)
    return hdr, tensors, ops


def test_header(blob):
    hdr, tensors, ops = parse(blob)
    assert hdr["magic"] == engine.MAGIC and hdr["version"] == engine.FORMAT_VERSION
    assert hdr["total"] == len(blob) and hdr["precision"] == 16 and hdr["size"] == 300
    assert hdr["classes"] == 91 and hdr["anchors"] == 1917 and hdr["n_ops"] == 66
    assert hdr["max_total"] == 100 and abs(hdr["iou_thr"] - 0.6) < 1e-7 and hdr["scales"] == (10.0, 10.0, 5.0, 5.0)
    assert tensors[0]["name"] == "input" and tensors[0]["c"] == 4
    assert sum(1 for o in ops if o["out_mode"] == arch.OUT_HEAD) == 6


def test_shapes_follow_tf_same_padding(blob):
    _, tensors, ops = parse(blob)
    by = {t["name"]: t for t in tensors}
    assert (by["Conv"]["h"], by["Conv"]["c"]) == (150, 32)
    assert (by["expanded_conv_1/depthwise"]["h"], by["expanded_conv_3/depthwise"]["h"]) == (75, 38)
    assert (by["expanded_conv_13/expand"]["h"], by["expanded_conv_13/expand"]["c"]) == (19, 576)
    assert (by["Conv_1"]["h"], by["Conv_1"]["c"]) == (10, 1280)
    pads = {o["name"].split("/")[-2] if o["kind"] == arch.OP_DW else o["name"]: (o["pad_t"], o["pad_l"])
            for o in ops if o["stride"] == 2}
    assert pads["FeatureExtractor/MobilenetV2/Conv"] == (0, 0)          # 300 -> 150: pad only after
    assert pads["expanded_conv_3"] == (1, 1)                            # 75 -> 38
    assert pads["expanded_conv_6"] == (0, 0)                            # 38 -> 19
    assert pads["expanded_conv_13"] == (1, 1)                           # 19 -> 10
    heads = [o for o in ops if o["out_mode"] == arch.OUT_HEAD]
    assert [o["anchor_off"] for o in heads] == [0, 1083, 1683, 1833, 1887, 1911]


@pytest.mark.parametrize("which", ["blob", "fused_blob", "default_blob", "hp_blob"])
def test_slots_never_alias_live_tensors(which, request):
    hdr, tensors, ops = parse(request.getfixturevalue(which))
    last = {}
    for i, o in enumerate(ops):
        last[o["src"]] = i
        if o["res"] >= 0:
            last[o["res"]] = i
    born = {0: -1}
    for i, o in enumerate(ops):
        if o["dst"] >= 0:
            born[o["dst"]] = i
    for a in born:
        for b in born:
            if a < b and tensors[a]["slot"] == tensors[b]["slot"]:
                # lifetimes [born, last] must be disjoint; an op's dst may not reuse its own inputs
                assert last[a] < born[b] or last[b] < born[a], (tensors[a]["name"], tensors[b]["name"])
    assert hdr["n_slots"] <= len(tensors) // 4      # sharing actually happens


def unpack_conv(blob, hdr, o):
    taps = o["ksize"] ** 2
    n = (o["n_pad"] // 16) * taps * o["kc"] * 512
    w = np.frombuffer(blob, np.float16, n, hdr["weights_off"] + o["w_off"]).astype(np.float32)
    w = w.reshape(o["n_pad"] // 16, taps, o["kc"], 4, 16, 8)            # [t][tap][c][g][r][j]
    w = w.transpose(1, 2, 3, 5, 0, 4).reshape(taps, o["kc"] * 32, o["n_pad"])
    return w


def test_weight_fragments_round_trip(blob, synth_weights):
    hdr, tensors, ops = parse(blob)
    prog = arch.build(fuse=False)
    for o, op in zip(ops, prog.ops):
        wf, bf = engine.fold_batch_norm(synth_weights, op)
        bias_n = o["n_pad"] if o["kind"] == arch.OP_CONV else op.cout
        bias = np.frombuffer(blob, np.float32, bias_n, hdr["weights_off"] + o["b_off"])
        np.testing.assert_array_equal(bias[:op.cout], bf.astype(np.float32))
        if o["kind"] == arch.OP_CONV:
            w = unpack_conv(blob, hdr, o)
            ref = wf.reshape(op.k * op.k, op.cin, op.cout).astype(np.float32).astype(np.float16).astype(np.float32)
            np.testing.assert_array_equal(w[:, :op.cin, :op.cout], ref)
            assert not w[:, op.cin:, :].any() and not w[:, :, op.cout:].any() and not bias[op.cout:].any()
        elif o["kind"] == arch.OP_DW:
            w = np.frombuffer(blob, np.float16, 9 * op.cin, hdr["weights_off"] + o["w_off"]).reshape(9, op.cin)
            np.testing.assert_array_equal(w, wf.reshape(9, op.cin).astype(np.float16))


def test_fused_blocks_pack_the_same_weights(blob, fused_blob, synth_weights):
    """OP_MBCONV records carry exactly the folded weights of the three layers they replace."""
    hdr, tensors, ops = parse(blob)
    fhdr, ftensors, fops = parse(fused_blob)
    assert fhdr["n_ops"] == 34 and sum(1 for o in fops if o["kind"] == arch.OP_MBCONV) == 17
    by_name = {o["name"]: o for o in ops}
    fprog = arch.build(fuse=True, fuse_stem=False)
    names = {t["name"] for t in ftensors}
    assert "expanded_conv_13/expand" in names and "expanded_conv_12/expand" not in names   # SSD tap stays in HBM
    for o, op in zip(fops, fprog.ops):
        if o["kind"] != arch.OP_MBCONV:
            assert o["w_off"] >= 0
            continue
        pj = by_name[o["name"] + "/project"]
        dw = by_name[o["name"] + "/depthwise"]
        assert (o["cin"], o["cout"], o["n_pad"], o["kc"], o["stride"]) == (pj["cin"], pj["cout"], pj["n_pad"], pj["kc"], dw["stride"])
        assert (o["hin"], o["hout"], o["pad_t"], o["pad_l"]) == (dw["hin"], dw["hout"], dw["pad_t"], dw["pad_l"])
        assert o["cmid"] == dw["cin"] and o["cmid_pad"] == (o["cmid"] + 31) // 32 * 32 and o["kc"] == o["cmid_pad"] // 32
        np.testing.assert_array_equal(unpack_conv(fused_blob, fhdr, dict(o, ksize=1)), unpack_conv(blob, hdr, pj))
        wd = np.frombuffer(fused_blob, np.float16, 9 * o["cmid_pad"], fhdr["weights_off"] + o["wd_off"]).reshape(9, -1)
        ref = np.frombuffer(blob, np.float16, 9 * dw["cin"], hdr["weights_off"] + dw["w_off"]).reshape(9, -1)
        np.testing.assert_array_equal(wd[:, :o["cmid"]], ref)
        assert not wd[:, o["cmid"]:].any()
        bd = np.frombuffer(fused_blob, np.float32, o["cmid_pad"], fhdr["weights_off"] + o["bd_off"])
        np.testing.assert_array_equal(bd[:o["cmid"]], np.frombuffer(blob, np.float32, dw["cin"], hdr["weights_off"] + dw["b_off"]))
        if o["cin0"]:
            ex = by_name[o["name"] + "/expand"]
            assert (o["cin0"], o["kc0"], o["nmid_pad"]) == (ex["cin"], ex["kc"], ex["n_pad"]) and ex["cout"] == o["cmid"]
            np.testing.assert_array_equal(
                unpack_conv(fused_blob, fhdr, dict(ksize=1, n_pad=o["nmid_pad"], kc=o["kc0"], w_off=o["we_off"])),
                unpack_conv(blob, hdr, ex))
            np.testing.assert_array_equal(
                np.frombuffer(fused_blob, np.float32, o["nmid_pad"], fhdr["weights_off"] + o["be_off"]),
                np.frombuffer(blob, np.float32, ex["n_pad"], hdr["weights_off"] + ex["b_off"]))
        else:
            assert o["name"].endswith(("expanded_conv", "expanded_conv_13"))


def test_stem_folded_into_the_first_block(default_blob, fused_blob, synth_weights):
    """Default program: op 0 reads the 300x300x4 input, its expand stage is the stem conv as a K = 27 (-> 32) GEMM."""
    hdr, tensors, ops = parse(default_blob)
    _, _, fops = parse(fused_blob)
    assert hdr["n_ops"] == 32 and "Conv" not in {t["name"] for t in tensors}
    o, blk0 = ops[0], fops[1]
    assert o["kind"] == arch.OP_MBCONV and o["stem"] == 1 and o["stem_pad"] == 0 and tensors[o["src"]]["name"] == "input"
    assert (o["cin0"], o["kc0"], o["nmid_pad"], o["cmid"], o["hin"], o["hout"], o["stride"]) == (32, 1, 32, 32, 150, 150, 1)
    assert (o["cout"], o["n_pad"], o["kc"], o["pad_t"]) == (blk0["cout"], blk0["n_pad"], blk0["kc"], blk0["pad_t"])
    prog = arch.build(fuse=False)
    wf, bf = engine.fold_batch_norm(synth_weights, prog.ops[0])                 # [3,3,3,32]
    w = unpack_conv(default_blob, hdr, dict(ksize=1, n_pad=32, kc=1, w_off=o["we_off"]))   # [1][32][32]
    ref = wf.reshape(27, 32).astype(np.float32).astype(np.float16).astype(np.float32)
    np.testing.assert_array_equal(w[0, :27, :], ref)
    assert not w[0, 27:, :].any()
    np.testing.assert_array_equal(np.frombuffer(default_blob, np.float32, 32, hdr["weights_off"] + o["be_off"]), bf.astype(np.float32))
    rest = [b for b in fops[2:] if not b["name"].endswith("expanded_conv_13/expand")]   # (block 13's expand conv: next test)
    assert len(rest) == len(ops) - 1 == len(fops) - 3
    for a, b in zip(ops[1:], rest):                                            # everything behind it is unchanged
        assert (a["kind"], a["name"], a["cin"], a["cout"]) == (b["kind"], b["name"], b["cin"], b["cout"])


def test_block_13_stores_the_first_feature_map_itself(default_blob, hp_blob, fused_blob, synth_weights):
    """Stem-folded programs: block 13 has an expand stage of its own (from block 12's output) and a SECOND output, its expanded tensor
    19 x 19 x 576 -- the first SSD feature map, which BoxPredictor_0 reads -- instead of a separate 1x1 conv in front of it (one launch
    less per batch).  The fuse_stem=False program keeps that conv; both hold the same weights."""
    _, ftensors, fops = parse(fused_blob)
    tap = next(o for o in fops if o["name"].endswith("expanded_conv_13/expand"))
    fb13 = next(o for o in fops if o["name"].endswith("expanded_conv_13"))
    assert tap["kind"] == arch.OP_CONV and fb13["cin0"] == 0 and fb13["dst2"] == 0 and ftensors[fb13["src"]]["name"] == "expanded_conv_13/expand"
    for blob in (default_blob, hp_blob):
        hdr, tensors, ops = parse(blob)
        assert not [o for o in ops if o["name"].endswith("expanded_conv_13/expand")]
        b13 = next(o for o in ops if o["name"].endswith("expanded_conv_13"))
        assert [o["dst2"] for o in ops if o is not b13] == [0] * (len(ops) - 1)
        t2 = tensors[b13["dst2"] - 1]
        assert (t2["name"], t2["h"], t2["w"], t2["c"], t2["flags"]) == ("expanded_conv_13/expand", 19, 19, 576, 0)
        assert (b13["cin0"], b13["kc0"], b13["nmid_pad"], b13["cmid"], b13["stride"], b13["hin"], b13["hout"]) == (96, 3, 576, 576, 2, 19, 10)
        assert tensors[b13["src"]]["name"] == "expanded_conv_12/output" and not (b13["flags"] & 1)
        head0 = next(o for o in ops if o["name"] == "BoxPredictor_0")
        assert head0["src"] == b13["dst2"] - 1
        # the second output has a buffer of its own while block 13 .. BoxPredictor_0 run
        live = [o for o in ops[ops.index(b13):ops.index(head0) + 1]]
        assert all(tensors[o["dst"]]["slot"] != t2["slot"] for o in live if o["dst"] >= 0 and o is not head0)
        assert tensors[b13["src"]]["slot"] != t2["slot"]
        w = unpack_conv(blob, hdr, dict(ksize=1, n_pad=b13["nmid_pad"], kc=b13["kc0"], w_off=b13["we_off"]))
        np.testing.assert_array_equal(w, unpack_conv(fused_blob, parse(fused_blob)[0], dict(tap, ksize=1)))
    assert arch.build(tap_in_block=False).ops[13].kind == arch.OP_CONV


def test_split_operand_blocks_carry_hi_and_lo_weights(hp_blob, default_blob, synth_weights):
    """Default `-p 16` program: blocks 0 .. 12 hold every GEMM weight as hi + lo halves whose sum is the folded fp64 weight
    to ~2^-22 (the expand stage with the 1/6 of the unorm16 chunk buffer folded in), fp32 depthwise weights carrying
    6/65535, and the tensors between them are pairs."""
    hdr, tensors, ops = parse(hp_blob)
    phdr, ptensors, pops = parse(default_blob)
    assert hdr["hp_blocks"] == arch.HP_LAST_BLOCK + 1 and phdr["hp_blocks"] == 0
    assert [t["name"] for t in tensors] == [t["name"] for t in ptensors]
    assert not any(t["flags"] for t in ptensors) and not any(o["flags"] for o in pops)
    prog = arch.build(hp_upto=arch.HP_LAST_BLOCK)
    n_hp = 0
    for o, po, op in zip(ops, pops, prog.ops):
        assert (o["kind"], o["name"], o["cin"], o["cout"], o["n_pad"], o["kc"]) == (po["kind"], po["name"], po["cin"], po["cout"], po["n_pad"], po["kc"])
        if not op.hp:
            assert o["flags"] == 0 and not tensors[o["src"]]["flags"]
            continue
        n_hp += 1
        assert o["flags"] & 1 and tensors[o["src"]]["flags"] == 1
        assert bool(o["flags"] & 2) == bool(tensors[o["dst"]]["flags"]) == (op.block < arch.HP_LAST_BLOCK)
        parts = list(op.parts)
        ex = parts.pop(0) if (op.stem or op.cin0) else None
        dw, pj = parts
        # project: hi + lo == folded weight (to 2^-21 of its magnitude), hi == the plain program's fp16 weight
        wf, bf = engine.fold_batch_norm(synth_weights, pj)
        hi = unpack_conv(hp_blob, hdr, dict(o, ksize=1)).astype(np.float64)
        lo = unpack_conv(hp_blob, hdr, dict(o, ksize=1, w_off=o["w_lo_off"])).astype(np.float64)
        ref = wf.reshape(1, pj.cin, pj.cout)
        plain = unpack_conv(default_blob, phdr, dict(po, ksize=1)).astype(np.float64)   # (rounded through fp32: a rare ulp apart)
        assert np.abs(hi - plain).max() <= np.abs(plain).max() * 2.0 ** -10 and (hi != plain).mean() < 1e-3
        assert np.abs(hi[:, :pj.cin, :pj.cout] + lo[:, :pj.cin, :pj.cout] - ref).max() <= np.abs(ref).max() * 2.0 ** -21
        assert np.abs(lo).max() <= np.abs(hi).max() * 2.0 ** -10 and lo.any()
        # expand (or stem): the same with the factor 1/6
        wf, bf = engine.fold_batch_norm(synth_weights, ex)
        kin = 32 if op.stem else ex.cin
        if op.stem:                  # K order of the stem gather: tap*4 + c, the ninth tap in the pad slots of taps 0 .. 2
            rows = engine.stem_k_rows(wf)
            assert np.array_equal(rows[5], wf[0, 1, 1]) and np.array_equal(rows[7], wf[2, 2, 1]) and not rows[15].any()
            assert np.count_nonzero(np.abs(rows).sum(1)) == 27
            ref = rows.reshape(1, 32, ex.cout) / 6.0
        else:
            ref = wf.reshape(1, kin, ex.cout) / 6.0
        d = dict(ksize=1, n_pad=o["nmid_pad"], kc=o["kc0"])
        hi = unpack_conv(hp_blob, hdr, dict(d, w_off=o["we_off"])).astype(np.float64)
        lo = unpack_conv(hp_blob, hdr, dict(d, w_off=o["we_lo_off"])).astype(np.float64)
        assert np.abs(hi[:, :kin, :ex.cout] + lo[:, :kin, :ex.cout] - ref).max() <= np.abs(ref).max() * 2.0 ** -21
        assert not hi[:, kin:, :].any() and not lo[:, kin:, :].any()
        be = np.frombuffer(hp_blob, np.float32, o["nmid_pad"], hdr["weights_off"] + o["be_off"])
        np.testing.assert_allclose(be[:ex.cout], bf / 6.0, rtol=1e-7, atol=0)
        # depthwise: fp32, scaled by 6 / 65535
        wf, bf = engine.fold_batch_norm(synth_weights, dw)
        wd = np.frombuffer(hp_blob, np.float32, 9 * o["cmid_pad"], hdr["weights_off"] + o["wd_off"]).reshape(9, -1)
        np.testing.assert_allclose(wd[:, :o["cmid"]], wf.reshape(9, -1) * (6.0 / 65535.0), rtol=1e-7, atol=0)
        assert not wd[:, o["cmid"]:].any()
    assert n_hp == arch.HP_LAST_BLOCK + 1
    assert tensors[0]["flags"] == 1                      # the network input is a pair, too
    with pytest.raises(ValueError):
        arch.build(fuse_stem=False, hp_upto=3)


def test_fold_matches_oracle_fold(synth_weights):
    """Product-side BN folding / head fusion vs the oracle's independent description of the graph."""
    from oracle import ssd_mobilenet_v2 as net
    prog = arch.build(fuse=False)
    spec = {s.name: s for s in net.graph_spec()}
    assert len(spec) == 72 and len(prog.ops) == 66
    for op in prog.ops:
        w2, b2 = engine.fold_batch_norm(synth_weights, op)
        if op.out_mode == arch.OUT_HEAD:
            sb, sc = spec[op.scope + "/BoxEncodingPredictor"], spec[op.scope + "/ClassPredictor"]
            assert (sb.cin, sb.cout, sc.cout, sb.k) == (op.cin, op.n_box, op.cout - op.n_box, op.k)
            wb, bb = net.fold_bn(synth_weights, sb)
            wc, bc = net.fold_bn(synth_weights, sc)
            np.testing.assert_array_equal(np.concatenate([wb, wc], 3), w2.astype(np.float32))
            np.testing.assert_array_equal(np.concatenate([bb, bc]), b2.astype(np.float32))
            continue
        s_ = spec[op.scope]
        assert (s_.cin, s_.cout, s_.stride, s_.k) == (op.cin, op.cout, op.stride, op.k)
        assert (s_.res is not None) == (op.res is not None) and s_.relu6 == (op.act == arch.ACT_RELU6)
        w1, b1 = net.fold_bn(synth_weights, s_)
        np.testing.assert_allclose(w1, w2.astype(np.float32), rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(b1, b2.astype(np.float32), rtol=1e-6, atol=1e-7)


def test_anchor_table_matches_oracle(blob):
    from oracle import postprocess as post
    hdr, _, _ = parse(blob)
    a = np.frombuffer(blob, np.float32, 1917 * 4, hdr["anchors_off"]).reshape(1917, 4)
    np.testing.assert_array_equal(a, post.anchors_center_size(post.generate_anchors()))


def test_cli_and_errors(tmp_path, synth_weights):
    np.savez(tmp_path / "m.npz", **synth_weights)
    out = tmp_path / "model" / "mi355x.bin"
    assert engine.main(["-i", str(tmp_path / "m.npz"), "-o", str(out), "-p", "16"]) == 0
    assert out.stat().st_size > 30e6
    with pytest.raises(FileNotFoundError):
        engine.load_weights(str(tmp_path / "missing.npz"))
    bad = dict(synth_weights)
    bad.pop("FeatureExtractor/MobilenetV2/Conv/weights")
    with pytest.raises(KeyError):
        engine.build_engine(bad)
    with pytest.raises(ValueError):
        engine.build_engine(synth_weights, precision=8)


def test_fp32_engine_packs_unrounded_weights(synth_weights):
    """-p 32: one op per layer, fp32 weights in the float4-per-lane fragment order of csrc/k_f32.hip."""
    blob32 = engine.build_engine(synth_weights, precision=32)
    hdr, tensors, ops = parse(blob32)
    assert hdr["precision"] == 32 and hdr["n_ops"] == 66 and not any(o["kind"] == arch.OP_MBCONV for o in ops)
    prog = arch.build(fuse=False)
    for o, op in zip(ops, prog.ops):
        wf, bf = engine.fold_batch_norm(synth_weights, op)
        if o["kind"] == arch.OP_CONV:
            taps = o["ksize"] ** 2
            assert o["kc"] == (op.cin + 15) // 16
            n = (o["n_pad"] // 16) * taps * o["kc"] * 256
            w = np.frombuffer(blob32, np.float32, n, hdr["weights_off"] + o["w_off"])
            w = w.reshape(o["n_pad"] // 16, taps, o["kc"], 4, 16, 4)            # [t][tap][c][g][r][j]
            w = w.transpose(1, 2, 3, 5, 0, 4).reshape(taps, o["kc"] * 16, o["n_pad"])
            np.testing.assert_array_equal(w[:, :op.cin, :op.cout], wf.reshape(taps, op.cin, op.cout).astype(np.float32))
            assert not w[:, op.cin:, :].any() and not w[:, :, op.cout:].any()
        elif o["kind"] == arch.OP_DW:
            w = np.frombuffer(blob32, np.float32, 9 * op.cin, hdr["weights_off"] + o["w_off"]).reshape(9, op.cin)
            np.testing.assert_array_equal(w, wf.reshape(9, op.cin).astype(np.float32))


def test_builder_reports_the_channel_spread_and_warns_beyond_what_was_validated(tmp_path, synth_weights, capsys):
    """the -p 16 program's 1e-3 margin was established on He-initialised weights.  The builder measures what
    the simulation (tools/err_budget.py, SPREAD=...) shows to matter and says so."""
    from watsor_amd.synth import spread_channel_scales
    base = engine.channel_spread_decades(synth_weights)
    assert 0.1 < base < 0.35
    wide = spread_channel_scales(synth_weights, 1.5)
    assert 1.1 < engine.channel_spread_decades(wide) < 1.6
    np.savez(str(tmp_path / "wide.npz"), **wide)
    out = str(tmp_path / "m" / "mi355x.bin")
    assert engine.main(["-i", str(tmp_path / "wide.npz"), "-o", out, "--robust", "off"]) == 0
    cap = capsys.readouterr()
    assert "WARNING" in cap.err and "-p 32" in cap.err and "--robust on" in cap.err and "decades" in cap.out
    with pytest.raises(ValueError):
        engine.main(["-i", str(tmp_path / "wide.npz"), "-o", out, "--robust", "off", "--precision-check", "error"])
    assert engine.main(["-i", str(tmp_path / "wide.npz"), "-o", out, "--precision-check", "error"]) == 0    # (auto: the robust program)
    capsys.readouterr()
    assert engine.main(["-i", str(tmp_path / "wide.npz"), "-o", out, "-p", "32"]) == 0
    assert "WARNING" not in capsys.readouterr().err
    assert engine.main(["-i", "synthetic", "-o", out]) == 0
    assert "WARNING" not in capsys.readouterr().err


def test_robust_program_packs_all_blocks_split_and_float_form_scaled(hp_blob, synth_weights):
    """`build_engine(robust=True)`: all 17 blocks split-operand with flag 4 (float-form chunk buffer t = (x / 6) T, T = 2^-120 (2 - 2^-13):
    depthwise taps carry 6 * 2^60 / (2 - 2^-13), the kernel scales the tap sum back by 2^60; the bias is the plain one), every tensor between them a pair; block 13 reads block 12's pair
    output and stores the first SSD feature map as its second output (plain fp16), like the default program's block 13; block 16
    stores its output twice (flag 8, a plain 640-channel tensor) and Conv_1 multiplies it with [hi | lo] halves of its weights."""
    rb = engine.build_engine(synth_weights, robust=True)
    hdr, tensors, ops = parse(rb)
    dhdr, dtensors, dops = parse(hp_blob)
    assert hdr["hp_blocks"] == arch.HP_ALL_BLOCKS + 1 == 17
    assert [o["name"] for o in ops] == [o["name"] for o in dops] and [t["name"] for t in tensors] == [t["name"] for t in dtensors]
    prog = arch.build(hp_upto=arch.HP_ALL_BLOCKS)
    blocks = [(o, op) for o, op in zip(ops, prog.ops) if o["kind"] == arch.OP_MBCONV]
    assert len(blocks) == 17
    for o, op in blocks:
        assert o["flags"] & 1 and tensors[o["src"]]["flags"] == 1 and o["cin0"] > 0
        assert bool(o["flags"] & 4) == (op.block <= 9)           # the float form on the large maps, linear behind them (engine.build_engine: float_form_upto)
        assert bool(o["flags"] & 2) == bool(tensors[o["dst"]]["flags"]) == (op.block < 16)
        assert bool(o["flags"] & 8) == (op.block == 16) and tensors[o["dst"]]["c"] == (640 if op.block == 16 else o["cout"])
        dw = op.parts[-2]
        wf, bf = engine.fold_batch_norm(synth_weights, dw)
        wd = np.frombuffer(rb, np.float32, 9 * o["cmid_pad"], hdr["weights_off"] + o["wd_off"]).reshape(9, -1)
        bd = np.frombuffer(rb, np.float32, o["cmid_pad"], hdr["weights_off"] + o["bd_off"])
        T, S = 2.0 ** -120 * (2.0 - 2.0 ** -13), 2.0 ** 60
        if not o["flags"] & 4:                                   # linear buffer: 6 / 65535 in the taps, the bias as it is
            np.testing.assert_allclose(wd[:, :o["cmid"]], wf.reshape(9, -1) * (6.0 / 65535.0), rtol=1e-6, atol=0)
            np.testing.assert_allclose(bd[:o["cmid"]], bf, rtol=1e-6, atol=1e-7)
            continue
        np.testing.assert_allclose(wd[:, :o["cmid"]], wf.reshape(9, -1) * (6.0 * S / (2.0 - 2.0 ** -13)), rtol=1e-6, atol=0)
        assert np.isfinite(wd).all()
        np.testing.assert_allclose(bd[:o["cmid"]], bf, rtol=1e-6, atol=1e-7)
        # what the kernel computes from stored codes: S * sum_t wd_t * t_t + bd with t = (x / 6) T  ==  sum_t w_t * x_t + b  (padding: x = 0
        # is t = 0 is code 0)
        x = np.random.default_rng(op.block).uniform(0, 6, (9, o["cmid"]))
        x[:3] = 0.0
        got = S * (wd[:, :o["cmid"]].astype(np.float64) * (x / 6.0 * T)).sum(0) + bd[:o["cmid"]]
        np.testing.assert_allclose(got, (wf.reshape(9, -1) * x).sum(0) + bf, rtol=0, atol=2e-5)
        # the code itself: bits 10 .. 25 of the fp32 pattern of t, rounded; full scale is code 65535, 2^-7 of it code 8192 (where the subnormal range ends)
        code = lambda z: (int(np.float32(z * T).view(np.uint32)) + 512) >> 10                       # noqa: E731
        assert code(1.0) == 65535 and code(0.0) == 0 and code(2.0 ** -7 * (1 - 2.0 ** -14)) in (8191, 8192)
        back = lambda c: float(np.uint32(c << 10).view(np.float32)) / T                                # noqa: E731
        for z in (1e-4, 3e-3, 0.03, 0.4, 0.9999):
            assert abs(back(code(z)) - z) <= max(z * 2.0 ** -13, 2.0 ** -20)
    b13 = next(o for o, op in blocks if op.block == 13)
    assert (b13["cin0"], b13["kc0"], b13["cmid"], b13["cout"], b13["stride"]) == (96, 3, 576, 160, 2)
    assert tensors[b13["src"]]["name"] == "expanded_conv_12/output" and tensors[b13["src"]]["flags"] == 1
    t2 = tensors[b13["dst2"] - 1]
    assert (t2["name"], t2["c"], t2["flags"]) == ("expanded_conv_13/expand", 576, 0)
    assert next(o for o in ops if o["name"] == "BoxPredictor_0")["src"] == b13["dst2"] - 1
    # Conv_1: K = 640 = [hi halves | lo halves] of the folded weights over the two copies of block 16's output
    c1, d1 = next(o for o in ops if o["name"].endswith("Conv_1")), next(o for o in dops if o["name"].endswith("Conv_1"))
    assert (c1["cin"], c1["kc"], d1["cin"], d1["kc"]) == (640, 20, 320, 10) and tensors[c1["src"]]["c"] == 640
    w, _ = engine.fold_batch_norm(synth_weights, next(op for op in prog.ops if op.scope.endswith("Conv_1")))
    frag = np.frombuffer(rb, np.float16, c1["n_pad"] * 640, hdr["weights_off"] + c1["w_off"]).reshape(c1["n_pad"] // 16, 20, 64, 8)
    # lane l of N-tile t, chunk q holds W[k = q * 32 + (l >> 4) * 8 + j][n = t * 16 + (l & 15)] (csrc/wz_program.h)
    k, n = 5 * 32 + 2 * 8 + 3, 7 * 16 + 9
    hi, lo = float(frag[7, 5, 2 * 16 + 9, 3]), float(frag[7, 15, 2 * 16 + 9, 3])
    assert hi == float(np.float16(w[0, 0, k, n])) and abs(hi + lo - w[0, 0, k, n]) <= 2.0 ** -21 * abs(w[0, 0, k, n]) + 1e-9
    # ... none of the default program's blocks carries flag 4; its blocks 13 .. 16 are plain
    assert not any(o["flags"] & 4 for o in dops)
    assert [bool(o["flags"] & 1) for o in dops if o["kind"] == arch.OP_MBCONV] == [True] * 13 + [False] * 4
    with pytest.raises(ValueError):
        engine.build_engine(synth_weights, precision=32, robust=True)
    with pytest.raises(ValueError):
        engine.build_engine(synth_weights, hp_upto=-1, robust=True)


def test_builder_picks_the_robust_program_by_channel_spread(tmp_path, synth_weights, capsys):
    from watsor_amd.synth import spread_channel_scales
    np.savez(str(tmp_path / "wide.npz"), **spread_channel_scales(synth_weights, 1.0))
    np.savez(str(tmp_path / "wider.npz"), **spread_channel_scales(synth_weights, 2.5))
    out = str(tmp_path / "m" / "mi355x.bin")

    def hp_blocks():
        return parse(open(out, "rb").read())[0]["hp_blocks"]

    assert engine.main(["-i", str(tmp_path / "wide.npz"), "-o", out]) == 0           # auto: 1.0 decades -> robust, inside what it was validated for
    cap = capsys.readouterr()
    assert hp_blocks() == 17 and "robust" in cap.out and "WARNING" not in cap.err
    assert engine.main(["-i", str(tmp_path / "wide.npz"), "-o", out, "--robust", "off"]) == 0
    cap = capsys.readouterr()
    assert hp_blocks() == 13 and "WARNING" in cap.err
    assert engine.main(["-i", "synthetic", "-o", out]) == 0                          # one scale: the default program
    assert hp_blocks() == 13 and "WARNING" not in capsys.readouterr().err
    assert engine.main(["-i", "synthetic", "-o", out, "--robust", "on"]) == 0
    assert hp_blocks() == 17
    assert engine.main(["-i", str(tmp_path / "wider.npz"), "-o", out]) == 0           # beyond the robust program's range as well: it says so
    cap = capsys.readouterr()
    assert hp_blocks() == 17 and "WARNING" in cap.err and "-p 32" in cap.err
    with pytest.raises(ValueError):
        engine.main(["-i", "synthetic", "-o", out, "--robust", "on", "--plain-fp16"])


def test_float_form_beyond_the_blocks_that_have_it_is_refused(synth_weights):
    """blocks 13 .. 16 exist with the linear chunk buffer only -- an engine that asked for the float form there would be built
    happily and then refused by the runtime ("no split-operand kernel took op").  The builder says so instead."""
    import pytest
    from watsor_amd import engine
    with pytest.raises(ValueError, match="float-form chunk buffer exists for blocks 0 .. 12"):
        engine.build_engine(synth_weights, robust=True, float_form_upto=13)
    assert engine.FLOAT_FORM_LAST_BLOCK == 12
