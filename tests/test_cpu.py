"""CPU-side tests (no GPU): the oracle against the reference-generated golden vectors, host logic (BN folding, weight
packing, state_dict schema, sharding, detection all_gather over gloo) and the C ABI surface of libccnet_hip.so."""
import ctypes
import glob
import importlib
import math
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "net*_b*_*.npz")))
KINDS = ("net", "net_deploy", "net_amff", "net_small_target", "net_old")


def _fx(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: z[k] for k in z.files}


# ----------------------------------------------------------------------------- oracle pinned to the reference's outputs
@pytest.mark.parametrize("name", [f for f in FIXTURES if "_b1_" in f])
def test_oracle_forward_matches_reference_fixture(name):
    from ccnet_amd import synth
    from oracle import ccnet_oracle as orc
    fx = _fx(name)
    kind, B, H, W = str(fx["kind"]), int(fx["B"]), int(fx["H"]), int(fx["W"])
    net = importlib.import_module("ccnet_amd.network." + kind).Net(float(fx["depth"]), float(fx["width"]), int(fx["nc"]))
    sd = synth.synth_state_dict(net.state_dict(), 0, kind)
    assert abs(sum(synth.tensor_checksum(v) for v in sd.values()) - float(fx["weight_checksum"])) <= 1e-9 * abs(float(fx["weight_checksum"]))
    x = synth.synth_images(B, H, W, 0)
    taps = {}
    heads = orc.forward(kind, sd, x, float(fx["depth"]), float(fx["width"]), "silu", taps)
    for k, h in enumerate(heads):
        ref = torch.from_numpy(fx["head%d" % k])
        assert float((h - ref).abs().max() / ref.abs().max()) < 2e-5
    for key in fx:
        if key.startswith("tap_"):
            t = taps[key[4:]][:, ::3, ::2, ::2]
            ref = torch.from_numpy(fx[key])
            assert float((t - ref).abs().max() / ref.abs().max().clamp_min(1e-12)) < 2e-5, key


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_decode_and_post_process_match_reference_fixture(name):
    from oracle import ccnet_oracle as orc
    fx = _fx(name)
    H, W, nc = int(fx["H"]), int(fx["W"]), int(fx["nc"])
    heads = []
    while "head%d" % len(heads) in fx:
        heads.append(torch.from_numpy(fx["head%d" % len(heads)]))
    dec = orc.decode_outputs(heads, [H, W])
    # sigmoid/exp come from ATen's CPU kernels: identical build -> identical bits; allow 1 ulp across machines
    assert torch.allclose(dec, torch.from_numpy(fx["decoded"]), rtol=3e-7, atol=1e-9)
    info = [list(map(int, r)) for r in fx["image_info"]]
    got = orc.post_process(torch.from_numpy(fx["decoded"]), nc, [H, W], info, float(fx["conf"]), float(fx["nms"]))
    counts = list(fx["det_counts"])
    assert [g.shape[0] for g in got] == counts
    assert np.array_equal(np.concatenate(got, 0), fx["dets"])


def test_oracle_nms_known_answers():
    from oracle import ccnet_oracle as orc

    def run(boxes, scores, cls, thr):
        return list(orc.batched_nms_reference(np.asarray(boxes, np.float32).reshape(-1, 4), np.asarray(scores, np.float32),
                                              np.asarray(cls, np.float32), thr))

    assert run([[0, 0, 1, 1], [0, 0, 1, 1]], [0.9, 0.8], [0, 0], 0.5) == [0]             # identical boxes
    assert run([[0, 0, 1, 1], [1, 0, 2, 1]], [0.9, 0.8], [0, 0], 0.5) == [0, 1]          # touching, IoU 0
    assert run([[0, 0, 2, 1], [0, 0, 1, 1]], [0.9, 0.8], [0, 0], 0.5) == [0, 1]          # IoU == thr is kept ('>')
    assert run([[0, 0, 2, 1], [0, 0, 1, 1]], [0.8, 0.9], [0, 0], 0.49) == [1]            # order follows the scores
    assert run([[0, 0, 1, 1], [0, 0, 1, 1]], [0.9, 0.8], [0, 1], 0.5) == [0, 1]          # other class never suppresses
    assert run([[0, 0, 1, 1]] * 3, [0.7, 0.7, 0.7], [0, 0, 0], 0.5) == [0]               # ties: lower index first
    assert run([[.5, .5, .5, .5]] * 2, [0.9, 0.8], [0, 0], 0.5) == [0, 1]                # zero area: NaN IoU keeps both
    assert run([], [], [], 0.5) == []


def test_oracle_post_process_empty_and_threshold_edge():
    from oracle import ccnet_oracle as orc
    pred = torch.zeros(2, 4, 11)
    pred[..., 2:4] = 0.1
    pred[0, 1, 4], pred[0, 1, 5] = 0.5, 1.0            # obj*cls == conf exactly: kept ('>=')
    out = orc.post_process(pred, 6, [64, 64], [[0, 64, 64], [1, 64, 64]], 0.5, 0.4)
    assert out[0].shape == (1, 7) and out[1].shape == (0, 7) and out[1].dtype == np.float32
    assert orc.post_process(torch.zeros(1, 0, 11), 6, [64, 64], [[0, 64, 64]]) == [None]


# ----------------------------------------------------------------------------- synthetic data generator
def test_synth_is_deterministic_and_keyed():
    from ccnet_amd import synth
    a = synth.synth_tensor("dark2.0.conv.conv.weight", (64, 128, 3, 3), 0)
    b = synth.synth_tensor("dark2.0.conv.conv.weight", (64, 128, 3, 3), 0)
    c = synth.synth_tensor("dark3.0.conv.conv.weight", (64, 128, 3, 3), 0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(a.std()) * math.sqrt(128 * 9) - 1.7) < 0.05
    x = synth.synth_images(2, 32, 32, 0)
    assert torch.equal(x[1], synth.synth_images(3, 32, 32, 0)[1])
    assert abs(float(x.mean())) < 0.05 and abs(float(x.std()) - 1) < 0.05


# ----------------------------------------------------------------------------- state_dict schema (drop-in boundary)
@pytest.mark.parametrize("kind,nkeys", [("net", 639), ("net_deploy", 471), ("net_amff", 535), ("net_small_target", 582), ("net_old", None)])
def test_state_dict_schema_matches_survey(kind, nkeys):
    net = importlib.import_module("ccnet_amd.network." + kind).Net(0.33, 0.5, 6)
    sd = net.state_dict()
    if nkeys is not None:  # SURVEY gives no count for net_old; its fixture's weight checksum pins the schema instead
        assert len(sd) == nkeys
    assert tuple(sd["cls_preds.0.weight"].shape) == (6, 128, 1, 1) and tuple(sd["obj_preds.2.bias"].shape) == (1,)
    if kind == "net":
        assert tuple(sd["stem.conv.conv.weight"].shape) == (32, 12, 3, 3)
        assert tuple(sd["dark5.2.transformer.encoder.layers.0.self_attn.in_proj_weight"].shape) == (768, 256)
        assert "dark5.2.transformer.decoder.layers.0.multihead_attn.out_proj.bias" in sd
    elif kind == "net_old":
        assert tuple(sd["stem.0.conv.weight"].shape) == (32, 3, 3, 3) and "stem.0.bn.running_var" in sd
        assert tuple(sd["dark5.2.transformer.encoder.layers.0.self_attn.in_proj_weight"].shape) == (768, 256)
        assert tuple(sd["dark3.0.conv.weight"].shape) == (128, 64, 3, 3) and "dark3.1.m.2.conv2.bn.weight" in sd
    else:
        assert tuple(sd["stem.conv.weight"].shape) == (32, 3, 6, 6) and "stem.bn.running_var" in sd
    # the fixtures' calibrated BN statistics cover exactly this model's BN layers
    with np.load(os.path.join(GOLD, "bn_calib_%s.npz" % kind)) as z:
        assert set(z.files) == {k for k in sd if k.endswith("running_mean") or k.endswith("running_var")}


def test_load_model_weights_contract(tmp_path):
    from ccnet_amd.network.net_deploy import Net
    from ccnet_amd.utils.util import load_model_weights
    src = Net(0.33, 0.5, 6)
    sd = {k: torch.full_like(v, 0.25) if v.is_floating_point() else v for k, v in src.state_dict().items()}
    sd["cls_preds.0.weight"] = torch.zeros(80, 128, 1, 1)  # wrong shape: skipped, not an error
    path = str(tmp_path / "model_best.pth")
    torch.save({"epoch": 3, "model": sd, "ap50": 0.5, "ap5095": 0.321}, path)
    dst = Net(0.33, 0.5, 6)
    before = dst.state_dict()["cls_preds.0.weight"].clone()
    dst, ap = load_model_weights(dst, path, torch.device("cpu"))
    assert ap == 0.321
    assert float(dst.state_dict()["stem.conv.weight"].mean()) == 0.25
    assert torch.equal(dst.state_dict()["cls_preds.0.weight"], before)
    sd["not.a.key"] = torch.zeros(1)
    torch.save({"epoch": 3, "model": sd, "ap50": 0.5, "ap5095": 0.3}, path)
    with pytest.raises(KeyError):
        load_model_weights(Net(0.33, 0.5, 6), path, torch.device("cpu"))


def test_net_constructor_errors_and_cpu_refusal():
    from ccnet_amd.network.net_deploy import Net
    from ccnet_amd.utils import bbox
    from ccnet_amd.utils.initialization import model_initializer
    with pytest.raises(AttributeError):
        Net(0.33, 0.5, 6, act="gelu")
    net = Net(*model_initializer("s"), 6)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 64, 64))      # CPU tensor: no silent fallback
    with pytest.raises(RuntimeError):
        net.train()
    with pytest.raises(RuntimeError):
        bbox.decode_outputs([torch.zeros(1, 11, 8, 8)], [64, 64])
    with pytest.raises(RuntimeError):
        bbox.post_process(torch.zeros(1, 10, 11), 6, [64, 64], [[0, 64, 64]])
    assert model_initializer("l") == (1.0, 1.0) and model_initializer("tiny") == (0.33, 0.375)


# ----------------------------------------------------------------------------- host-side weight math
def test_fold_bn_and_s2d_identity():
    from ccnet_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 8, 12, 10, generator=g)
    w = torch.randn(16, 32, 3, 3, generator=g) * 0.1
    bn = (torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g), torch.randn(16, generator=g), torch.rand(16, generator=g) + 0.5)
    s2d = torch.cat((x[..., ::2, ::2], x[..., ::2, 1::2], x[..., 1::2, ::2], x[..., 1::2, 1::2]), 1)
    ref = F.batch_norm(F.conv2d(s2d, w, None, 1, 1), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    wf, bf = ops.fold_bn(w, bn)
    got = F.conv2d(x.double(), ops.s2d_to_6x6(wf), bf, 2, 2)
    assert float((got.float() - ref).abs().max()) < 1e-5


def test_pack_layouts():
    from ccnet_amd import ops, _lib
    w6 = torch.arange(2 * 3 * 6 * 6, dtype=torch.float64).reshape(2, 3, 6, 6)
    wp, bp = ops.pack_stem(w6, torch.zeros(2), "cpu")
    assert wp.shape == (2, 160)
    assert float(wp[1, (2 * 6 + 4) * 8 + 5]) == float(w6[1, 2, 4, 5]) and float(wp[0, 6]) == 0 and float(wp[0, 159]) == 0
    cp, kp = _lib.conv_pack_dims(_lib.BF16, 128, 1152)
    assert (cp, kp) == (128, 1152)
    assert _lib.conv_pack_dims(_lib.F32, 5, 128) == (16, 128)
    assert _lib.conv_pack_dims(_lib.BF16, 200, 100) == (256, 128)
    w = torch.randn(5, 8, 3, 3)
    pc = ops.PackedConv(w, torch.zeros(5), _lib.F32, "cpu")
    assert pc.weight.shape == (16, 96) and torch.equal(pc.weight[:5, :72].reshape(5, 3, 3, 8).permute(0, 3, 1, 2), w)
    assert float(pc.weight[5:].abs().max()) == 0 and float(pc.weight[:, 72:].abs().max()) == 0


def test_fragment_stream_layouts_follow_the_header():
    """include/ccnet_hip.h describes the weight streams of ccnet_conv_halo, ccnet_conv_pw and ccnet_bneck_fused fragment by fragment
    (one MFMA A operand = [64 lanes][8]: lane l holds row l & 15, k (l >> 4) * 8 .. + 7).  Packed on the CPU with weights that encode
    their own (output channel, input channel, tap), every fragment must sit where the header says."""
    from ccnet_amd import ops

    def coded(cout, cin, kh):
        o = torch.arange(cout, dtype=torch.float64).view(cout, 1, 1, 1)
        c = torch.arange(cin, dtype=torch.float64).view(1, cin, 1, 1)
        t = torch.arange(kh * kh, dtype=torch.float64).view(1, 1, kh, kh)
        return (o + 0 * c + 0 * t).expand(cout, cin, kh, kh), (c + 0 * o + 0 * t).expand(cout, cin, kh, kh), (t + 0 * o + 0 * c).expand(cout, cin, kh, kh)

    def frag(stream, n64, f):
        return stream[n64, f].reshape(4, 16, 8).float()      # [kq][row][e]

    # ccnet_conv_halo: per 64 output channels: chunk of 32 input channels, tap, 16-row n-tile; then the first 9 fragments again
    cout, cin = 128, 64
    wo, wc, wt = coded(cout, cin, 3)
    so, sc, st = (ops.PackedHalo(w, None, "cpu").weight for w in (wo, wc, wt))
    assert so.shape == (2, (cin // 32) * 36 + 9, 512)
    for n64, cc, tap, j in ((0, 0, 0, 0), (1, 1, 5, 3), (0, 1, 8, 2)):
        f = cc * 36 + tap * 4 + j
        assert torch.equal(frag(so, n64, f)[:, :, 0], (n64 * 64 + j * 16 + torch.arange(16.0)).expand(4, 16))
        assert torch.equal(frag(sc, n64, f)[:, 0, :], cc * 32 + torch.arange(32.0).view(4, 8))
        assert float(frag(st, n64, f).min()) == float(frag(st, n64, f).max()) == tap
    assert torch.equal(so[:, -9:], so[:, :9]) and torch.equal(sc[:, -9:], sc[:, :9])
    # ccnet_conv_pw: per 64 output channels: k-step of 32, n-tile; then the first 8 again
    wo, wc, _ = coded(128, 128, 1)
    so, sc = (ops.PackedPw(w, None, "cpu").weight for w in (wo, wc))
    assert so.shape == (2, 4 * 4 + 8, 512)
    f = 2 * 4 + 1
    assert torch.equal(frag(so, 1, f)[:, :, 0], (64 + 16 + torch.arange(16.0)).expand(4, 16))
    assert torch.equal(frag(sc, 1, f)[:, 0, :], 64 + torch.arange(32.0).view(4, 8))
    assert torch.equal(so[:, -8:], so[:, :8])
    # ccnet_bneck_fused: per 64 output channels of conv2: ALL of conv1 (k-step major, n-tile), conv2 in conv_halo's order, the head again
    for c, ring in ((64, 8), (128, 16), (256, 16)):
        o1, c1, _ = coded(c, c, 1)
        o2, c2, t2 = coded(c, c, 3)
        zero1, zero2 = torch.zeros(c, c, 1, 1, dtype=torch.float64), torch.zeros(c, c, 3, 3, dtype=torch.float64)
        b = torch.zeros(c)
        f1 = (c // 32) * (c // 16)
        s_o1 = ops.PackedBneck(o1, b, zero2, b, "cpu").weight
        s_c1 = ops.PackedBneck(c1, b, zero2, b, "cpu").weight
        s_o2 = ops.PackedBneck(zero1, b, o2, b, "cpu").weight
        s_t2 = ops.PackedBneck(zero1, b, t2, b, "cpu").weight
        assert s_o1.shape == (c // 64, f1 + (c // 32) * 36 + ring, 512)
        ks, nt = c // 32 - 1, c // 16 - 1
        for n64 in range(c // 64):                                  # conv1 is the same in every block's stream
            assert torch.equal(frag(s_o1, n64, ks * (c // 16) + nt)[:, :, 0], (nt * 16 + torch.arange(16.0)).expand(4, 16))
            assert torch.equal(frag(s_c1, n64, ks * (c // 16) + nt)[:, 0, :], ks * 32 + torch.arange(32.0).view(4, 8))
        n64, cc, tap, j = c // 64 - 1, c // 32 - 1, 7, 2
        f = f1 + cc * 36 + tap * 4 + j
        assert torch.equal(frag(s_o2, n64, f)[:, :, 0], (n64 * 64 + j * 16 + torch.arange(16.0)).expand(4, 16))
        assert float(frag(s_t2, n64, f).min()) == float(frag(s_t2, n64, f).max()) == tap
        assert torch.equal(s_o1[:, -ring:], s_o1[:, :ring])


# ----------------------------------------------------------------------------- C ABI surface
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ccnet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(ccnet_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    lib = ctypes.CDLL(os.path.join(ROOT, "ccnet_amd", "libccnet_hip.so"))
    for n in names:
        assert hasattr(lib, n), "libccnet_hip.so does not export %s" % n


def test_c_abi_validates_before_launching():
    """Bad descriptors are rejected on the host with a message; nothing reaches a GPU (there is none here)."""
    from ccnet_amd import _lib
    L = _lib.lib()
    assert L.ccnet_version() >= 100
    d = _lib.ConvDesc()
    d.dtype, d.out_dtype, d.nseg = 1, 1, 3
    assert L.ccnet_conv2d(ctypes.byref(d), None) == -1 and b"nseg" in L.ccnet_last_error()
    assert L.ccnet_layernorm(None, None, None, None, None, 1, 4, 256, 1e-5, None) == -1
    assert L.ccnet_nms_workspace_bytes(2, 100) > 2 * 128 * 2 * 8
    x = _lib.Slice(0x1000, 64, 0, 64, 0)
    assert L.ccnet_bneck_fused(x, 1, 8, 8, 0x2000, 0x3000, 0x4000, 1, _lib.ACT["silu"], x, None) == -1 and b"overlaps" in L.ccnet_last_error()
    y = _lib.Slice(0x100000, 64, 0, 64, 0)
    assert L.ccnet_bneck_fused(x, 1, 8, 8, 0x2000, 0x3000, 0x4000, 1, _lib.ACT["relu"], y, None) == -1 and b"SiLU" in L.ccnet_last_error()
    x256 = _lib.Slice(0x1000, 256, 0, 256, 0)
    assert L.ccnet_bneck_fused(x256, 1, 8, 30, 0x2000, 0x3000, 0x4000, 1, _lib.ACT["silu"], _lib.Slice(0x100000, 256, 0, 256, 0), None) == -1
    assert L.ccnet_bneck_fused_stream_elems(96) == 0 and L.ccnet_bneck_fused_stream_elems(64) == (8 + 72 + 8) * 512
    h = ctypes.c_void_p()
    assert L.ccnet_plan_begin(ctypes.byref(h)) == 0
    assert L.ccnet_plan_begin(ctypes.byref(ctypes.c_void_p())) == -1          # one open plan per thread
    assert L.ccnet_plan_end() == 0 and L.ccnet_plan_num_ops(h) == 0
    assert L.ccnet_plan_end() == -1
    L.ccnet_plan_destroy(h)


# ----------------------------------------------------------------------------- multi-GPU layer on gloo
def test_shard_range_partitions_batch():
    from ccnet_amd.parallel import shard_range
    for total, world in [(256, 8), (10, 4), (3, 8), (64, 1)]:
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from ccnet_amd.parallel import all_gather_detections
    rng = np.random.RandomState(rank)
    local = [rng.rand(n, 7).astype(np.float32) for n in ([3, 0] if rank == 0 else [5])]
    if rank == 1:
        local.append(None)
    out = all_gather_detections(local)
    hinted = all_gather_detections(local, max_local=2)
    assert len(hinted) == len(out) and all((a is None and b is None) or np.array_equal(a, b) for a, b in zip(out, hinted))
    # one image with more rows than the first collective carries: every rank must take the second one
    big = [np.full((300 + rank, 7), float(rank), np.float32)] if rank == 1 else [np.zeros((2, 7), np.float32)]
    gb = all_gather_detections(big, max_local=1)
    assert [g.shape[0] for g in gb] == [2, 301] and float(gb[1][300, 3]) == 1.0
    q.put((rank, [None if o is None else o.copy() for o in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_detections_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    exp0 = [np.random.RandomState(0).rand(3, 7).astype(np.float32)]
    r0 = np.random.RandomState(0)
    e = [r0.rand(3, 7).astype(np.float32), r0.rand(0, 7).astype(np.float32)]
    r1 = np.random.RandomState(1)
    e += [r1.rand(5, 7).astype(np.float32), None]
    for rank in (0, 1):
        got = res[rank]
        assert len(got) == 4
        for a, b in zip(got, e):
            assert (a is None and b is None) or (a.shape == b.shape and np.array_equal(a, b))


# ------------------------------------------------------------------------------------------ mAP restatement
def test_map_restatement_matches_reference_get_map():
    """ccnet_amd/utils/map.py against golden values produced by the reference's own get_map / voc_ap
    (oracle/make_golden_map.py): three synthetic datasets x ten IoU thresholds, and six raw precision/recall curves."""
    import json
    from ccnet_amd.utils import map as cmap
    with open(os.path.join(GOLD, "map_case.json")) as f:
        gold = json.load(f)
    for case in gold["get_map"]:
        gt = [[tuple(g) for g in img] for img in case["gt"]]
        dets = [[(d[0], float(str(d[1])[:6]), d[2], d[3], d[4], d[5]) for d in img] for img in case["dets"]]
        for thr, ref in zip(case["thresholds"], case["map"]):
            got, _ = cmap.average_precision(gt, dets, case["num_classes"], thr)
            assert abs(got - ref) < 1e-12, (case["seed"], thr, got, ref)
        m, vals = cmap.map_50_95(gt, dets, case["num_classes"])
        assert abs(m - float(np.mean(case["map"]))) < 1e-12
    for c in gold["voc_ap"]:
        assert abs(cmap.voc_ap(c["rec"], c["prec"]) - c["ap"]) < 1e-12


def test_map_records_follow_eval_txt_format():
    """tools/eval.py:89-103: score = obj*cls, str()[:6]; pixels truncated with int(); columns top,left,bottom,right."""
    from ccnet_amd.utils import map as cmap
    out = [np.array([[10.9, 20.2, 30.7, 40.99, 0.9, 0.87654321, 2.0]], dtype=np.float32), np.zeros((0, 7), np.float32), None]
    recs = cmap.detections_to_records(out)
    assert recs[1] == [] and recs[2] == []
    c, conf, l, t, r, b = recs[0][0]
    assert (c, l, t, r, b) == (2, 20, 10, 40, 30)
    assert conf == float(str(np.float32(0.9) * np.float32(0.87654321))[:6])


def test_socket_inference_surface(tmp_path):
    """tools/socket2springboot.py:33-123: unknown model names give {}, classes.txt is one name per line."""
    from ccnet_amd.tools import socket2springboot as s2s
    assert s2s.inference("Other", "x.pth", "NEUDET", [224, 224], 0.5, 0.6, ["a.jpg"]) == {}
    p = tmp_path / "classes.txt"
    p.write_text("crazing\ninclusion\n\npatches\n")
    assert s2s.get_classes(str(p)) == (["crazing", "inclusion", "patches"], 3)
    import inspect
    assert list(inspect.signature(s2s.inference).parameters)[:7] == ["model_name", "model_path", "dataset", "input_shape", "conf_thres",
                                                                      "nms_thres", "image_paths"]


def test_eval_dirs_match_reference_get_map(tmp_path):
    """tools/eval.py file round trip: the golden get_map cases written as ground_truth/ and detection_results/ txt files
    and scored through ccnet_amd.tools.eval.map_from_dirs reproduce the reference's numbers."""
    import json
    from ccnet_amd.tools import eval as ceval
    with open(os.path.join(GOLD, "map_case.json")) as f:
        case = json.load(f)["get_map"][0]
    names = ["cls0", "cls1", "cls2"]
    (tmp_path / "gt").mkdir(); (tmp_path / "dr").mkdir()
    for i, (g, d) in enumerate(zip(case["gt"], case["dets"])):
        with open(tmp_path / "gt" / ("img%03d.txt" % i), "w") as f:
            for c, l, t, r, b in g:
                f.write("cls%d %d %d %d %d\n" % (c, l, t, r, b))
        with open(tmp_path / "dr" / ("img%03d.txt" % i), "w") as f:
            for c, conf, l, t, r, b in d:
                f.write("cls%d %s %d %d %d %d\n" % (c, str(conf)[:6], l, t, r, b))
    got = ceval.map_from_dirs(str(tmp_path / "gt"), str(tmp_path / "dr"), names, case["thresholds"])
    for thr, ref in zip(case["thresholds"], case["map"]):
        assert abs(got[thr] - ref) < 1e-12


# ----------------------------------------------------------------------------- image preparation
def _prep_cases():
    z = np.load(os.path.join(GOLD, "preprocess_cases.npz"))
    names = sorted({k.rsplit("/", 1)[0] for k in z.files})
    return [(n, z[n + "/src"], tuple(int(v) for v in z[n + "/input_shape"]), z[n + "/out"]) for n in names]


def test_preprocess_oracle_matches_reference_fixture():
    """Fixtures come from the reference's own evaluate_one_img / Datasets.__getitem__ (oracle/make_golden_preprocess.py)."""
    from oracle import preprocess_oracle as po
    cases = _prep_cases()
    assert len(cases) == 8
    for name, src, shape, exp in cases:
        if src.ndim == 2:
            src = np.repeat(src[..., None], 3, 2)  # convert("RGB") of a grey image replicates the channel
        got = (po.preprocess_eval if name.startswith("eval/") else po.preprocess_letterbox)(src, shape)
        assert got.dtype == np.float32 and np.array_equal(got, exp), name


def test_preprocess_oracle_matches_pillow_live():
    """The resize is Pillow's; where Pillow is importable the restatement is checked against it on random sizes (bit-exact)."""
    Image = pytest.importorskip("PIL.Image")
    from oracle import preprocess_oracle as po
    rng = np.random.default_rng(1)
    for t in range(25):
        ih, iw = (int(v) for v in rng.integers(1, 200, 2))
        oh, ow = (int(v) for v in rng.integers(1, 260, 2))
        if t % 5 == 0:
            ow = iw
        img = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
        ref = np.array(Image.fromarray(img, "RGB").resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(po.resize_bicubic_u8(img, ow, oh), ref), (ih, iw, oh, ow)


def test_resample_tables_of_the_library_match_oracle():
    """ccnet_resample_coeffs is host code (no GPU): its tables must equal the oracle's for up-, down- and same-size axes."""
    from ccnet_amd.utils import preprocess as pp
    from oracle import preprocess_oracle as po
    for in_size, out_size in [(200, 224), (200, 640), (1080, 640), (1920, 640), (97, 64), (50, 64), (7, 7), (1, 5), (5, 1), (3000, 17), (640, 1280)]:
        b, k = pp.resample_tables(in_size, out_size)
        ob, ok, oks = po.precompute_coeffs(in_size, out_size)
        assert k.shape[1] == oks and np.array_equal(b, ob) and np.array_equal(k, ok), (in_size, out_size)
        assert (b[:, 0] >= 0).all() and (b[:, 0] + b[:, 1] <= in_size).all() and (b[:, 1] <= oks).all()
    with pytest.raises(RuntimeError):
        pp.resample_tables(0, 5)
    with pytest.raises(RuntimeError):
        pp.resample_tables(10 ** 6, 1)  # more taps than the library supports


def test_normalisation_lut_and_letterbox_geometry():
    from ccnet_amd.utils import preprocess as pp
    from oracle import preprocess_oracle as po
    lut = pp.normalisation_lut()
    assert lut.shape == (3, 256) and lut.dtype == np.float32
    img = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, 2)
    assert np.array_equal(po.normalise(img)[:, 0, :], lut)
    for iw, ih, shape in [(120, 80, (96, 96)), (60, 150, (64, 96)), (48, 48, (96, 96)), (200, 200, (640, 640)), (1920, 1080, (640, 640)), (333, 77, (224, 224))]:
        assert pp.letterbox_geometry(iw, ih, shape) == po.letterbox_geometry(iw, ih, shape)


def test_socket_protocol_over_a_real_connection():
    """tools/socket2springboot.py:174-215: text up to "over" is JSON; the reply is str(dict); any exception -> "500".
    A model_name other than "Net" answers {} without touching a GPU, which is what this CPU test can exercise end to end."""
    import json
    import socket
    from ccnet_amd.tools import socket2springboot as s2s
    req = {"model_name": "Other", "model_path": "w.pth", "dataset": "NEUDET", "input_shape": [224, 224], "conf_thres": 0.5, "nms_thres": 0.6,
           "image_path": ["a.jpg", "b.jpg"]}
    assert s2s.handle_request(json.dumps(req) + "over") == "{}"
    assert s2s.handle_request("not json over") == "500"
    assert s2s.handle_request(json.dumps({"model_name": "Net"}) + "over") == "500"      # missing keys
    assert s2s.handle_request(json.dumps(req) + "over\n") == "500"                       # :190 cuts four characters of the raw text
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.bind(("127.0.0.1", 0))
    srv.listen(1)
    cli = socket.create_connection(srv.getsockname())
    conn, addr = srv.accept()
    t = s2s.ServerThreading(conn, addr)
    t.start()
    payload = (json.dumps(req) + "over").encode()
    cli.sendall(payload[:20])
    cli.sendall(payload[20:])
    assert cli.recv(1 << 16).decode() == "{}"
    t.join(10)
    cli.close()
    srv.close()
    res = {"image": [{"image_path": "a", "image_id": 0}], "annotations": [{"image_id": 0, "box": [1, 2, 3, 4], "predicted_class": "c", "conf": float(np.float32(0.93))}]}
    assert s2s._reply_text(res) == "{'image': [{'image_path': 'a', 'image_id': 0}], 'annotations': [{'image_id': 0, 'box': [1, 2, 3, 4], 'predicted_class': 'c', 'conf': 0.93}]}"


def test_product_path_fails_loudly_without_the_hip_library(tmp_path):
    """No CPU fallback: with the shared library absent every entry point raises instead of computing something else.  Run in a
    child process because the loaded library is cached per process."""
    import subprocess
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from ccnet_amd import _lib\n"
        "from ccnet_amd.utils import preprocess as pp\n"
        "for fn in (lambda: _lib.lib(), lambda: pp.resample_tables(10, 20), lambda: _lib.conv_pack_dims(1, 64, 576)):\n"
        "    try:\n"
        "        fn()\n"
        "    except RuntimeError as e:\n"
        "        assert 'is missing' in str(e) and 'no CPU fallback' in str(e), str(e)\n"
        "    else:\n"
        "        raise SystemExit('no error raised')\n"
        "print('loud')\n" % ROOT)
    env = dict(os.environ, CCNET_HIP_LIB=str(tmp_path / "nowhere" / "libccnet_hip.so"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "loud" in r.stdout, r.stdout + r.stderr


def test_compat_shim_resolves_reference_import_lines():
    """compat/ first on PYTHONPATH: the import lines of the reference's callers (tools/demo.py:18-21,
    tools/socket2springboot.py:22-26, tools/eval.py:24-29) resolve to ccnet_amd without editing the caller."""
    import subprocess
    code = (
        "from network.net import Net\n"
        "from network.net_deploy import Net as D\n"
        "from network.net_amff import Net as A\n"
        "from network.net_small_target import Net as S\n"
        "from network.net_old import Net as O\n"
        "from utils.bbox import decode_outputs, post_process\n"
        "from utils.initialization import device_initializer, model_initializer\n"
        "from utils.util import get_classes, draw_rectangle, load_model_weights, get_root_path, get_val_lines\n"
        "import tools.demo, tools.eval, tools.socket2springboot, utils.map\n"
        "assert Net.__module__ == 'ccnet_amd.network.net' and D.__module__ == 'ccnet_amd.network.net_deploy'\n"
        "assert A.__module__ == 'ccnet_amd.network.net_amff' and S.__module__ == 'ccnet_amd.network.net_small_target'\n"
        "assert O.__module__ == 'ccnet_amd.network.net_old'\n"
        "assert decode_outputs.__module__ == 'ccnet_amd.utils.bbox' and post_process.__module__ == 'ccnet_amd.utils.bbox'\n"
        "assert load_model_weights.__module__ == 'ccnet_amd.utils.util' and model_initializer('s') == (0.33, 0.5)\n"
        "assert tools.demo.inference.__module__ == 'ccnet_amd.tools.demo'\n"
        "m = Net(0.33, 0.5, 6, 'silu'); assert len(m.state_dict()) == 639 and m.precision == 'fp32'\n"
        "print('shim ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "compat") + os.pathsep + ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0 and "shim ok" in r.stdout, r.stdout + r.stderr


def test_demo_inference_returns_the_drawn_image_through_the_shim():
    """tools/demo.py:28-81 through compat/: the reference's six positional arguments, and the return value is the IMAGE
    (the same array, drawn on in place by draw_rectangle) -- `image = inference(...); cv2.imshow("image", image)` keeps
    working.  OpenCV is absent: a stand-in cv2 records the drawing calls; the three hot-path statements need a GPU, so
    predict() is replaced by canned post_process output here (tests/test_forward_gpu.py runs the real one)."""
    import subprocess
    code = (
        "import sys, types, numpy as np\n"
        "calls = []\n"
        "cv2 = types.ModuleType('cv2'); cv2.INTER_CUBIC = 2; cv2.FONT_HERSHEY_SIMPLEX = 0\n"
        "cv2.resize = lambda img, size, interpolation=None: np.zeros((size[1], size[0], 3), np.uint8)\n"
        "cv2.getTextSize = lambda text, font, scale, thick: ((len(text) * 6, 8), 3)\n"
        "def rectangle(img, p0, p1, colour, thickness=1):\n"
        "    calls.append(('rect', p0, p1)); img[p0[1]:p1[1] + 1, p0[0]:p1[0] + 1] = 255\n"
        "cv2.rectangle = rectangle\n"
        "cv2.putText = lambda img, text, org, font, scale, colour, thickness=1: calls.append(('text', text, org))\n"
        "sys.modules['cv2'] = cv2\n"
        "import tools.demo as demo\n"
        "from tools.demo import inference\n"
        "class M:\n"
        "    _cfg = {'nc': 6}\n"
        "rows = np.array([[10.7, 20.2, 50.9, 80.1, 0.9, 0.8, 3.0], [-4.0, 5.0, 300.0, 999.0, 0.5, 0.5, 0.0]], np.float32)\n"
        "seen = {}\n"
        "def fake_predict(model, images, num_classes, input_shape, image_info, conf_thres, nms_thres):\n"
        "    seen.update(nc=num_classes, info=image_info, shape=tuple(images.shape), conf=conf_thres, nms=nms_thres)\n"
        "    return [rows]\n"
        "demo.predict = fake_predict\n"
        "image = np.zeros((120, 160, 3), np.uint8)\n"
        "out = inference(M(), image, [224, 224], 0.5, 0.6, 'cpu')\n"
        "assert out is image and out.any(), 'the reference returns the image it drew on'\n"
        "assert seen == dict(nc=6, info=[[0, 160, 120]], shape=(1, 3, 224, 224), conf=0.5, nms=0.6), seen\n"
        "assert ('rect', (20, 10), (80, 50)) in calls and ('rect', (5, 0), (160, 120)) in calls, calls   # floor, clipped to the image\n"
        "assert [c[1] for c in calls if c[0] == 'text'] == ['3:72.0%', '0:25.0%'], calls\n"
        "demo.class_names = ['a', 'b', 'c', 'd', 'e', 'f']; demo.num_classes = 6; del calls[:]   # the reference's module globals\n"
        "inference(M(), image, [224, 224], 0.5, 0.6, 'cpu')\n"
        "assert [c[1] for c in calls if c[0] == 'text'] == ['d:72.0%', 'a:25.0%'], calls\n"
        "demo.predict = lambda *a: [None]\n"
        "blank = np.zeros((8, 8, 3), np.uint8)\n"
        "assert inference(M(), blank, [224, 224], 0.5, 0.6, 'cpu') is blank and not blank.any()\n"
        "demo.predict = fake_predict\n"
        "res, info = demo.detect(M(), image, [224, 224], 0.5, 0.6, 'cpu')\n"
        "assert res[0] is rows and info == [[0, 160, 120]]\n"
        "print('demo ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "compat") + os.pathsep + ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0 and "demo ok" in r.stdout, r.stdout + r.stderr


def test_socket_server_validates_client_sizes_and_bounds_its_caches(tmp_path):
    """input_shape comes from the client and every distinct size compiles a plan: sizes are validated before anything is
    allocated; the model cache is an LRU keyed by the checkpoint's path, mtime and size."""
    import json
    from ccnet_amd.tools import socket2springboot as s2s
    names = ["c%d" % i for i in range(6)]
    for bad in ([100, 100], [224], [224, 4096 * 32], [0, 224], ["224", 224], [224.0, 224], [-32, 32]):
        with pytest.raises(ValueError):
            s2s.check_input_shape(bad)
        req = {"model_name": "Net", "model_path": "", "dataset": "NEUDET", "input_shape": bad, "conf_thres": 0.5, "nms_thres": 0.6, "image_path": []}
        assert s2s.handle_request(json.dumps(req) + "over", class_names=names) == "500"
    assert s2s.check_input_shape([224, 640]) == [224, 640]
    # a checkpoint rewritten in place is reloaded (the reference reloads on every request, :58-61)
    from ccnet_amd import synth
    from ccnet_amd.network.net import Net
    net = Net(0.33, 0.5, 6, "silu")
    path = str(tmp_path / "w.pth")
    sd = synth.synth_state_dict(net.state_dict(), seed=0)
    torch.save({"epoch": 1, "model": sd, "ap50": 0.0, "ap5095": 0.0}, path)
    s2s._models.clear()
    m1 = s2s._model_for(path, 6, "cpu")
    assert s2s._model_for(path, 6, "cpu") is m1
    sd2 = synth.synth_state_dict(net.state_dict(), seed=1)
    torch.save({"epoch": 2, "model": sd2, "ap50": 0.0, "ap5095": 0.0}, path)
    os.utime(path, ns=(os.stat(path).st_atime_ns, os.stat(path).st_mtime_ns + 1_000_000))
    m2 = s2s._model_for(path, 6, "cpu")
    assert m2 is not m1 and torch.equal(m2.state_dict()["stems.0.conv.weight"], sd2["stems.0.conv.weight"])
    assert len(s2s._models) == 1
    for nc in range(1, s2s.MAX_MODELS + 3):
        s2s._model_for("", nc, "cpu")
    assert len(s2s._models) == s2s.MAX_MODELS
    s2s._models.clear()


def test_isa_lint_refuses_the_op_sel_form_and_passes_the_built_objects(tmp_path):
    """scripts/check_isa.py: the packed-fp32 operand form that measured wrong on MI355X (low lane reads the high dword of a source
    pair, profiles/r03_hazard_*.txt) is reported; the forms that measured clean are not; the objects of the built library are clean."""
    import glob
    import subprocess
    lint = os.path.join(ROOT, "scripts", "check_isa.py")
    bad = tmp_path / "bad.s"
    bad.write_text("_Zkernel:\n\tv_pk_add_f32 v[0:1], v[70:71], v[74:75] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                   "\tv_pk_fma_f32 v[4:5], v[4:5], v[28:29], v[24:25] op_sel:[1,0,0]\n")
    good = tmp_path / "good.s"
    good.write_text("_Zkernel:\n\tv_pk_mul_f32 v[0:1], v[0:1], v[76:77] op_sel_hi:[1,0]\n\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] neg_lo:[0,1] neg_hi:[0,1]\n"
                    "\tv_pk_fma_f32 v[24:25], v[4:5], v[36:37], v[24:25] op_sel_hi:[0,1,1]\n")
    r = subprocess.run([sys.executable, lint, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout.count("low lane reads the high dword") == 2, r.stdout + r.stderr
    r = subprocess.run([sys.executable, lint, str(good)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == "", r.stdout + r.stderr
    objs = sorted(glob.glob(os.path.join(ROOT, "ccnet_amd", "csrc", "build", "*.o")))
    if not objs:
        pytest.skip("no object files (the library was built elsewhere)")
    r = subprocess.run([sys.executable, lint] + objs, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
