"""Per-kernel parity on the GPU, through the C ABI: every HIP kernel against a plain torch fp32 CPU reference of
the same op (floating point: tolerance stated per test), NMS/filter against the oracle (bit-exact)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from ccnet_amd import ops
    return ops


def tdt(code):
    return torch.bfloat16 if code == 1 else torch.float32


def to_nhwc(x, code, ldc=None, coff=0):
    """NCHW cpu fp32 -> NHWC device tensor in a (possibly wider) buffer."""
    B, Cc, H, W = x.shape
    ldc = ldc or Cc
    buf = torch.zeros(B, H, W, ldc, dtype=tdt(code), device=DEV)
    buf[..., coff:coff + Cc] = x.permute(0, 2, 3, 1).to(DEV).to(tdt(code))
    return buf


def from_nhwc(buf, coff, Cc):
    return buf[..., coff:coff + Cc].float().cpu().permute(0, 3, 1, 2).contiguous()


def q(x, code):
    """Round through the storage dtype (so the reference sees the same operand values)."""
    return x.to(tdt(code)).float()


def tol(code):
    # fp32: accumulation-order noise only.  bf16: one rounding of the output (2^-9) plus accumulated operand products.
    return 2e-5 if code == 0 else 1.2e-2


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


CONV_CASES = [
    # cin, cout, k, stride, pad, H, W, act, residual
    (64, 128, 1, 1, 0, 20, 20, "silu", False),
    (32, 32, 3, 1, 1, 17, 23, "silu", True),
    (128, 128, 3, 1, 1, 20, 20, "silu", True),
    (64, 64, 3, 2, 1, 24, 24, "relu", False),
    (32, 64, 6, 2, 2, 32, 32, "silu", False),
    (128, 16, 1, 1, 0, 13, 11, "lrelu", False),
    (256, 512, 1, 1, 0, 7, 7, "relu6", False),
    (8, 24, 3, 1, 1, 9, 9, "none", False),
    (128, 5, 1, 1, 0, 20, 20, "none", False),
    # shapes the LDS halo-patch kernel takes (3x3 s1 p1 / 6x6 s2 p2, output width % 16 == 0, Cout >= 64)
    (128, 128, 3, 1, 1, 24, 32, "silu", True),
    (64, 192, 3, 1, 1, 20, 16, "silu", False),
    (128, 64, 3, 1, 1, 8, 48, "none", False),
    (256, 80, 3, 1, 1, 5, 16, "relu", True),
    (64, 128, 6, 2, 2, 32, 64, "silu", False),
    (128, 64, 6, 2, 2, 20, 32, "none", True),
    # widths % 16 == 8: the 16 x 8 tile geometry of the patch kernel
    (128, 128, 3, 1, 1, 20, 40, "silu", True),
    (64, 96, 3, 1, 1, 33, 24, "silu", False),
    (64, 64, 6, 2, 2, 48, 80, "silu", False),
]


@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case, code):
    ops = _ops()
    cin, cout, k, stride, pad, H, W, act, use_res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    B = 2
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    pc = ops.PackedConv(w, b, code, DEV)
    xin = to_nhwc(x, code, ldc=cin + 16, coff=8)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out_dt = 0 if cout == 5 else code
    ldo = cout + 8 if cout % 8 == 0 else 11
    out = torch.zeros(B, Ho, Wo, ldo, dtype=tdt(out_dt), device=DEV)
    res = None
    ref_res = 0
    if use_res:
        r = torch.randn(B, cout, Ho, Wo, generator=g)
        resbuf = to_nhwc(r, code)
        res = ops.sl(resbuf)
        ref_res = q(r, code)
    ops.conv2d([ops.sl(xin, 8, cin)], pc, B, H, W, stride, pad, ops.sl(out, 0 if cout % 8 else 8, cout), act, res, out_dt)
    torch.cuda.synchronize()
    actf = {"silu": F.silu, "relu": F.relu, "relu6": F.relu6, "lrelu": lambda t: F.leaky_relu(t, 0.1), "none": lambda t: t}[act]
    ref = actf(F.conv2d(q(x, code), q(w, code), b, stride, pad)) + ref_res
    got = from_nhwc(out, 0 if cout % 8 else 8, cout)
    assert rel(got, ref) < tol(code), rel(got, ref)
    # channels outside the slice stay untouched
    if cout % 8 == 0:
        assert float(out[..., :8].float().abs().max()) == 0.0


@pytest.mark.parametrize("code", [0, 1])
def test_conv2d_two_segments_with_upsample(code):
    """torch.cat([nn.Upsample(2)(a), b], 1) -> 1x1 and 3x3 conv without materialising either (network/net.py:208-213)."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 12, 16
    a = torch.randn(B, 32, H // 2, W // 2, generator=g)
    bb = torch.randn(B, 64, H, W, generator=g)
    for k in (1, 3):
        w = torch.randn(48, 96, k, k, generator=g) / math.sqrt(96 * k * k)
        bias = torch.randn(48, generator=g) * 0.1
        pc = ops.PackedConv(w, bias, code, DEV)
        ta, tb = to_nhwc(a, code), to_nhwc(bb, code)
        out = torch.zeros(B, H, W, 48, dtype=tdt(code), device=DEV)
        ops.conv2d([ops.sl(ta, 0, 32, shift=1), ops.sl(tb)], pc, B, H, W, 1, k // 2, ops.sl(out), "silu")
        torch.cuda.synchronize()
        cat = torch.cat([F.interpolate(q(a, code), scale_factor=2, mode="nearest"), q(bb, code)], 1)
        ref = F.silu(F.conv2d(cat, q(w, code), bias, 1, k // 2))
        assert rel(from_nhwc(out, 0, 48), ref) < tol(code)


@pytest.mark.parametrize("B,H,W,cins,cout,act", [
    (21, 57, 55, (128,), 128, "silu"),       # M = 65835: the last block holds 11 pixels
    (64, 40, 40, (256,), 256, "silu"),       # two passes, 128 KB of weights in LDS (dark4 conv3)
    (17, 64, 61, (128, 128), 192, "none"),   # concatenated input, second pass of four tiles, no activation
    (9, 96, 80, (64, 64), 64, "silu"),       # one pass of four tiles
    (16, 80, 80, (32, 224), 128, "silu"),    # uneven segments
    (2, 16, 20, (128,), 128, "silu"),        # too few pixels for this kernel: the implicit-GEMM kernel takes it
])
def test_conv1x1_stream_kernel_matches_torch(B, H, W, cins, cout, act):
    """The thin 1x1 layers (BaseConv k1 of CSPLayer / the neck, network/model_block.py:31-60) on the LDS-resident-weight
    streaming kernel (conv_stream.hip): channel slices with offsets, concatenated inputs, ragged pixel counts."""
    ops = _ops()
    g = torch.Generator().manual_seed(23)
    K = sum(cins)
    xs = [torch.randn(B, c, H, W, generator=g) for c in cins]
    w = torch.randn(cout, K, 1, 1, generator=g) / math.sqrt(K)
    bias = torch.randn(cout, generator=g) * 0.1
    pc = ops.PackedConv(w, bias, 1, DEV)
    ts = [to_nhwc(x, 1, ldc=x.shape[1] + 16 * i, coff=8 * i) for i, x in enumerate(xs)]
    out = torch.full((B, H, W, cout + 24), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.conv2d([ops.sl(t, 8 * i, c) for i, (t, c) in enumerate(zip(ts, cins))], pc, B, H, W, 1, 0, ops.sl(out, 8, cout), act)
    torch.cuda.synchronize()
    ref = F.conv2d(torch.cat([q(x, 1) for x in xs], 1), q(w, 1), bias)
    if act == "silu":
        ref = F.silu(ref)
    assert rel(from_nhwc(out, 8, cout), ref) < tol(1)
    # channels around the slice are untouched
    assert bool((out[..., :8] == 7.0).all()) and bool((out[..., 8 + cout:] == 7.0).all())


@pytest.mark.parametrize("cins,cout,up", [((128,), 128, 0), ((128, 128), 256, 0), ((128, 128), 128, 1)])
def test_conv1x1_stream_equals_implicit_gemm_bitwise(cins, cout, up):
    """Which kernel a thin 1x1 layer gets depends on its pixel count (conv_stream.hip takes 65536 and more), so the two must
    agree bit for bit: a 16-image batch (streaming kernel) against its two halves (implicit-GEMM kernel).  up = 1: the first
    input is a half-resolution map read through nearest upsampling (the neck's torch.cat([upsample(a), b]), network/net.py:208-213);
    that case is also checked against torch."""
    ops = _ops()
    g = torch.Generator().manual_seed(29)
    B, H, W = 16, 64, 64
    K = sum(cins)
    shifts = [up] + [0] * (len(cins) - 1)
    ts = [(torch.randn(B, H >> sh, W >> sh, c, generator=g) * 2).to(torch.bfloat16).to(DEV) for c, sh in zip(cins, shifts)]
    w = torch.randn(cout, K, 1, 1, generator=g) / math.sqrt(K)
    bias = torch.randn(cout, generator=g)
    pc = ops.PackedConv(w, bias, 1, DEV)
    whole = torch.zeros(B, H, W, cout, dtype=torch.bfloat16, device=DEV)
    ops.conv2d([ops.sl(t, 0, c, shift=sh) for t, c, sh in zip(ts, cins, shifts)], pc, B, H, W, 1, 0, ops.sl(whole), "silu")
    halves = torch.zeros_like(whole)
    for h in range(2):
        sub = [t[h * 8:(h + 1) * 8].contiguous() for t in ts]
        out = torch.zeros(8, H, W, cout, dtype=torch.bfloat16, device=DEV)
        ops.conv2d([ops.sl(t, 0, c, shift=sh) for t, c, sh in zip(sub, cins, shifts)], pc, 8, H, W, 1, 0, ops.sl(out), "silu")
        halves[h * 8:(h + 1) * 8] = out
    torch.cuda.synchronize()
    assert torch.equal(whole.view(torch.int16), halves.view(torch.int16))
    if up:
        nchw = [t.float().cpu().permute(0, 3, 1, 2) for t in ts]
        cat = torch.cat([F.interpolate(nchw[0], scale_factor=2, mode="nearest")] + nchw[1:], 1)
        ref = F.silu(F.conv2d(cat, q(w, 1), bias))
        assert rel(from_nhwc(whole, 0, cout), ref) < tol(1)


@pytest.mark.parametrize("B,H,W,cins,cout,act,res", [
    (64, 20, 20, (512,), 512, "silu", False),     # dark5 / MobileViT size
    (5, 20, 20, (128,), 128, "silu", True),       # bottleneck with shortcut
    (3, 13, 27, (64, 128), 96, "silu", False),    # H != W, concatenated input, Cout no multiple of the tile
    (1, 5, 5, (64,), 64, "none", False),          # one image smaller than a tile
    (7, 10, 10, (256,), 192, "silu", True),
])
def test_conv3x3_linear_patch_geometry(B, H, W, cins, cout, act, res):
    """3x3 stride-1 layers on maps whose width is no multiple of 8 (the 20 x 20 level: Bottleneck / MobileViT convs,
    network/model_block.py:193-218, 290-297) on the halo-patch kernel's linear geometry: images side by side in one zero-padded
    space, tiles that straddle image boundaries, pad positions never stored."""
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    K = sum(cins)
    xs = [torch.randn(B, c, H, W, generator=g) for c in cins]
    w = torch.randn(cout, K, 3, 3, generator=g) / math.sqrt(K * 9)
    bias = torch.randn(cout, generator=g) * 0.1
    pc = ops.PackedConv(w, bias, 1, DEV)
    ts = [to_nhwc(x, 1, ldc=x.shape[1] + 16 * i, coff=8 * i) for i, x in enumerate(xs)]
    r = torch.randn(B, cout, H, W, generator=g)
    tr = to_nhwc(r, 1)
    out = torch.full((B, H, W, cout + 8), 3.0, dtype=torch.bfloat16, device=DEV)
    ops.conv2d([ops.sl(t, 8 * i, c) for i, (t, c) in enumerate(zip(ts, cins))], pc, B, H, W, 1, 1, ops.sl(out, 0, cout), act,
               ops.sl(tr) if res else None)
    torch.cuda.synchronize()
    ref = F.conv2d(torch.cat([q(x, 1) for x in xs], 1), q(w, 1), bias, 1, 1)
    if act == "silu":
        ref = F.silu(ref)
    if res:
        ref = ref + q(r, 1)
    assert rel(from_nhwc(out, 0, cout), ref) < tol(1)
    assert bool((out[..., cout:] == 3.0).all())


@pytest.mark.parametrize("B,C,H,W,cout", [(3, 64, 40, 40, 128), (64, 256, 40, 40, 256), (2, 32, 24, 56, 64), (5, 128, 26, 18, 96)])
def test_s2d_conv_linear_patch_geometry(B, C, H, W, cout):
    """SpaceToDepth + BaseConv k3 in its 6x6 stride-2 pad-2 form (network/model_block.py:166-175) with an output map whose
    width is no multiple of 8 (dark5 and the last bottom-up conv write 20 x 20): linear geometry of the halo-patch kernel,
    64-channel chunks that are one (dy, dx) phase (C >= 64) or a dx pair (C = 32)."""
    ops = _ops()
    g = torch.Generator().manual_seed(37)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(cout, C, 6, 6, generator=g) / math.sqrt(C * 36)
    bias = torch.randn(cout, generator=g) * 0.1
    pc = ops.PackedConv(w, bias, 1, DEV)
    tx = to_nhwc(x, 1)
    Ho, Wo = H // 2, W // 2
    out = torch.zeros(B, Ho, Wo, cout, dtype=torch.bfloat16, device=DEV)
    ops.conv2d([ops.sl(tx)], pc, B, H, W, 2, 2, ops.sl(out), "silu")
    torch.cuda.synchronize()
    ref = F.silu(F.conv2d(q(x, 1), q(w, 1), bias, 2, 2))
    assert rel(from_nhwc(out, 0, cout), ref) < tol(1)


HALO_CASES = [
    # B, H, W (stored input), cins, cout, s2d, act, residual          geometry the kernel picks
    (2, 32, 48, (64,), 128, False, "silu", False),      # 16 x 16 tiles, BN 128
    (3, 16, 32, (32, 96), 64, False, "none", True),     # two segments, BN 64, residual
    (2, 24, 40, (128,), 128, False, "silu", True),      # linear (Wp = 42), partial last tile
    (5, 20, 20, (256,), 256, False, "relu", False),     # linear, two column blocks, tiles straddle images
    (1, 5, 5, (64,), 64, False, "silu", False),         # one image smaller than a tile
    (2, 50, 56, (64,), 192, False, "silu", False),      # 16 x 16 tiles with ragged edges in both directions, BN 64 (192 = 3 x 64)
    (2, 64, 96, (32,), 64, True, "silu", False),        # SpaceToDepth, C = 32: a chunk is one (dy, dx)
    (3, 40, 40, (64,), 128, True, "silu", False),       # SpaceToDepth to a 20 x 20 map: linear
    (2, 32, 64, (128,), 256, True, "lrelu", False),     # SpaceToDepth, C = 128: four chunks per (dy, dx)
    (64, 80, 80, (128,), 128, False, "silu", True),     # a network-sized layer: many tiles per workgroup, the ring wraps per tile
]


@pytest.mark.parametrize("case", HALO_CASES)
def test_conv_halo_matches_torch(case):
    """ccnet_conv_halo (conv_halo.hip): BaseConv k3 s1 p1 and SpaceToDepth + BaseConv k3 (network/model_block.py:37-58, 152-175) with
    register-streamed weights, against torch fp32 on the same bf16 operands; channel slices with offsets, untouched neighbours."""
    ops = _ops()
    B, H, W, cins, cout, s2d, act, use_res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    K = sum(cins)
    xs = [torch.randn(B, c, H, W, generator=g) for c in cins]
    cv = 4 * K if s2d else K
    w = torch.randn(cout, cv, 3, 3, generator=g) / math.sqrt(cv * 9)
    bias = torch.randn(cout, generator=g) * 0.1
    pk = ops.PackedHalo(w, bias, DEV)
    ts = [to_nhwc(x, 1, ldc=x.shape[1] + 16 * (i + 1), coff=8 * (i + 1)) for i, x in enumerate(xs)]
    Ho, Wo = (H // 2, W // 2) if s2d else (H, W)
    out = torch.full((B, Ho, Wo, cout + 16), 5.0, dtype=torch.bfloat16, device=DEV)
    r = torch.randn(B, cout, Ho, Wo, generator=g)
    tr = to_nhwc(r, 1, ldc=cout + 8, coff=4)
    ops.conv_halo([ops.sl(t, 8 * (i + 1), c) for i, (t, c) in enumerate(zip(ts, cins))], pk, s2d, B, H, W, ops.sl(out, 8, cout), act,
                  ops.sl(tr, 4, cout) if use_res else None)
    torch.cuda.synchronize()
    xin = torch.cat([q(x, 1) for x in xs], 1)
    if s2d:   # SpaceToDepth.forward: TL, TR, BL, BR on the channel axis (model_block.py:169-173)
        xin = torch.cat([xin[..., ::2, ::2], xin[..., ::2, 1::2], xin[..., 1::2, ::2], xin[..., 1::2, 1::2]], 1)
    actf = {"silu": F.silu, "relu": F.relu, "relu6": F.relu6, "lrelu": lambda t: F.leaky_relu(t, 0.1), "none": lambda t: t}[act]
    ref = actf(F.conv2d(xin, q(w, 1), bias, 1, 1))
    if use_res:
        ref = ref + q(r, 1)
    assert rel(from_nhwc(out, 8, cout), ref) < tol(1), rel(from_nhwc(out, 8, cout), ref)
    assert bool((out[..., :8] == 5.0).all()) and bool((out[..., 8 + cout:] == 5.0).all())


def test_conv_halo_is_run_to_run_identical_and_close_to_the_lds_weight_kernel():
    """The same layer on ccnet_conv_halo twice (bit-identical) and on ccnet_conv2d's halo-patch kernel (same rounding points, another
    accumulation order: within two bf16 ulps of each other relative to the output scale)."""
    ops = _ops()
    g = torch.Generator().manual_seed(41)
    B, H, W, C, cout = 16, 40, 40, 128, 128
    x = (torch.randn(B, H, W, C, generator=g)).to(torch.bfloat16).to(DEV)
    w = torch.randn(cout, C, 3, 3, generator=g) / math.sqrt(C * 9)
    bias = torch.randn(cout, generator=g) * 0.1
    pk, pc = ops.PackedHalo(w, bias, DEV), ops.PackedConv(w, bias, 1, DEV)
    outs = []
    for _ in range(3):
        o = torch.zeros(B, H, W, cout, dtype=torch.bfloat16, device=DEV)
        ops.conv_halo([ops.sl(x)], pk, False, B, H, W, ops.sl(o), "silu")
        outs.append(o)
    ref = torch.zeros(B, H, W, cout, dtype=torch.bfloat16, device=DEV)
    ops.conv2d([ops.sl(x)], pc, B, H, W, 1, 1, ops.sl(ref), "silu")
    torch.cuda.synchronize()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)) and torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16))
    assert float((outs[0].float() - ref.float()).abs().max() / ref.float().abs().max()) < 1.6e-2
    # the same images inside a batch eight times as large (other tile count per workgroup, other rounds): same bits
    xb = x.repeat(8, 1, 1, 1)
    ob = torch.zeros(8 * B, H, W, cout, dtype=torch.bfloat16, device=DEV)
    ops.conv_halo([ops.sl(xb)], pk, False, 8 * B, H, W, ops.sl(ob), "silu")
    torch.cuda.synchronize()
    assert torch.equal(ob[:B].view(torch.int16), outs[0].view(torch.int16)) and torch.equal(ob[7 * B:].view(torch.int16), outs[0].view(torch.int16))


HALO_S2_CASES = [
    # B, H, W, C, cout, act, residual
    (2, 32, 32, 64, 128, "silu", False),      # 16 x 16 outputs: one 2D tile per image
    (3, 40, 40, 32, 64, "silu", False),       # 20-wide outputs: linear geometry, 64-channel tiles, one chunk per phase
    (1, 18, 22, 96, 64, "relu", False),       # 9 x 11 outputs: partial tiles, three chunks per phase, runtime activation
    (2, 160, 96, 64, 128, "silu", True),      # 80 x 48 outputs: 2D tiles with a partial column of tiles, residual
    (2, 24, 40, 128, 256, "none", False),     # two column blocks of workgroups
]


@pytest.mark.parametrize("case", HALO_S2_CASES)
def test_conv_halo_stride2_matches_torch_and_the_implicit_gemm(case):
    """ccnet_conv_halo mode 2 (BaseConv k3 s2 p1 as five two-tap passes over the phases of the 2x2 pixel groups) against torch and
    against ccnet_conv2d on the same operands (same rounding points, another accumulation order), twice (bit-identical)."""
    ops = _ops()
    B, H, W, C, cout, act, use_res = case
    g = torch.Generator().manual_seed(77 + C + H)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(cout, C, 3, 3, generator=g) / math.sqrt(C * 9)
    bias = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, cout, H // 2, W // 2, generator=g) if use_res else None
    xt, rt = to_nhwc(x, 1), (to_nhwc(res, 1) if use_res else None)
    pk, pc = ops.PackedHaloS2(w, bias, DEV), ops.PackedConv(w, bias, 1, DEV)
    outs = []
    for _ in range(2):
        o = torch.zeros(B, H // 2, W // 2, cout, dtype=torch.bfloat16, device=DEV)
        ops.conv_halo([ops.sl(xt)], pk, 2, B, H, W, ops.sl(o), act, ops.sl(rt) if use_res else None)
        outs.append(o)
    ref = torch.zeros(B, H // 2, W // 2, cout, dtype=torch.bfloat16, device=DEV)
    ops.conv2d([ops.sl(xt)], pc, B, H, W, 2, 1, ops.sl(ref), act, ops.sl(rt) if use_res else None)
    torch.cuda.synchronize()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    y = F.conv2d(q(x, 1), q(w, 1), bias, 2, 1)
    y = {"silu": F.silu, "relu": F.relu, "none": lambda v: v}[act](y)
    if use_res:
        y = y + q(res, 1)
    assert rel(from_nhwc(outs[0], 0, cout), y) < tol(1)
    assert float((outs[0].float() - ref.float()).abs().max() / ref.float().abs().max()) < 1.6e-2


PW_CASES = [
    # B, H, W, [(cin, shift, ldc, coff)], cout, act, residual
    (4, 20, 20, [(512, 0, 512, 0)], 512, "silu", False),
    (3, 40, 40, [(256, 1, 256, 0), (256, 0, 256, 0)], 256, "silu", False),     # upsampled first segment (PAN concat)
    (2, 12, 20, [(128, 0, 128, 0)], 64, "none", True),                          # 64-channel tiles, a partial tile of 224 rows
    (1, 7, 9, [(64, 0, 64, 0), (64, 0, 64, 0)], 128, "relu", False),            # one partial tile (63 pixels), runtime activation
    (2, 16, 16, [(192, 0, 320, 64)], 128, "silu", True),                        # a slice of a wider buffer, three chunks
    (5, 8, 8, [(64, 2, 64, 0), (128, 1, 192, 64)], 192, "silu", False),         # both segments upsampled (x4, x2)
    (72, 40, 40, [(256, 0, 256, 0)], 256, "silu", False),                       # 900 workgroups: several tiles per workgroup
    (64, 20, 20, [(256, 0, 256, 0), (256, 0, 256, 0)], 256, "silu", True),      # 200 workgroups, two segments, a residual
]


@pytest.mark.parametrize("case", PW_CASES)
def test_conv_pw_equals_conv2d_bitwise_and_matches_torch(case):
    """ccnet_conv_pw (pixels in LDS, weights streamed into registers) against ccnet_conv2d on the same operands: bit for bit (same
    MFMA shape and k order, fp32 accumulators from zero, bias added to the finished sum), and both against torch."""
    ops = _ops()
    B, H, W, segs, cout, act, use_res = case
    g = torch.Generator().manual_seed(101 + cout + H)
    cin = sum(c for c, _, _, _ in segs)
    xs, ts, sls = [], [], []
    for c, sh, ldc, coff in segs:
        x = torch.randn(B, c, H >> sh, W >> sh, generator=g)
        t = to_nhwc(x, 1, ldc=ldc, coff=coff)
        xs.append(F.interpolate(q(x, 1), scale_factor=1 << sh, mode="nearest") if sh else q(x, 1))
        ts.append(t)
        sls.append(ops.sl(t, coff, c, shift=sh))
    w = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
    bias = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, cout, H, W, generator=g) if use_res else None
    rt = to_nhwc(res, 1) if use_res else None
    pk, pc = ops.PackedPw(w, bias, DEV), ops.PackedConv(w, bias, 1, DEV)
    out = torch.zeros(B, H, W, cout + 8, dtype=torch.bfloat16, device=DEV)
    ref = torch.zeros(B, H, W, cout + 8, dtype=torch.bfloat16, device=DEV)
    ops.conv_pw(sls, pk, B, H, W, ops.sl(out, 4, cout), act, ops.sl(rt) if use_res else None)
    ops.conv2d(sls, pc, B, H, W, 1, 0, ops.sl(ref, 4, cout), act, ops.sl(rt) if use_res else None)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    y = F.conv2d(torch.cat(xs, 1), q(w, 1), bias)
    y = {"silu": F.silu, "relu": F.relu, "none": lambda v: v}[act](y)
    if use_res:
        y = y + q(res, 1)
    assert rel(from_nhwc(out, 4, cout), y) < tol(1)
    assert float(out[..., :4].float().abs().max()) == 0.0 and float(out[..., 4 + cout:].float().abs().max()) == 0.0   # neighbours untouched


BNECK_CASES = [
    # B, C, H, W, shortcut, in (ldc, coff), out (ldc, coff)
    (3, 128, 40, 40, True, (256, 0), (128, 0)),      # dark4's bottlenecks: x_1 is the first half of [x_1 | x_2]; linear geometry
    (2, 64, 80, 80, True, (128, 0), (64, 0)),        # net_deploy dark3: 16 x 16 tiles
    (5, 128, 20, 20, False, (128, 0), (256, 128)),   # no shortcut (PAN), output into a slice of a wider buffer
    (1, 64, 23, 37, True, (64, 0), (64, 0)),         # partial 16 x 16 tiles
    (2, 128, 9, 30, True, (136, 8), (128, 0)),       # linear geometry with a partial last tile, input slice at a channel offset
    (40, 64, 80, 80, True, (64, 0), (64, 0)),        # 1000 tiles on 512 workgroup slots: second tiles, the ring across the tile boundary
    (48, 128, 40, 40, True, (128, 0), (128, 0)),     # 325 tiles on 256 slots, linear geometry
    (4, 256, 20, 20, True, (512, 0), (256, 0)),      # 256 channels: tiles of 128 linear positions, eight planes
    (64, 256, 20, 20, False, (256, 0), (256, 0)),    # ... 232 tiles, no shortcut
    (140, 256, 20, 20, True, (256, 0), (256, 0)),    # ... 506 tiles on 256 slots
    (3, 256, 13, 21, True, (256, 0), (512, 256)),    # ... the widest map it takes, output at a channel offset
]


@pytest.mark.parametrize("case", BNECK_CASES)
def test_bneck_fused_equals_the_two_launches_bitwise_and_matches_torch(case):
    """ccnet_bneck_fused (1x1 -> 3x3 [+ x] with the hidden map in LDS) against ccnet_conv2d followed by ccnet_conv_halo on the same
    operands: bit for bit (same MFMA shape, k order, bias placement, rounding points), and against torch (model_block.py:193-218)."""
    ops = _ops()
    B, Cc, H, W, shortcut, (ildc, icoff), (oldc, ocoff) = case
    g = torch.Generator().manual_seed(7 + Cc + H)
    x = torch.randn(B, Cc, H, W, generator=g)
    w1 = torch.randn(Cc, Cc, 1, 1, generator=g) / math.sqrt(Cc)
    b1 = torch.randn(Cc, generator=g) * 0.3          # SiLU(b1) != 0: rows outside the image must be zero, not conv1(0)
    w2 = torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(9 * Cc)
    b2 = torch.randn(Cc, generator=g) * 0.1
    xt = to_nhwc(x, 1, ldc=ildc, coff=icoff)
    xs = ops.sl(xt, icoff, Cc)
    pk = ops.PackedBneck(w1, b1, w2, b2, DEV)
    out = torch.zeros(B, H, W, oldc, dtype=torch.bfloat16, device=DEV)
    ops.bneck_fused(xs, pk, B, H, W, shortcut, "silu", ops.sl(out, ocoff, Cc))
    out2 = torch.zeros_like(out)
    ops.bneck_fused(xs, pk, B, H, W, shortcut, "silu", ops.sl(out2, ocoff, Cc))
    hid = torch.zeros(B, H, W, Cc, dtype=torch.bfloat16, device=DEV)
    ref = torch.zeros(B, H, W, oldc, dtype=torch.bfloat16, device=DEV)
    ops.conv2d([xs], ops.PackedConv(w1, b1, 1, DEV), B, H, W, 1, 0, ops.sl(hid), "silu")
    ops.conv_halo([ops.sl(hid)], ops.PackedHalo(w2, b2, DEV), 0, B, H, W, ops.sl(ref, ocoff, Cc), "silu", xs if shortcut else None)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))
    h = q(F.silu(F.conv2d(q(x, 1), q(w1, 1), b1)), 1)
    y = F.silu(F.conv2d(h, q(w2, 1), b2, padding=1))
    if shortcut:
        y = y + q(x, 1)
    assert rel(from_nhwc(out, ocoff, Cc), y) < tol(1)
    if oldc > Cc:
        mask = torch.ones(oldc, dtype=torch.bool)
        mask[ocoff:ocoff + Cc] = False
        assert float(out[..., mask.to(DEV)].float().abs().max()) == 0.0   # neighbours untouched


@pytest.mark.parametrize("cout,cin", [(256, 256), (128, 128), (256, 512)])
def test_conv1x1_two_segments_equal_the_concatenated_buffer_bitwise(cout, cin):
    """conv3 of a CSPLayer reads [x_1 | x_2] either as one buffer or as two segments (graph.py:csp with one fused bottleneck): the k
    order is the same, so the results must be equal bit for bit, whichever 1x1 kernel the shape selects."""
    ops = _ops()
    from ccnet_amd.network import graph
    B, H, W = 3, 20, 20
    g = torch.Generator().manual_seed(5 + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
    bias = torch.randn(cout, generator=g) * 0.1
    whole = to_nhwc(x, 1)
    a, b = whole[..., :cin // 2].contiguous(), whole[..., cin // 2:].contiguous()
    comp = graph.Compiler({}, {}, B, H, W, 1, DEV, "silu")
    one = comp.conv_w(w.double(), bias.double(), [graph.T(whole, 0, cin, H, W)], 1, 0, "silu")
    two = comp.conv_w(w.double(), bias.double(), [graph.T(a, 0, cin // 2, H, W), graph.T(b, 0, cin // 2, H, W)], 1, 0, "silu")
    torch.cuda.synchronize()
    assert torch.equal(one.buf.view(torch.int16), two.buf.view(torch.int16))


def test_bneck_fused_random_shapes_equal_the_two_launches_bitwise():
    """40 seeded random problems (all three channel counts, both geometries, maps from 3 x 3 up, batches that give one tile and
    several tiles per workgroup, with and without the shortcut, slices of wider buffers) against ccnet_conv2d + ccnet_conv_halo."""
    ops = _ops()
    rng = np.random.RandomState(11)
    g = torch.Generator().manual_seed(11)
    for case in range(40):
        Cc = int(rng.choice([64, 128, 256]))
        if Cc == 256:
            H, W = int(rng.randint(3, 30)), int(rng.randint(3, 22))
        else:
            H, W = int(rng.randint(3, 70)), int(rng.randint(3, 90))
        B = int(rng.choice([1, 2, 3, 7, 33])) if H * W < 1500 else int(rng.choice([1, 2, 5]))
        if case % 10 == 0:
            B = max(B, (600 * (128 if Cc == 256 else 256)) // (H * W) + 1)      # more tiles than workgroup slots
        shortcut = bool(rng.randint(2))
        ipad, opad = int(rng.choice([0, 8, 64])), int(rng.choice([0, 4, 32]))
        x = torch.randn(B, Cc, H, W, generator=g)
        w1 = torch.randn(Cc, Cc, 1, 1, generator=g) / math.sqrt(Cc)
        b1 = torch.randn(Cc, generator=g) * 0.3
        w2 = torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(9 * Cc)
        b2 = torch.randn(Cc, generator=g) * 0.1
        xt = to_nhwc(x, 1, ldc=Cc + ipad, coff=ipad)
        xs = ops.sl(xt, ipad, Cc)
        out = torch.zeros(B, H, W, Cc + opad, dtype=torch.bfloat16, device=DEV)
        ref = torch.zeros_like(out)
        hid = torch.zeros(B, H, W, Cc, dtype=torch.bfloat16, device=DEV)
        ops.bneck_fused(xs, ops.PackedBneck(w1, b1, w2, b2, DEV), B, H, W, shortcut, "silu", ops.sl(out, opad, Cc))
        ops.conv2d([xs], ops.PackedConv(w1, b1, 1, DEV), B, H, W, 1, 0, ops.sl(hid), "silu")
        ops.conv_halo([ops.sl(hid)], ops.PackedHalo(w2, b2, DEV), 0, B, H, W, ops.sl(ref, opad, Cc), "silu", xs if shortcut else None)
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), (case, B, Cc, H, W, shortcut, ipad, opad)


def test_bneck_fused_rejects_what_it_does_not_take():
    ops = _ops()
    pk = ops.PackedBneck(torch.zeros(64, 64, 1, 1), torch.zeros(64), torch.zeros(64, 64, 3, 3), torch.zeros(64), DEV)
    x = torch.zeros(1, 8, 8, 128, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(Exception):
        ops.bneck_fused(ops.sl(x, 0, 64), pk, 1, 8, 8, True, "silu", ops.sl(x, 0, 64))     # in place: a tile's halo is its neighbours' output
    with pytest.raises(Exception):
        ops.bneck_fused(ops.sl(x, 0, 64), pk, 1, 8, 8, True, "relu", ops.sl(x, 64, 64))    # SiLU only
    with pytest.raises(Exception):
        ops.bneck_fused(ops.sl(x, 0, 32), pk, 1, 8, 8, True, "silu", ops.sl(x, 64, 32))    # 32 channels
    pk256 = ops.PackedBneck(torch.zeros(256, 256, 1, 1), torch.zeros(256), torch.zeros(256, 256, 3, 3), torch.zeros(256), DEV)
    x256 = torch.zeros(1, 8, 24, 256, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(Exception):
        ops.bneck_fused(ops.sl(x256), pk256, 1, 8, 24, True, "silu", ops.sl(x256.clone()))  # 256 channels on a map wider than 21
    ops.bneck_fused(ops.sl(x, 0, 64), pk, 1, 8, 8, True, "silu", ops.sl(x, 64, 64))        # the other half of the same buffer is fine
    torch.cuda.synchronize()


def test_conv_pw_rejects_what_it_does_not_take():
    ops = _ops()
    with pytest.raises(Exception):
        ops.PackedPw(torch.zeros(64, 96, 1, 1), None, DEV)          # Cin % 64
    pk = ops.PackedPw(torch.zeros(64, 64, 1, 1), None, DEV)
    x = torch.zeros(1, 4, 4, 64, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(1, 4, 4, 64, dtype=torch.float32, device=DEV)
    with pytest.raises(Exception):
        ops.conv_pw([ops.sl(x, 0, 32)], pk, 1, 4, 4, ops.sl(x.clone()), "silu")    # a 32-channel slice
    with pytest.raises(Exception):
        ops.conv_pw([ops.sl(x, shift=3)], pk, 1, 4, 4, ops.sl(x.clone()), "silu")  # upsampling by 8


@pytest.mark.parametrize("code", [0, 1])
def test_conv3x3_two_segments_patch_kernel(code):
    """3x3 over torch.cat([a, b], 1) at a 16-aligned width (MobileViT conv4, network/model_block.py:296-297): the halo-patch
    kernel switches segment per 128-byte channel chunk."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 12, 32
    a = torch.randn(B, 64, H, W, generator=g)
    bb = torch.randn(B, 128, H, W, generator=g)
    w = torch.randn(96, 192, 3, 3, generator=g) / math.sqrt(192 * 9)
    bias = torch.randn(96, generator=g) * 0.1
    pc = ops.PackedConv(w, bias, code, DEV)
    ta, tb = to_nhwc(a, code, ldc=80, coff=16), to_nhwc(bb, code)
    out = torch.zeros(B, H, W, 96, dtype=tdt(code), device=DEV)
    ops.conv2d([ops.sl(ta, 16, 64), ops.sl(tb)], pc, B, H, W, 1, 1, ops.sl(out), "silu")
    torch.cuda.synchronize()
    ref = F.silu(F.conv2d(torch.cat([q(a, code), q(bb, code)], 1), q(w, code), bias, 1, 1))
    assert rel(from_nhwc(out, 0, 96), ref) < tol(code)


@pytest.mark.parametrize("code", [0, 1])
def test_s2d_conv_equals_6x6(code):
    """SpaceToDepth + BaseConv k3 (network/model_block.py:166-175) through the 6x6 stride-2 weight transform."""
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    B, Cc, H, W = 2, 16, 20, 12
    x = torch.randn(B, Cc, H, W, generator=g)
    w3 = torch.randn(40, 4 * Cc, 3, 3, generator=g) / math.sqrt(36 * Cc)
    pc = ops.PackedConv(ops.s2d_to_6x6(w3), None, code, DEV)
    out = torch.zeros(B, H // 2, W // 2, 40, dtype=tdt(code), device=DEV)
    xin = to_nhwc(x, code)
    ops.conv2d([ops.sl(xin)], pc, B, H, W, 2, 2, ops.sl(out), "none")
    torch.cuda.synchronize()
    xq = q(x, code)
    s2d = torch.cat((xq[..., ::2, ::2], xq[..., ::2, 1::2], xq[..., 1::2, ::2], xq[..., 1::2, 1::2]), 1)
    ref = F.conv2d(s2d, q(w3, code), None, 1, 1)
    assert rel(from_nhwc(out, 0, 40), ref) < tol(code)


@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("cout,H,W", [(32, 64, 64), (24, 34, 70), (64, 16, 32)])
def test_stem_conv(code, cout, H, W):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    B = 2
    x = torch.randn(B, 3, H, W, generator=g)
    w6 = torch.randn(cout, 3, 6, 6, generator=g) / math.sqrt(108)
    b = torch.randn(cout, generator=g) * 0.1
    wp, bp = ops.pack_stem(w6, b, DEV)
    out = torch.zeros(B, H // 2, W // 2, cout, dtype=tdt(code), device=DEV)
    xd = x.to(DEV)
    ops.stem_conv(xd, wp, bp, cout, "silu", ops.sl(out), code)
    torch.cuda.synchronize()
    ref = F.silu(F.conv2d(q(x, code), q(w6, code), b, 2, 2))
    assert rel(from_nhwc(out, 0, cout), ref) < tol(code)


@pytest.mark.parametrize("code", [0, 1])
def test_dwconv3x3(code):
    ops = _ops()
    g = torch.Generator().manual_seed(13)
    B, Cc, H, W = 2, 64, 7, 9
    x = torch.randn(B, Cc, H, W, generator=g)
    w = torch.randn(Cc, 1, 3, 3, generator=g) / 3
    b = torch.randn(Cc, generator=g) * 0.1
    out = torch.zeros(B, H, W, Cc, dtype=tdt(code), device=DEV)
    wt, bd, xin = w.reshape(Cc, 9).t().contiguous().to(DEV), b.to(DEV), to_nhwc(x, code)
    ops.dwconv3x3(ops.sl(xin), code, B, H, W, wt, bd, ops.sl(out))
    torch.cuda.synchronize()
    ref = F.conv2d(q(x, code), w, b, 1, 1, 1, Cc)
    assert rel(from_nhwc(out, 0, Cc), ref) < (1e-5 if code == 0 else 6e-3)


@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("H,W", [(20, 20), (7, 5), (40, 40), (80, 72)])  # 80x72 > 5088 pixels: the global-memory fallback
def test_sppf_pool_exact(code, H, W):
    ops = _ops()
    g = torch.Generator().manual_seed(17)
    B, Cc = 2, 32
    x = q(torch.randn(B, Cc, H, W, generator=g), code)
    buf = to_nhwc(x, code, ldc=4 * Cc)
    ops.sppf_pool(ops.sl(buf, 0, Cc), code, B, H, W)
    torch.cuda.synchronize()
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    ref = torch.cat((x, y1, y2, y3), 1)
    assert torch.equal(from_nhwc(buf, 0, 4 * Cc), ref)  # max is exact in any dtype


@pytest.mark.parametrize("code", [0, 1])
def test_maxpool3x3s2_exact(code):
    ops = _ops()
    g = torch.Generator().manual_seed(19)
    B, Cc, H, W = 2, 16, 11, 14
    x = q(torch.randn(B, Cc, H, W, generator=g), code)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.zeros(B, Ho, Wo, Cc, dtype=tdt(code), device=DEV)
    xin = to_nhwc(x, code)
    ops.maxpool3x3s2(ops.sl(xin), code, B, H, W, ops.sl(out))
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(out, 0, Cc), F.max_pool2d(x, 3, 2, 1))


def _cbam_ref(x, fc1, fc2, w7):
    def mlp(v):
        return F.conv2d(F.relu(F.conv2d(v, fc1[:, :, None, None])), fc2[:, :, None, None])
    gate = torch.sigmoid(mlp(F.adaptive_avg_pool2d(x, 1)) + mlp(F.adaptive_max_pool2d(x, 1)))
    x = x * gate
    s = torch.cat([x.mean(1, keepdim=True), x.max(1, keepdim=True)[0]], 1)
    return x * torch.sigmoid(F.conv2d(s, w7[None], None, 1, w7.shape[-1] // 2))


@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("Cc,H,W", [(128, 28, 28), (512, 7, 7), (64, 40, 36), (24, 5, 5), (1024, 6, 6)])
def test_cbam(code, Cc, H, W):
    ops = _ops()
    g = torch.Generator().manual_seed(23)
    B = 2
    x = torch.randn(B, Cc, H, W, generator=g)
    cr = max(Cc // 8, 1)
    fc1 = torch.randn(cr, Cc, generator=g) * 2 / math.sqrt(Cc)
    fc2 = torch.randn(Cc, cr, generator=g) * 2 / math.sqrt(cr)
    w7 = torch.randn(2, 7, 7, generator=g) * 0.3
    scratch = torch.zeros(ops.gate_scratch_floats(B, H, W, Cc), dtype=torch.float32, device=DEV)
    out = torch.zeros(B, H, W, Cc, dtype=tdt(code), device=DEV)
    xin, d1, d2, d7 = to_nhwc(x, code), fc1.to(DEV), fc2.to(DEV), w7.to(DEV)
    ops.cbam(ops.sl(xin), code, B, H, W, d1, d2, d7, scratch, ops.sl(out))
    torch.cuda.synchronize()
    ref = _cbam_ref(q(x, code), fc1, fc2, w7)
    assert rel(from_nhwc(out, 0, Cc), ref) < (2e-5 if code == 0 else 6e-3)
    # the same scratch again: identical output
    out2 = torch.zeros(B, H, W, Cc, dtype=tdt(code), device=DEV)
    ops.cbam(ops.sl(xin), code, B, H, W, d1, d2, d7, scratch, ops.sl(out2))
    torch.cuda.synchronize()
    assert torch.equal(out2, out)


@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("B,Cc,H,W,k", [(64, 512, 20, 20, 7), (5, 256, 24, 24, 7), (3, 1024, 10, 10, 7), (2, 512, 7, 9, 3), (33, 512, 13, 21, 7)])
def test_cbam_one_workgroup_per_image_equals_the_four_launches_bitwise(code, B, Cc, H, W, k, monkeypatch):
    """Small maps run the whole gate in one workgroup per image (cbam_image_kernel): the same device functions in the same roles as
    the four launches, so the outputs must be equal bit for bit -- and match torch (attention_block.py:19-22,44-48,65-70)."""
    ops = _ops()
    g = torch.Generator().manual_seed(29 + Cc + H)
    x = torch.randn(B, Cc, H, W, generator=g)
    cr = Cc // 8
    fc1 = torch.randn(cr, Cc, generator=g) * 2 / math.sqrt(Cc)
    fc2 = torch.randn(Cc, cr, generator=g) * 2 / math.sqrt(cr)
    wk = torch.randn(2, k, k, generator=g) * 0.3
    scratch = torch.zeros(ops.gate_scratch_floats(B, H, W, Cc), dtype=torch.float32, device=DEV)
    xin, d1, d2, dk = to_nhwc(x, code, ldc=Cc + 8, coff=8), fc1.to(DEV), fc2.to(DEV), wk.to(DEV)
    outs = []
    for env in ("1", "0"):
        monkeypatch.setenv("CCNET_CBAM_IMAGE", env)
        out = torch.zeros(B, H, W, Cc + 16, dtype=tdt(code), device=DEV)
        ops.cbam(ops.sl(xin, 8, Cc), code, B, H, W, d1, d2, dk, scratch, ops.sl(out, 16, Cc))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    ref = _cbam_ref(q(x, code), fc1, fc2, wk)
    assert rel(from_nhwc(outs[0], 16, Cc), ref) < (2e-5 if code == 0 else 6e-3)
    assert float(outs[0][..., :16].float().abs().max()) == 0.0   # neighbours untouched


def test_gates_are_run_to_run_identical_and_batch_independent_on_a_large_map():
    """80x80x128, 64 images: 50 pooling blocks per image, the chunk partials summed in a fixed order split over thread groups;
    the output must not change between launches, and an image alone must equal the image in the batch."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, Cc, H, W = 64, 128, 80, 80
    x = torch.randn(B, H, W, Cc, generator=g).to(torch.bfloat16).to(DEV)
    fc1 = (torch.randn(8, Cc, generator=g) * 0.2).to(DEV)
    fc2 = (torch.randn(Cc, 8, generator=g) * 0.5).to(DEV)
    w7 = (torch.randn(2, 7, 7, generator=g) * 0.3).to(DEV)
    w3 = torch.randn(3, generator=g).to(DEV)
    scratch = torch.zeros(ops.gate_scratch_floats(B, H, W, Cc), dtype=torch.float32, device=DEV)
    outs = []
    for _ in range(4):
        o = torch.zeros(B, H, W, Cc, dtype=torch.bfloat16, device=DEV)
        ops.cbam(ops.sl(x), 1, B, H, W, fc1, fc2, w7, scratch, ops.sl(o))
        e = torch.zeros(B, H, W, Cc, dtype=torch.bfloat16, device=DEV)
        ops.eca(ops.sl(x), 1, B, H, W, w3, scratch, ops.sl(e))
        outs.append((o, e))
    torch.cuda.synchronize()
    for o, e in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(e, outs[0][1])
    # image 5 alone equals image 5 in the batch
    o1 = torch.zeros(1, H, W, Cc, dtype=torch.bfloat16, device=DEV)
    ops.cbam(ops.sl(x[5:6].contiguous()), 1, 1, H, W, fc1, fc2, w7, scratch, ops.sl(o1))
    torch.cuda.synchronize()
    assert torch.equal(o1[0], outs[0][0][5])
    gate = torch.sigmoid(F.conv1d(x.float().mean((1, 2))[:, None], w3.view(1, 1, 3), None, 1, 1))[:, 0]
    assert rel(outs[0][1].float(), x.float() * gate[:, None, None, :]) < 6e-3


@pytest.mark.parametrize("code", [0, 1])
def test_eca(code):
    ops = _ops()
    g = torch.Generator().manual_seed(29)
    B, Cc, H, W = 2, 128, 14, 14
    x = torch.randn(B, Cc, H, W, generator=g)
    w3 = torch.randn(3, generator=g)
    scratch = torch.zeros(ops.gate_scratch_floats(B, H, W, Cc), dtype=torch.float32, device=DEV)
    out = torch.zeros(B, H, W, Cc, dtype=tdt(code), device=DEV)
    xin, d3 = to_nhwc(x, code), w3.to(DEV)
    ops.eca(ops.sl(xin), code, B, H, W, d3, scratch, ops.sl(out))
    torch.cuda.synchronize()
    xq = q(x, code)
    y = F.adaptive_avg_pool2d(xq, 1)
    y = F.conv1d(y.squeeze(-1).transpose(-1, -2), w3.view(1, 1, 3), None, 1, 1).transpose(-1, -2).unsqueeze(-1)
    ref = xq * torch.sigmoid(y)
    assert rel(from_nhwc(out, 0, Cc), ref) < (2e-5 if code == 0 else 6e-3)


@pytest.mark.parametrize("code", [0, 1])
def test_asff_fuse(code):
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    B, Cc, H, W = 2, 64, 8, 12
    l0 = torch.randn(B, Cc, H // 4, W // 4, generator=g)
    l1 = torch.randn(B, Cc, H // 2, W // 2, generator=g)
    l2 = torch.randn(B, Cc, H, W, generator=g)
    lg = torch.randn(B, 3, H, W, generator=g) * 2
    out = torch.zeros(B, H, W, Cc, dtype=tdt(code), device=DEV)
    lgb = to_nhwc(lg, code, ldc=8)
    t0, t1, t2 = to_nhwc(l0, code), to_nhwc(l1, code), to_nhwc(l2, code)
    ops.asff_fuse(ops.sl(t0, shift=2), ops.sl(t1, shift=1), ops.sl(t2), ops.sl(lgb, 0, 3), code, B, H, W, ops.sl(out))
    torch.cuda.synchronize()
    wgt = torch.softmax(q(lg, code), 1)
    ref = (F.interpolate(q(l0, code), scale_factor=4, mode="nearest") * wgt[:, 0:1]
           + F.interpolate(q(l1, code), scale_factor=2, mode="nearest") * wgt[:, 1:2] + q(l2, code) * wgt[:, 2:3])
    assert rel(from_nhwc(out, 0, Cc), ref) < (2e-5 if code == 0 else 6e-3)


@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("E,rows", [(256, 100), (96, 7), (640, 33)])
def test_layernorm_residual(code, E, rows):
    ops = _ops()
    g = torch.Generator().manual_seed(37)
    x, r = torch.randn(rows, E, generator=g), torch.randn(rows, E, generator=g)
    gam, bet = torch.rand(E, generator=g) + 0.5, torch.randn(E, generator=g) * 0.1
    xd, rd = x.to(DEV).to(tdt(code)), r.to(DEV).to(tdt(code))
    out = torch.zeros(rows, E, dtype=tdt(code), device=DEV)
    gd, bd = gam.to(DEV), bet.to(DEV)
    ops.layernorm(xd, rd, gd, bd, out, code, rows, E)
    out2 = torch.zeros(rows, E, dtype=tdt(code), device=DEV)
    ops.layernorm(xd, None, gd, bd, out2, code, rows, E)
    torch.cuda.synchronize()
    ref = F.layer_norm(q(x, code) + q(r, code), (E,), gam, bet, 1e-5)
    ref2 = F.layer_norm(q(x, code), (E,), gam, bet, 1e-5)
    t = 2e-5 if code == 0 else 6e-3
    assert rel(out.float().cpu(), ref) < t and rel(out2.float().cpu(), ref2) < t


@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("Sq,Sk,N,heads,hd", [(1, 1, 49, 2, 128), (4, 4, 10, 2, 128), (5, 70, 3, 2, 96), (64, 64, 6, 2, 128)])
def test_attention_over_batch_axis(code, Sq, Sk, N, heads, hd):
    """MultiheadAttention core with batch_first=False operands (S, N, E) (network/model_block.py:288-290)."""
    ops = _ops()
    g = torch.Generator().manual_seed(41)
    E = heads * hd
    qkv_q = torch.randn(Sq, N, 3 * E, generator=g)
    kv = torch.randn(Sk, N, 3 * E, generator=g) if Sk != Sq else qkv_q
    qd, kd = qkv_q.to(DEV).to(tdt(code)), kv.to(DEV).to(tdt(code))
    out = torch.zeros(Sq, N, E, dtype=tdt(code), device=DEV)
    ops.attention(ops.sl(qd, 0, E), ops.sl(kd, E, E), ops.sl(kd, 2 * E, E), code, Sq, Sk, N, heads, hd, ops.sl(out))
    torch.cuda.synchronize()
    qq, kk, vv = q(qkv_q[..., :E], code), q(kv[..., E:2 * E], code), q(kv[..., 2 * E:], code)
    qh = qq.reshape(Sq, N * heads, hd).transpose(0, 1)
    kh = kk.reshape(Sk, N * heads, hd).transpose(0, 1)
    vh = vv.reshape(Sk, N * heads, hd).transpose(0, 1)
    att = torch.softmax(torch.bmm(qh, kh.transpose(1, 2)) / math.sqrt(hd), -1)
    ref = torch.bmm(att, vh).transpose(0, 1).reshape(Sq, N, E)
    assert rel(out.float().cpu(), ref) < (2e-5 if code == 0 else 6e-3)


def test_decode_matches_oracle_layouts():
    from oracle import ccnet_oracle as orc
    ops = _ops()
    g = torch.Generator().manual_seed(43)
    B, nc = 3, 6
    hw = [(20, 32), (10, 16), (5, 8)]
    heads = [torch.randn(B, 5 + nc, h, w, generator=g) * 2 for h, w in hw]
    A = sum(h * w for h, w in hw)
    ref = orc.decode_outputs(heads, [160, 256])
    # (a) NCHW contiguous, (b) NHWC buffers permuted to NCHW views (what Net.forward returns)
    for mode in ("nchw", "nhwc"):
        dev = []
        for h in heads:
            d = h.to(DEV)
            if mode == "nhwc":
                d = d.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
            dev.append(d)
        out = torch.zeros(B, A, 5 + nc, dtype=torch.float32, device=DEV)
        ops.decode(dev, nc, 160, 256, out)
        torch.cuda.synchronize()
        # fp32 tolerance: expf/sigmoid differ from ATen's vectorised CPU versions by a few ulp
        assert torch.allclose(out.cpu(), ref, rtol=2e-6, atol=1e-7), float((out.cpu() - ref).abs().max())


def test_scored_decode_and_filter_equal_the_plain_ones():
    """ccnet_decode_scored writes the same rows as ccnet_decode bit for bit plus obj * max class per anchor; the filter reading
    those scores keeps the same rows in the same order.  bbox.decode_outputs hands the scores to post_process only while the
    tensor is the one it returned, unmodified."""
    from ccnet_amd.utils import bbox
    ops = _ops()
    g = torch.Generator().manual_seed(44)
    for B, nc, hw in ((3, 6, [(20, 32), (10, 16), (5, 8)]), (2, 1, [(7, 9)]), (2, 59, [(33, 17), (3, 3)])):
        heads = [(torch.randn(B, 5 + nc, h, w, generator=g) * 2).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) for h, w in hw]
        A = sum(h * w for h, w in hw)
        plain = torch.zeros(B, A, 5 + nc, device=DEV)
        both = torch.zeros(B, A, 5 + nc, device=DEV)
        scores = torch.zeros(B, A, device=DEV)
        ops.decode(heads, nc, 160, 256, plain)
        ops.decode(heads, nc, 160, 256, both, scores)
        torch.cuda.synchronize()
        assert torch.equal(plain, both)
        assert torch.equal(scores, plain[..., 4] * plain[..., 5:].max(-1)[0])
        for conf in (0.05, 0.5):
            c0, n0 = torch.zeros(B, A, 7, device=DEV), torch.zeros(B, dtype=torch.int32, device=DEV)
            c1, n1 = torch.zeros(B, A, 7, device=DEV), torch.zeros(B, dtype=torch.int32, device=DEV)
            ops.filter_candidates(plain, nc, conf, c0, n0)
            ops.filter_candidates(plain, nc, conf, c1, n1, scores)
            torch.cuda.synchronize()
            assert torch.equal(n0, n1) and int(n0.sum()) > 0
            for b in range(B):
                assert torch.equal(c0[b, :int(n0[b])], c1[b, :int(n1[b])])
    with pytest.raises(Exception):   # 5 + nc > 64: no scores
        ops.decode([torch.zeros(1, 85, 4, 4, device=DEV)], 80, 64, 64, torch.zeros(1, 16, 85, device=DEV), torch.zeros(1, 16, device=DEV))
    dec = bbox.decode_outputs(heads, [160, 256])
    assert bbox._scores_of(dec, dec, 59) is not None and bbox._scores_of(dec, dec, 6) is None
    assert bbox._scores_of(dec.clone(), dec.clone(), 59) is None
    info = [[i, 320, 240] for i in range(2)]
    want = bbox.post_process(dec.clone(), 59, [160, 256], info, 0.3, 0.5)      # rows path
    got = bbox.post_process(dec, 59, [160, 256], info, 0.3, 0.5)               # scores path
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    dec[..., 4] *= 0.5                                                         # modified in place: the scores are stale and must not be used
    assert bbox._scores_of(dec, dec, 59) is None
    want = bbox.post_process(dec.clone(), 59, [160, 256], info, 0.3, 0.5)
    got = bbox.post_process(dec, 59, [160, 256], info, 0.3, 0.5)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    big = bbox.decode_outputs([torch.randn(1, 85, 4, 4, device=DEV)], [64, 64])   # 80 classes: plain decode, no scores
    assert getattr(big, "_ccnet_scores", None) is None


def _run_post(pred, nc, conf, thr, info):
    from ccnet_amd.utils import bbox
    return bbox.post_process(pred.clone(), nc, [160, 256], info, conf, thr)


@pytest.mark.parametrize("conf,thr", [(0.5, 0.6), (0.001, 0.6), (0.3, 0.4), (0.99999, 0.5)])
def test_post_process_bit_exact_vs_oracle(conf, thr):
    from oracle import ccnet_oracle as orc
    g = torch.Generator().manual_seed(47)
    B, A, nc = 3, 1500, 6
    pred = torch.rand(B, A, 5 + nc, generator=g)
    pred[..., 2:4] = pred[..., 2:4] * 0.2 + 0.01
    pred[1, :, 4] *= 0.3
    info = [[i, 300 + 7 * i, 200 + 3 * i] for i in range(B)]
    ref = orc.post_process(pred, nc, [160, 256], info, conf, thr)
    got = _run_post(pred.to(DEV), nc, conf, thr, info)
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        assert a.dtype == np.float32 and a.shape == b.shape
        assert np.array_equal(a, b)


def test_nms_known_answers():
    """Hand-checkable cases: identical boxes, touching boxes (IoU 0), IoU exactly at the threshold (kept: '>' not '>='),
    cross-class overlap kept, score ties (lower index wins), zero-area boxes (NaN IoU never suppresses), empty input."""
    from oracle import ccnet_oracle as orc
    from ccnet_amd.utils import bbox

    def run(boxes, scores, cls, thr):
        n = len(boxes)
        cand = torch.zeros(1, max(n, 1), 7)
        for i in range(n):
            cand[0, i, :4] = torch.tensor(boxes[i])
            cand[0, i, 4], cand[0, i, 5], cand[0, i, 6] = scores[i], 1.0, cls[i]
        keep = bbox.nms_keep_indices(cand.to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV), thr)[0]
        b = np.asarray(boxes, np.float32).reshape(-1, 4)
        ref = orc.batched_nms_reference(b, np.asarray(scores, np.float32), np.asarray(cls, np.float32), thr)
        assert list(keep) == list(ref), (keep, ref)
        return list(keep)

    assert run([[0, 0, 1, 1], [0, 0, 1, 1]], [0.9, 0.8], [0, 0], 0.5) == [0]
    assert run([[0, 0, 1, 1], [1, 0, 2, 1]], [0.9, 0.8], [0, 0], 0.5) == [0, 1]
    # IoU = 0.5 exactly: [0,0,2,1] vs [0,0,1,1] -> inter 1, union 2
    assert run([[0, 0, 2, 1], [0, 0, 1, 1]], [0.9, 0.8], [0, 0], 0.5) == [0, 1]
    assert run([[0, 0, 2, 1], [0, 0, 1, 1]], [0.9, 0.8], [0, 0], 0.49) == [0]
    assert run([[0, 0, 1, 1], [0, 0, 1, 1]], [0.9, 0.8], [0, 1], 0.5) == [0, 1]
    assert run([[0, 0, 1, 1], [0, 0, 1, 1], [0, 0, 1, 1]], [0.7, 0.7, 0.7], [0, 0, 0], 0.5) == [0]
    assert run([[0.5, 0.5, 0.5, 0.5], [0.5, 0.5, 0.5, 0.5]], [0.9, 0.8], [0, 0], 0.5) == [0, 1]
    assert run([], [], [], 0.5) == []


def test_nms_heavy_bit_exact():
    """SURVEY 8(d) NMS microbench shape at reduced size: log-normal boxes, 6 classes, uniform scores."""
    from oracle import ccnet_oracle as orc
    from ccnet_amd.utils import bbox
    rng = np.random.RandomState(0)
    B, n = 2, 3000
    cand = np.zeros((B, n + 50, 7), np.float32)
    counts = []
    for b in range(B):
        nb = n - 137 * b
        c = rng.uniform(0, 1, (nb, 2))
        wh = np.exp(rng.normal(math.log(0.05), 0.5, (nb, 2)))
        cand[b, :nb, 0:2] = c - wh / 2
        cand[b, :nb, 2:4] = c + wh / 2
        cand[b, :nb, 4] = rng.uniform(0, 1, nb)
        cand[b, :nb, 5] = rng.uniform(0.5, 1, nb)
        cand[b, :nb, 6] = rng.randint(0, 6, nb)
        counts.append(nb)
    keep = bbox.nms_keep_indices(torch.from_numpy(cand).to(DEV), torch.tensor(counts, dtype=torch.int32, device=DEV), 0.6)
    for b in range(B):
        nb = counts[b]
        ref = orc.batched_nms_reference(cand[b, :nb, :4], cand[b, :nb, 4] * cand[b, :nb, 5], cand[b, :nb, 6], 0.6)
        assert np.array_equal(np.asarray(keep[b], np.int64), ref)


# ------------------------------------------------------------------------------------------ fused transformer
@pytest.mark.parametrize("S,N,n_enc,n_dec", [(64, 6, 2, 2), (5, 3, 1, 2), (1, 7, 1, 1), (33, 4, 6, 6)])
def test_transformer_fused_matches_torch(S, N, n_enc, n_dec):
    """ccnet_transformer_fused (bf16) against torch.nn.Transformer(src, src) in fp32, eval mode, batch_first=False --
    the exact module MobileViT builds (model_block.py:270-271).  bf16 storage of every intermediate: rms < 3 %."""
    import torch.nn as nn
    from ccnet_amd import ops
    torch.manual_seed(S * 100 + N)
    tr = nn.Transformer(d_model=256, nhead=2, num_encoder_layers=n_enc, num_decoder_layers=n_dec, dim_feedforward=1024).eval()
    with torch.no_grad():
        for name, prm in tr.named_parameters():  # non-trivial biases and LayerNorm parameters
            if prm.dim() == 1:
                prm.copy_(torch.randn_like(prm) * 0.1 + (1.0 if "norm" in name and name.endswith("weight") else 0.0))
    sd = {"t." + k: v for k, v in tr.state_dict().items()}
    src = torch.randn(S, N, 256) * 0.8
    with torch.no_grad():
        ref = tr(src, src)
    wst, pvec, fpw = ops.pack_transformer(sd, "t", n_enc, n_dec, "cuda")
    x = src.reshape(S * N, 256).to("cuda", torch.bfloat16).contiguous()
    out = torch.empty_like(x)
    ops.transformer_fused(x, 256, out, 256, S, N, wst, fpw, pvec, n_enc, n_dec)
    torch.cuda.synchronize()
    got = out.float().cpu().reshape(S, N, 256)
    err = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert err < 0.03, "rms rel err %.3e" % err
    assert float((got - ref).abs().max()) < 0.25


def test_transformer_fused_rejects_other_shapes():
    from ccnet_amd import _lib
    L = _lib.lib()
    t = torch.zeros(64, 256, device="cuda", dtype=torch.bfloat16)
    f = torch.zeros(4, device="cuda")
    rc = L.ccnet_transformer_fused(t.data_ptr(), 256, t.data_ptr(), 256, 65, 1, 256, 2, 1024, t.data_ptr(), 0, f.data_ptr(), 0, 0, None)
    assert rc != 0 and b"S" in L.ccnet_last_error()
    rc = L.ccnet_transformer_fused(t.data_ptr(), 256, t.data_ptr(), 256, 8, 1, 512, 2, 1024, t.data_ptr(), 0, f.data_ptr(), 0, 0, None)
    assert rc != 0


def test_nms_full_size_properties():
    """SURVEY 8(d) NMS microbench at full size (16 images x 10 000 candidates): too slow for the scalar oracle, so check
    size-independent properties: kept scores are non-increasing, no kept same-class pair overlaps above the threshold,
    every rejected box is covered by a kept higher-scoring same-class box, and NMS of the kept set keeps all of it."""
    from ccnet_amd.utils import bbox
    rng = np.random.RandomState(1)
    B, n, thr = 16, 10000, 0.6
    cand = np.zeros((B, n, 7), np.float32)
    c = rng.uniform(0, 1, (B, n, 2))
    wh = np.exp(rng.normal(math.log(0.05), 0.5, (B, n, 2)))
    cand[..., 0:2] = c - wh / 2
    cand[..., 2:4] = c + wh / 2
    cand[..., 4] = rng.uniform(0, 1, (B, n))
    cand[..., 5] = 1.0
    cand[..., 6] = rng.randint(0, 6, (B, n))
    ct = torch.from_numpy(cand).to(DEV)
    counts = torch.full((B,), n, dtype=torch.int32, device=DEV)
    keep = bbox.nms_keep_indices(ct, counts, thr)

    def iou(a, b):  # (p,4) x (q,4) on the GPU, torch fp32 reference arithmetic
        lt = torch.maximum(a[:, None, :2], b[None, :, :2])
        rb = torch.minimum(a[:, None, 2:], b[None, :, 2:])
        whi = (rb - lt).clamp(min=0)
        inter = whi[..., 0] * whi[..., 1]
        aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        return inter / (aa[:, None] + ab[None, :] - inter)

    for b in (0, 7, 15):
        k = torch.from_numpy(keep[b]).to(DEV)
        assert 0 < k.numel() < n
        boxes, score, cls = ct[b, :, :4], ct[b, :, 4], ct[b, :, 6]
        ks = score[k]
        assert bool((ks[:-1] >= ks[1:]).all())
        m = iou(boxes[k], boxes[k])
        same = cls[k][:, None] == cls[k][None, :]
        m = torch.where(same, m, torch.zeros_like(m))
        m.fill_diagonal_(0)
        assert float(m.max()) <= thr + 1e-6
        rej = torch.ones(n, dtype=torch.bool, device=DEV)
        rej[k] = False
        r = rej.nonzero().flatten()
        mr = iou(boxes[r], boxes[k])
        cover = (mr > thr - 1e-6) & (cls[r][:, None] == cls[k][None, :]) & (score[k][None, :] >= score[r][:, None])
        assert bool(cover.any(dim=1).all())
        sub = ct[b:b + 1, k].contiguous()
        again = bbox.nms_keep_indices(sub, torch.tensor([k.numel()], dtype=torch.int32, device=DEV), thr)[0]
        assert again.shape[0] == k.numel() and np.array_equal(again, np.arange(k.numel()))


def test_post_process_estimate_too_small_is_repaired_and_abi_never_overruns():
    """post_process launches NMS with a row estimate from the previous call and verifies it from the counts that come back with
    the results: a too small estimate must cost one more launch, never a wrong answer.  And the C ABI itself: ccnet_nms with
    nmax below an image's count works on the first nmax candidates only (no overrun of the nmax-sized buffers)."""
    from oracle import ccnet_oracle as orc
    from ccnet_amd import ops
    from ccnet_amd.utils import bbox
    g = torch.Generator().manual_seed(48)
    B, A, nc = 4, 2000, 6
    pred = torch.rand(B, A, 5 + nc, generator=g)
    pred[..., 2:4] = pred[..., 2:4] * 0.15 + 0.01
    pred[2, :, 4] *= 0.2
    info = [[i, 320, 240] for i in range(B)]
    ref = orc.post_process(pred, nc, [160, 256], info, 0.3, 0.5)
    d = pred.to(DEV)
    key = (B, A, str(d.device))
    for guess in (None, 1, 64, 4096):   # first call (waits for counts), far too small, too small, larger than A
        bbox._nmax_guess.pop(key, None)
        if guess is not None:
            bbox._nmax_guess[key] = guess
        got = bbox.post_process(d, nc, [160, 256], info, 0.3, 0.5)
        for a, b in zip(got, ref):
            assert a.shape == b.shape and np.array_equal(a, b), guess
        assert 64 <= bbox._nmax_guess[key] <= A
    # direct ABI call with nmax smaller than the counts
    cand = torch.empty(B, A, 7, device=DEV)
    counts = torch.empty(B, dtype=torch.int32, device=DEV)
    ops.filter_candidates(d, nc, 0.3, cand, counts)
    assert int(counts.max()) > 100
    nmax = 100
    hw = torch.ones(B, 2, dtype=torch.int32, device=DEV)
    det, keep_idx, nkeep, _ = bbox._nms(cand, counts, nmax, 0.5, hw)
    torch.cuda.synchronize()
    nk = nkeep.cpu().numpy()
    assert (nk <= nmax).all() and (nk > 0).any()
    small = bbox.nms_keep_indices(cand[:, :nmax].contiguous(), torch.clamp(counts, max=nmax), 0.5)
    ki = keep_idx.cpu().numpy()
    for b in range(B):
        assert np.array_equal(ki[b, :nk[b]], small[b])


def test_conv2d_random_sweep():
    """scripts/conv_fuzz.py: 120 seeded random conv problems (kernel sizes 1/3/5/6/7, strides, one and two segments with an
    upsampled first one, residuals, every activation, both dtypes, odd batch sizes) against torch -- so that each dispatch
    branch of ccnet_conv2d (halo-patch, LDS-DMA implicit GEMM, flat 1x1, generic fallback) sees shapes its listed cases do not."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "conv_fuzz.py")
    spec = importlib.util.spec_from_file_location("conv_fuzz", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, worst = mod.run(120, 7, verbose=False)
    assert bad == 0 and len(worst) >= 8, worst


def test_nms_single_block_path_equals_tiled_path():
    """ccnet_nms runs the whole NMS of an image in one block when the caller provides <= 256 rows per image and the tiled
    five-kernel pipeline otherwise; both use the same device arithmetic and must agree bit for bit (kept indices, order, rows),
    and with the oracle.  Cases: dense overlaps, score ties, an empty image, exactly 256 and 64-multiples of candidates."""
    from oracle import ccnet_oracle as orc
    from ccnet_amd.utils import bbox
    g = torch.Generator().manual_seed(5)
    for n_per in ([0, 1, 63, 64, 65, 200, 256, 255], [128, 128, 17, 192]):
        B, A = len(n_per), 600
        cand = torch.zeros(B, A, 7)
        for b, n in enumerate(n_per):
            c = torch.rand(n, 2, generator=g) * 0.6 + 0.2
            wh = torch.rand(n, 2, generator=g) * 0.25 + 0.02
            cand[b, :n, 0:2] = c - wh / 2
            cand[b, :n, 2:4] = c + wh / 2
            cand[b, :n, 4] = (torch.rand(n, generator=g) * 8).floor() / 8 + 0.05   # many exact score ties
            cand[b, :n, 5] = 1.0
            cand[b, :n, 6] = torch.randint(0, 3, (n,), generator=g).float()
        counts = torch.tensor(n_per, dtype=torch.int32)
        hw = torch.tensor([[480, 640]] * B, dtype=torch.int32).to(DEV)
        cd, ct = cand.to(DEV), counts.to(DEV)
        res = {}
        for nmax in (max(max(n_per), 1), 600):   # <= 256: single block; 600: tiled
            det, keep_idx, nkeep, _ = bbox._nms(cd, ct, nmax, 0.5, hw)
            torch.cuda.synchronize()
            nk = nkeep.cpu().numpy()
            res[nmax] = [(keep_idx[b, :nk[b]].cpu().numpy(), det[b, :nk[b]].cpu().numpy()) for b in range(B)]
        small, tiled = res[max(max(n_per), 1)], res[600]
        for b, n in enumerate(n_per):
            assert np.array_equal(small[b][0], tiled[b][0]) and np.array_equal(small[b][1], tiled[b][1]), (n_per, b)
            ref = orc.batched_nms_reference(cand[b, :n, :4].numpy(), (cand[b, :n, 4] * cand[b, :n, 5]).numpy(), cand[b, :n, 6].numpy(), 0.5)
            assert np.array_equal(small[b][0], np.asarray(ref, dtype=small[b][0].dtype)), (n_per, b)


def _nms_case(rng, n, ncls, lo, hi, size, cls_vals=None, ties=False):
    """(n,7) candidates: centres uniform in [lo, hi], log-normal sizes around `size`, classes 0..ncls-1 (or drawn from cls_vals)."""
    cand = np.zeros((n, 7), np.float32)
    c = rng.uniform(lo, hi, (n, 2))
    wh = np.exp(rng.normal(math.log(size), 0.6, (n, 2)))
    cand[:, 0:2] = c - wh / 2
    cand[:, 2:4] = c + wh / 2
    cand[:, 4] = np.floor(rng.uniform(0, 1, n) * 16) / 16 + 0.03 if ties else rng.uniform(0, 1, n)
    cand[:, 5] = rng.uniform(0.5, 1, n) if not ties else 1.0
    cand[:, 6] = rng.randint(0, ncls, n) if cls_vals is None else np.asarray(cls_vals, np.float32)[rng.randint(0, len(cls_vals), n)]
    return cand


def test_nms_class_groups_bit_exact():
    """The tiled NMS works per class GROUP (csrc/detect.hip): candidates bucketed by class, adjacent classes merged when their
    offset boxes are not provably disjoint, rank / bitmask / sweep per group, kept lists merged by score.  It must stay bit-identical
    to the global coordinate-trick algorithm of torchvision 0.9.0 (oracle.batched_nms_reference) in every regime:
    separable classes; boxes hanging far outside the image so that the offset boxes of neighbouring classes OVERLAP and suppress
    each other (the reference's quirk); one class only (all four mask panels); 80 classes (more groups than sweep blocks);
    class ids the partition does not handle (>= 256, fractional); a negative threshold (a zero overlap suppresses); exact score
    ties across classes; empty and one-box images beside large ones."""
    from oracle import ccnet_oracle as orc
    from ccnet_amd.utils import bbox
    rng = np.random.RandomState(11)
    cases = [
        ("separable", [_nms_case(rng, 2500, 6, 0, 640, 40), _nms_case(rng, 1900, 6, 0, 640, 60, ties=True)], 0.6),
        ("overhang", [_nms_case(rng, 2000, 4, -700, 700, 500), _nms_case(rng, 1500, 6, -100, 640, 200)], 0.5),
        ("one_class", [_nms_case(rng, 3000, 1, 0, 640, 30), _nms_case(rng, 700, 1, 0, 64, 30)], 0.6),
        ("80_classes", [_nms_case(rng, 4000, 80, 0, 640, 120), np.zeros((0, 7), np.float32), _nms_case(rng, 1, 80, 0, 640, 50)], 0.45),
        ("odd_class_ids", [_nms_case(rng, 1200, 0, 0, 640, 80, cls_vals=[0, 3, 255, 256, 300]), _nms_case(rng, 900, 0, 0, 640, 80, cls_vals=[1, 2.5, 4])], 0.5),
        ("negative_threshold", [_nms_case(rng, 600, 3, 0, 640, 20)], -0.1),
        ("sparse_classes", [_nms_case(rng, 2200, 0, -3, 643, 50, cls_vals=[2, 7, 8, 200], ties=True)], 0.6),
    ]
    # a hand-made cross-class suppression: class 0 [100,100,700,700] and class 1 [-501,-501,99,99] meet after the offset of 701
    quirk = np.zeros((300, 7), np.float32)
    quirk[:, :] = _nms_case(rng, 300, 2, 100, 600, 30)
    quirk[0] = [100, 100, 700, 700, 0.99, 1.0, 0]
    quirk[1] = [-501, -501, 99, 99, 0.98, 1.0, 1]
    cases.append(("quirk", [quirk], 0.5))
    for name, imgs, thr in cases:
        B, A = len(imgs), max(max(im.shape[0] for im in imgs), 300)
        cand = np.zeros((B, A, 7), np.float32)
        for b, im in enumerate(imgs):
            cand[b, :im.shape[0]] = im
        counts = [im.shape[0] for im in imgs]
        keep = bbox.nms_keep_indices(torch.from_numpy(cand).to(DEV), torch.tensor(counts, dtype=torch.int32, device=DEV), thr)
        for b, im in enumerate(imgs):
            ref = orc.batched_nms_reference(im[:, :4], im[:, 4] * im[:, 5], im[:, 6], thr)
            assert np.array_equal(np.asarray(keep[b], np.int64), np.asarray(ref, np.int64)), (name, b, len(keep[b]), len(ref))
        if name == "quirk":
            assert 1 not in list(keep[0])   # suppressed ACROSS classes, as in the reference


# ------------------------------------------------------------------------------------------ fused CSP layer
def _csp_state(prefix, cin, h, gen):
    """state_dict of a CSPLayer(cin, 2h, n=1) with non-trivial BN statistics (reference key names, model_block.py:86-116)."""
    sd = {}

    def base(p, co, ci, k):
        sd[p + ".conv.weight"] = torch.randn(co, ci, k, k, generator=gen) * (1.5 / math.sqrt(ci * k * k))
        sd[p + ".bn.weight"] = torch.rand(co, generator=gen) * 0.5 + 0.75
        sd[p + ".bn.bias"] = torch.randn(co, generator=gen) * 0.2
        sd[p + ".bn.running_mean"] = torch.randn(co, generator=gen) * 0.1
        sd[p + ".bn.running_var"] = torch.rand(co, generator=gen) * 0.5 + 0.5

    base(prefix + ".conv1", h, cin, 1)
    base(prefix + ".conv2", h, cin, 1)
    base(prefix + ".conv3", 2 * h, 2 * h, 1)
    base(prefix + ".m.0.conv1", h, h, 1)
    base(prefix + ".m.0.conv2", h, h, 3)
    return sd


@pytest.mark.parametrize("h,segs,up,shortcut,B,H,W", [
    (32, (64,), False, True, 3, 16, 32),       # dark2.1 of network/net.py
    (64, (128,), False, True, 2, 24, 48),      # dark3.1
    (64, (128, 128), True, False, 2, 16, 32),  # C3_p3: cat(upsample(fpn1), gate2)
    (32, (64, 64), True, False, 1, 8, 16),     # one tile: every halo pixel is outside the image
    (64, (64, 192), False, False, 1, 40, 16),
])
def test_csp_fused_matches_layerwise_and_torch(h, segs, up, shortcut, B, H, W):
    """ccnet_csp_fused (one launch per CSPLayer, csrc/csp_fused.hip) against (a) the same layer as five ccnet_conv2d launches --
    same bf16 rounding points, so only the accumulation order differs -- and (b) CSPLayer.forward in torch fp32
    (model_block.py:86-116, 193-218) on the bf16-rounded input, at bf16 tolerance."""
    import os
    import torch.nn.functional as F
    from ccnet_amd import ops
    from ccnet_amd._lib import BF16
    from ccnet_amd.network import graph
    gen = torch.Generator().manual_seed(h * 1000 + sum(segs) + H)
    cin = sum(segs)
    sd = _csp_state("blk", cin, h, gen)
    xs = []
    for i, c in enumerate(segs):
        hh, ww = (H // 2, W // 2) if (up and i == 0) else (H, W)
        xs.append((torch.randn(B, hh, ww, c, generator=gen) * 0.8).to(torch.bfloat16))

    def run(fused):
        os.environ["CCNET_NO_FUSED_CSP"] = "0" if fused else "1"
        os.environ["CCNET_FUSED_CSP_ALL"] = "1"   # hidden 64 too (the networks only send hidden 32 to the fused kernel)
        try:
            comp = graph.Compiler(sd, {}, B, H, W, BF16, torch.device(DEV), "silu")
            ins = []
            for i, x in enumerate(xs):
                t = graph.T(x.to(DEV).contiguous(), 0, x.shape[-1], x.shape[1], x.shape[2])
                ins.append(t.up() if (up and i == 0) else t)
            n0 = []
            orig = ops.csp_fused
            ops.csp_fused = lambda *a, **k: (n0.append(1), orig(*a, **k))[1]
            try:
                out = comp.csp("blk", ins, 1, shortcut)
            finally:
                ops.csp_fused = orig
            torch.cuda.synchronize()
            assert bool(n0) == fused, "fused kernel %s" % ("not taken" if fused else "taken")
            return out.buf[..., out.coff:out.coff + out.C].float().cpu()
        finally:
            os.environ.pop("CCNET_NO_FUSED_CSP", None)
            os.environ.pop("CCNET_FUSED_CSP_ALL", None)

    got, layer = run(True), run(False)
    # torch fp32 reference
    def bc(p, x, k):
        w, b = ops.fold_bn(sd[p + ".conv.weight"], (sd[p + ".bn.weight"], sd[p + ".bn.bias"], sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"]))
        return F.silu(F.conv2d(x, w.float(), b.float(), padding=(k - 1) // 2))
    parts = []
    for i, x in enumerate(xs):
        xf = x.float().permute(0, 3, 1, 2)
        parts.append(F.interpolate(xf, scale_factor=2, mode="nearest") if (up and i == 0) else xf)
    x = torch.cat(parts, 1)
    x1, x2 = bc("blk.conv1", x, 1), bc("blk.conv2", x, 1)
    y = bc("blk.m.0.conv2", bc("blk.m.0.conv1", x1, 1), 3)
    x1 = y + x1 if shortcut else y
    ref = bc("blk.conv3", torch.cat([x1, x2], 1), 1).permute(0, 2, 3, 1)
    rms = lambda a, b: float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    assert got.shape == ref.shape
    assert rms(got, layer) < 4e-3 and float((got - layer).abs().max()) < 0.06, (rms(got, layer), float((got - layer).abs().max()))
    assert rms(got, ref) < 1.5e-2 and rms(layer, ref) < 1.5e-2, (rms(got, ref), rms(layer, ref))


def test_transformer_fused_is_run_to_run_identical():
    """The fused Transformer at the headline shape (S = 64, N = 400) must give bit-identical outputs on repeated launches, for
    the decoder alone (cross-attention stages the encoder memory by LDS-DMA) and for the full 6 + 6 stack.  A build whose bf16
    packing used the vector conversion failed exactly this (tokens 48..63, csrc/common.h:pack_bf16x2)."""
    import importlib
    from ccnet_amd import ops, synth
    net = importlib.import_module("ccnet_amd.network.net").Net(0.33, 0.5, 6, "silu")
    sd = synth.synth_state_dict(net.state_dict(), seed=0, bn_calib="net")
    S, N = 64, 400
    g = torch.Generator().manual_seed(3)
    src = (torch.randn(S * N, 256, generator=g) * 0.7).to(DEV, torch.bfloat16)
    for ne, nd in ((0, 1), (6, 6)):
        wst, pvec, fpw = ops.pack_transformer(sd, "dark5.2.transformer", ne, nd, DEV)
        outs = []
        for _ in range(4):
            out = torch.full_like(src, 7.0)
            ops.transformer_fused(src, 256, out, 256, S, N, wst, fpw, pvec, ne, nd)
            torch.cuda.synchronize()
            outs.append(out)
        for o in outs[1:]:
            assert torch.equal(outs[0], o), "enc %d dec %d: %d elements differ between launches" % (ne, nd, int((outs[0] != o).sum()))


def test_filter_many_anchors_multi_block_equals_single_block():
    """ccnet_filter_ws (one workgroup per 32768 anchors after a counting pass, for maps like the 136 000 anchors of a 1280^2
    input) against ccnet_filter (one workgroup per image): same rows in the same order, same counts; images with no, few, many
    and all anchors kept, and an anchor count that is not a multiple of anything."""
    from ccnet_amd import _lib
    L = _lib.lib()
    B, A, nc = 5, 70001, 6
    g = torch.Generator().manual_seed(9)
    pred = torch.rand(B, A, 5 + nc, generator=g)
    pred[0, :, 4] = 0.0           # nothing kept
    pred[1, :, 4] *= 0.4          # few
    pred[3, :, 4:] = 1.0          # everything kept
    pred[4, 40000:, 4] = 0.0      # only the first two super-chunks contribute
    pred = pred.to(DEV).contiguous()
    res = []
    for use_ws in (False, True):
        cand = torch.full((B, A, 7), -1.0, device=DEV)
        counts = torch.full((B,), -1, dtype=torch.int32, device=DEV)
        if use_ws:
            n = int(L.ccnet_filter_workspace_ints(B, A))
            assert n == B * 3
            ws = torch.empty(n, dtype=torch.int32, device=DEV)
            _lib.check(L.ccnet_filter_ws(pred.data_ptr(), B, A, nc, 0.3, cand.data_ptr(), counts.data_ptr(), ws.data_ptr(), _lib.stream_ptr()))
        else:
            _lib.check(L.ccnet_filter(pred.data_ptr(), B, A, nc, 0.3, cand.data_ptr(), counts.data_ptr(), _lib.stream_ptr()))
        torch.cuda.synchronize()
        res.append((cand.cpu(), counts.cpu()))
    (c0, n0), (c1, n1) = res
    assert torch.equal(n0, n1) and int(n0[0]) == 0 and int(n0[3]) == A and 0 < int(n0[1]) < int(n0[2])
    for b in range(B):
        assert torch.equal(c0[b, :n0[b]], c1[b, :n1[b]]), b
    assert int(L.ccnet_filter_workspace_ints(64, 8400)) == 0


def test_csp_fused_is_run_to_run_identical():
    """Repeated launches of the fused CSP kernel at a network-sized problem (many tiles per persistent workgroup) must agree bit
    for bit -- hidden 32 with the weights staged in LDS (a missing barrier after that staging once made this fail), hidden 64
    with the weights streamed."""
    import os
    from ccnet_amd import ops
    from ccnet_amd._lib import BF16
    from ccnet_amd.network import graph
    os.environ["CCNET_FUSED_CSP_ALL"] = "1"
    try:
        for h, cin, B, H, W in ((32, 64, 16, 160, 160), (64, 128, 16, 80, 80)):
            gen = torch.Generator().manual_seed(h)
            sd = _csp_state("blk", cin, h, gen)
            comp = graph.Compiler(sd, {}, B, H, W, BF16, torch.device(DEV), "silu")
            x = graph.T((torch.randn(B, H, W, cin, generator=gen) * 0.8).to(DEV, torch.bfloat16), 0, cin, H, W)
            with ops.Recorder() as rec:
                out = comp.csp("blk", [x], 1, True)
            plan = ops.Plan(rec.handle, comp.keep)
            ref = None
            for _ in range(6):
                out.buf.fill_(3.0)
                plan.run()
                torch.cuda.synchronize()
                cur = out.buf.clone()
                assert ref is None or torch.equal(ref, cur), "hidden %d: %d elements differ" % (h, int((ref != cur).sum()))
                ref = cur
    finally:
        os.environ.pop("CCNET_FUSED_CSP_ALL", None)
