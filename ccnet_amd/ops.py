"""Host-side operator layer: weight folding/packing (fp64 on the CPU) and thin wrappers that hand NHWC device
buffers to the C ABI.  Every wrapper either launches on torch's current stream or, inside `Recorder`, appends to
a plan.  No compute happens in Python/ATen here."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import ACT, BF16, F32, ConvDesc, Slice, check, lib, null_slice, sl, stream_ptr

BN_EPS = 1e-5


# --------------------------------------------------------------------------------------------- weight packing
def fold_bn(w, bn=None, conv_bias=None, eps=BN_EPS):
    """(weight OIHW any float, bn = (gamma, beta, mean, var) or None, conv bias or None) -> (w fp64, b fp64).
    BaseConv = conv(no bias) -> BN(eval) (network/model_block.py:52-58)."""
    w = w.detach().to("cpu", torch.float64)
    cout = w.shape[0]
    b = torch.zeros(cout, dtype=torch.float64) if conv_bias is None else conv_bias.detach().to("cpu", torch.float64).clone()
    if bn is not None:
        g, beta, mean, var = [t.detach().to("cpu", torch.float64) for t in bn]
        s = g / torch.sqrt(var + eps)
        w = w * s.view(-1, *([1] * (w.dim() - 1)))
        b = (b - mean) * s + beta
    return w, b


def s2d_to_6x6(w3):
    """SpaceToDepth (TL,TR,BL,BR concat, model_block.py:169-173) + k3 s1 p1 conv == k6 s2 p2 conv on the original
    tensor with W6[o, c, 2u+dy, 2v+dx] = W3[o, (2dy+dx)*C + c, u, v]."""
    co, c4, kh, kw = w3.shape
    assert kh == 3 and kw == 3 and c4 % 4 == 0
    c = c4 // 4
    w6 = torch.zeros(co, c, 6, 6, dtype=w3.dtype)
    for dy in range(2):
        for dx in range(2):
            w6[:, :, dy::2, dx::2] = w3[:, (2 * dy + dx) * c:(2 * dy + dx + 1) * c]
    return w6


class PackedConv:
    """Device-resident packed weights of one conv: [Cout_pad][Kpad] in `dtype`, fp32 bias."""

    def __init__(self, w, b, dtype, device):
        w = w.to(torch.float64)
        cout, cin, kh, kw = w.shape
        self.cout, self.cin, self.kh, self.kw, self.dtype = cout, cin, kh, kw, dtype
        k = kh * kw * cin
        self.cout_pad, self.kpad = _lib.conv_pack_dims(dtype, cout, k)
        packed = torch.zeros(self.cout_pad, self.kpad, dtype=torch.float64)
        packed[:cout, :k] = w.permute(0, 2, 3, 1).reshape(cout, k)
        self.weight = packed.to(_lib.torch_dtype(dtype)).to(device).contiguous()
        self.bias = None if b is None else b.to(torch.float32).to(device).contiguous()


def pack_stem(w6, b, device):
    """[Cout][3][6][6] -> fp32 [Cout][160], k = (c*6 + ky)*8 + kx (include/ccnet_hip.h: ccnet_stem_conv)."""
    cout = w6.shape[0]
    assert tuple(w6.shape[1:]) == (3, 6, 6)
    p = torch.zeros(cout, 20, 8, dtype=torch.float64)
    p[:, :18, :6] = w6.to(torch.float64).reshape(cout, 18, 6)
    return p.reshape(cout, 160).to(torch.float32).to(device).contiguous(), b.to(torch.float32).to(device).contiguous()


# --------------------------------------------------------------------------------------------- op wrappers
def conv2d(ins, pc, B, H, W, stride, pad, out, act="none", residual=None, out_dtype=None):
    """ins: list of 1-2 Slice (channel concat); out/residual: Slice.  H, W: logical input size."""
    d = ConvDesc()
    d.dtype = pc.dtype
    d.out_dtype = pc.dtype if out_dtype is None else out_dtype
    d.B, d.H, d.W = B, H, W
    d.kh, d.kw, d.stride, d.pad = pc.kh, pc.kw, stride, pad
    d.Ho = (H + 2 * pad - pc.kh) // stride + 1
    d.Wo = (W + 2 * pad - pc.kw) // stride + 1
    d.Cout = pc.cout
    d.nseg = len(ins)
    for i, s in enumerate(ins):
        d.inp[i] = s
    if sum(s.C for s in ins) != pc.cin:
        raise ValueError("conv2d: input slices carry %d channels, weights expect %d" % (sum(s.C for s in ins), pc.cin))
    d.weight = pc.weight.data_ptr()
    d.Kpad, d.Cout_pad = pc.kpad, pc.cout_pad
    d.bias = None if pc.bias is None else pc.bias.data_ptr()
    d.act = ACT[act]
    d.residual = null_slice() if residual is None else residual
    d.out = out
    check(lib().ccnet_conv2d(C.byref(d), stream_ptr()))
    return d.Ho, d.Wo


HALO_RING = 9  # conv_halo.hip: HRING


class PackedHalo:
    """Register-stream packing of a 3x3 conv for ccnet_conv_halo (include/ccnet_hip.h): w (Cout, Cin, 3, 3) fp64 over the
    channels the conv sees (SpaceToDepth: the 4C virtual channels, model_block.py:169-173), Cin % 32 == 0, Cout % 64 == 0."""

    def __init__(self, w, b, device):
        w = w.to(torch.float64)
        cout, cin, kh, kw = w.shape
        assert (kh, kw) == (3, 3) and cin % 32 == 0 and cout % 64 == 0
        self.cout, self.cin = cout, cin
        # [tap][n64][j][row][cc][kq][e] -> [n64][cc][tap][j][kq][row][e]: per 64 output channels, chunk-major, tap, n-tile; a fragment
        # is [kq (lane >> 4)][row (lane & 15)][8]
        f = w.permute(2, 3, 0, 1).reshape(9, cout // 64, 4, 16, cin // 32, 4, 8).permute(1, 4, 0, 2, 5, 3, 6).reshape(cout // 64, -1, 512)
        f = torch.cat([f, f[:, :HALO_RING]], 1)  # the ring prefetches across the tile boundary: the stream's head again
        assert f.numel() == int(lib().ccnet_conv_halo_stream_elems(cin, cout))
        self.weight = f.to(torch.bfloat16).to(device).contiguous()
        self.bias = None if b is None else b.to(torch.float32).to(device).contiguous()


# pass vq over the staged chunks -> the two 3x3 taps (ky, kx) it multiplies (None: four zero fragments); passes 0..2 read the
# chunks of phase (dy, dx) = (0,0), (0,1), (1,0), passes 3 and 4 both those of phase (1,1)
HALO_S2_PASSES = [((1, 1), None), ((1, 0), (1, 2)), ((0, 1), (2, 1)), ((0, 0), (0, 2)), ((2, 0), (2, 2))]


class PackedHaloS2:
    """Register-stream packing of a 3x3 STRIDE-2 pad-1 conv for ccnet_conv_halo mode 2 (include/ccnet_hip.h): w (Cout, C, 3, 3),
    C % 32 == 0, Cout % 64 == 0.  Per 64 output channels: for each pass (HALO_S2_PASSES), for each 32-channel chunk of C, two taps of
    four n-tile fragments each."""

    def __init__(self, w, b, device):
        w = w.to(torch.float64)
        cout, cin, kh, kw = w.shape
        assert (kh, kw) == (3, 3) and cin % 32 == 0 and cout % 64 == 0
        self.cout, self.cin = cout, cin
        w8 = w.reshape(cout // 64, 4, 16, cin // 32, 4, 8, 3, 3)   # [n64][j][row][cc][kq][e][ky][kx]
        zero = torch.zeros(cout // 64, 4, 512, dtype=torch.float64)
        seq = []
        for taps in HALO_S2_PASSES:
            for cc in range(cin // 32):
                for tap in taps:
                    # [n64][j][row][kq][e] -> [n64][j][kq][row][e]: a fragment is [kq (lane >> 4)][row (lane & 15)][8]
                    seq.append(zero if tap is None else w8[:, :, :, cc, :, :, tap[0], tap[1]].permute(0, 1, 3, 2, 4).reshape(cout // 64, 4, 512))
        f = torch.cat(seq, 1)
        f = torch.cat([f, f[:, :8]], 1)  # the ring prefetches across the tile boundary: the stream's head again
        assert f.numel() == int(lib().ccnet_conv_halo_s2_stream_elems(cin, cout))
        self.weight = f.to(torch.bfloat16).to(device).contiguous()
        self.bias = None if b is None else b.to(torch.float32).to(device).contiguous()


def conv_halo_ok(dtype, out_dtype, ins, cin_virtual, cout, out, residual=None):
    """Shapes ccnet_conv_halo takes (the C entry point validates again)."""
    return (dtype == BF16 and (out_dtype is None or out_dtype == BF16) and cout % 64 == 0 and cin_virtual % 32 == 0
            and all(s.shift == 0 and s.C % 32 == 0 and s.ldc % 8 == 0 and s.coff % 8 == 0 for s in ins)
            and out.ldc % 4 == 0 and out.coff % 4 == 0 and (residual is None or (residual.ldc % 4 == 0 and residual.coff % 4 == 0)))


def conv_halo(ins, pk, s2d, B, H, W, out, act="none", residual=None):
    """s2d: False / 0 = 3x3 s1 p1, True / 1 = SpaceToDepth + 3x3 (PackedHalo over the 4C virtual channels), 2 = 3x3 s2 p1 (PackedHaloS2)."""
    arr = (Slice * len(ins))(*ins)
    check(lib().ccnet_conv_halo(arr, len(ins), int(s2d), B, H, W, pk.cout, pk.weight.data_ptr(),
                                None if pk.bias is None else pk.bias.data_ptr(), ACT[act], null_slice() if residual is None else residual,
                                out, stream_ptr()))


PW_RING = 8


class PackedPw:
    """Register-stream packing of a 1x1 conv for ccnet_conv_pw (include/ccnet_hip.h): w (Cout, Cin[, 1, 1]), Cin % 64 == 0,
    Cout % 64 == 0."""

    def __init__(self, w, b, device):
        w = w.to(torch.float64).reshape(w.shape[0], -1)
        cout, cin = w.shape
        assert cin % 64 == 0 and cout % 64 == 0
        self.cout, self.cin = cout, cin
        # [n64][j][row][ks][kq][e] -> [n64][ks][j][kq][row][e]: per 64 output channels, k-step major, n-tile; a fragment is
        # [kq (lane >> 4)][row (lane & 15)][8]
        f = w.reshape(cout // 64, 4, 16, cin // 32, 4, 8).permute(0, 3, 1, 4, 2, 5).reshape(cout // 64, -1, 512)
        f = torch.cat([f, f[:, :PW_RING]], 1)  # the ring prefetches across the tile boundary: the stream's head again
        assert f.numel() == int(lib().ccnet_conv_pw_stream_elems(cin, cout))
        self.weight = f.to(torch.bfloat16).to(device).contiguous()
        self.bias = None if b is None else b.to(torch.float32).to(device).contiguous()


def conv_pw_ok(dtype, out_dtype, ins, cout, out, residual=None):
    """Shapes ccnet_conv_pw takes (the C entry point validates again)."""
    return (dtype == BF16 and (out_dtype is None or out_dtype == BF16) and cout % 64 == 0 and len(ins) in (1, 2)
            and all(s.shift <= 2 and s.C % 64 == 0 and s.ldc % 8 == 0 and s.coff % 8 == 0 for s in ins)
            and out.ldc % 4 == 0 and out.coff % 4 == 0 and (residual is None or (residual.ldc % 4 == 0 and residual.coff % 4 == 0)))


def conv_pw(ins, pk, B, H, W, out, act="none", residual=None):
    arr = (Slice * len(ins))(*ins)
    check(lib().ccnet_conv_pw(arr, len(ins), B, H, W, pk.cout, pk.weight.data_ptr(), None if pk.bias is None else pk.bias.data_ptr(),
                              ACT[act], null_slice() if residual is None else residual, out, stream_ptr()))


BNECK_RING = {64: 8, 128: 16, 256: 16}  # bneck_fused.hip: bneck_ring(C)


class PackedBneck:
    """Register-stream packing of one Bottleneck (1x1 C->C, 3x3 C->C, C = 64, 128 or 256) for ccnet_bneck_fused (include/ccnet_hip.h):
    w1 (C, C[, 1, 1]), w2 (C, C, 3, 3) fp64, BN folded."""

    def __init__(self, w1, b1, w2, b2, device):
        w1 = w1.to(torch.float64).reshape(w1.shape[0], -1)
        w2 = w2.to(torch.float64)
        c = w1.shape[0]
        assert c in (64, 128, 256) and tuple(w1.shape) == (c, c) and tuple(w2.shape) == (c, c, 3, 3)
        self.c = c
        # conv1: [nt][row][ks][kq][e] -> [ks][nt][kq][row][e]; a fragment is [kq (lane >> 4)][row (lane & 15)][8]
        f1 = w1.reshape(c // 16, 16, c // 32, 4, 8).permute(2, 0, 3, 1, 4).reshape(1, -1, 512).expand(c // 64, -1, -1)
        # conv2: PackedHalo's order per 64 output channels: chunk, tap, n-tile
        f2 = w2.permute(2, 3, 0, 1).reshape(9, c // 64, 4, 16, c // 32, 4, 8).permute(1, 4, 0, 2, 5, 3, 6).reshape(c // 64, -1, 512)
        f = torch.cat([f1, f2], 1)
        f = torch.cat([f, f[:, :BNECK_RING[c]]], 1)  # the ring prefetches across the tile boundary: the stream's head again
        assert f.numel() == int(lib().ccnet_bneck_fused_stream_elems(c))
        self.weight = f.to(torch.bfloat16).to(device).contiguous()
        self.bias1 = b1.to(torch.float32).to(device).contiguous()
        self.bias2 = b2.to(torch.float32).to(device).contiguous()


def bneck_fused(x, pk, B, H, W, shortcut, act, out):
    check(lib().ccnet_bneck_fused(x, B, H, W, pk.weight.data_ptr(), pk.bias1.data_ptr(), pk.bias2.data_ptr(), 1 if shortcut else 0,
                                  ACT[act], out, stream_ptr()))


class PackedCsp:
    """Fragment-packed weights of one CSPLayer with a single bottleneck for ccnet_csp_fused (include/ccnet_hip.h):
    mats = [(w OIHW fp64, b fp64)] for [conv1;conv2], bottleneck conv1, bottleneck conv2 (3x3), conv3."""

    def __init__(self, w12, b12, wb1, bb1, wb2, bb2, w3, b3, device):
        h = wb1.shape[0]
        self.hidden, self.cin = h, w12.shape[1]
        mats = [w12.reshape(2 * h, -1), wb1.reshape(h, h), wb2.permute(0, 2, 3, 1).reshape(h, 9 * h), w3.reshape(w3.shape[0], 2 * h)]
        assert w12.shape[0] == 2 * h and w3.shape[0] == 2 * h and tuple(wb2.shape) == (h, h, 3, 3)
        frags = torch.cat([_mfma_frags(m.to(torch.float64)).reshape(-1) for m in mats])
        assert frags.numel() == int(lib().ccnet_csp_fused_weight_elems(self.cin, h))
        self.weight = frags.to(torch.bfloat16).to(device).contiguous()
        self.bias = torch.cat([b12, bb1, bb2, b3]).to(torch.float32).to(device).contiguous()


def csp_fused(ins, pk, B, H, W, shortcut, act, out):
    arr = (Slice * len(ins))(*ins)
    check(lib().ccnet_csp_fused(arr, len(ins), B, H, W, pk.hidden, 1 if shortcut else 0, ACT[act], pk.weight.data_ptr(), pk.bias.data_ptr(),
                                out, stream_ptr()))


def stem_conv(x, w, b, cout, act, out, out_dtype):
    B, _, H, W = x.shape
    check(lib().ccnet_stem_conv(x.data_ptr(), B, H, W, w.data_ptr(), b.data_ptr(), cout, ACT[act], out, out_dtype, stream_ptr()))


def dwconv3x3(inp, dtype, B, H, W, w, b, out):
    check(lib().ccnet_dwconv3x3(inp, dtype, B, H, W, w.data_ptr(), b.data_ptr(), out, stream_ptr()))


def sppf_pool(buf, dtype, B, H, W):
    check(lib().ccnet_sppf_pool(buf, dtype, B, H, W, stream_ptr()))


def maxpool3x3s2(inp, dtype, B, H, W, out):
    check(lib().ccnet_maxpool3x3s2(inp, dtype, B, H, W, out, stream_ptr()))


def gate_scratch_floats(B, H, W, Cc):
    return int(lib().ccnet_gate_scratch_floats(B, H, W, Cc))


def cbam(inp, dtype, B, H, W, fc1, fc2, w7, scratch, out):
    cr = fc1.shape[0]
    k = w7.shape[-1]
    check(lib().ccnet_cbam(inp, dtype, B, H, W, fc1.data_ptr(), fc2.data_ptr(), cr, w7.data_ptr(), k, scratch.data_ptr(), out, stream_ptr()))


def eca(inp, dtype, B, H, W, w3, scratch, out):
    check(lib().ccnet_eca(inp, dtype, B, H, W, w3.data_ptr(), scratch.data_ptr(), out, stream_ptr()))


def asff_fuse(l0, l1, l2, logits, dtype, B, H, W, out):
    check(lib().ccnet_asff_fuse(l0, l1, l2, logits, dtype, B, H, W, out, stream_ptr()))


def layernorm(x, res, gamma, beta, out, dtype, rows, E, eps=1e-5):
    check(lib().ccnet_layernorm(x.data_ptr(), None if res is None else res.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                out.data_ptr(), dtype, rows, E, eps, stream_ptr()))


def attention(q, k, v, dtype, Sq, Sk, N, heads, hd, out):
    check(lib().ccnet_attention(q, k, v, dtype, Sq, Sk, N, heads, hd, out, stream_ptr()))


def _mfma_frags(w):
    """(N, K) matrix -> [N/16][K/32][512] fragments: the 64-lane A operand of v_mfma_f32_16x16x32_bf16
    (lane l holds row l & 15, k = (l >> 4) * 8 .. + 7)."""
    n, k = w.shape
    return w.reshape(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(n // 16, k // 32, 512)


def pack_transformer(sd, prefix, n_enc, n_dec, device):
    """Weight stream and parameter vector of ccnet_transformer_fused for the nn.Transformer under `prefix`
    (d_model 256, 2 heads, FF 1024).  The order below IS the kernel's program order (transformer_fused.hip), for each of
    its 4 waves: per attention block, per head, the Q/K/V projections (wave w owns 32 columns of each), then out_proj (wave w
    owns 64 output columns); per FFN, eight 128-wide hidden
    chunks, linear1 chunk (wave w owns 32 of its columns) then the matching K slice of linear2.  Inside a GEMM call fragments run k-tile major, n-tile minor."""
    E, FF, NW, RING = 256, 1024, 4, 16
    f64 = lambda t: t.detach().to(torch.float64).cpu()

    def attn(q):
        fin = _mfma_frags(f64(sd[q + ".in_proj_weight"]))
        fout = _mfma_frags(f64(sd[q + ".out_proj.weight"]))
        per_wave = []
        for w in range(NW):
            seq = []
            for h in range(2):
                for part in range(3):
                    nt = (part * E + h * 128 + w * 32) // 16
                    for kt in range(8):
                        for j in range(2):
                            seq.append(fin[nt + j, kt])
            for kt in range(8):
                for j in range(4):
                    seq.append(fout[w * 4 + j, kt])
            per_wave.append(torch.stack(seq))
        return per_wave, [f64(sd[q + ".in_proj_bias"]), f64(sd[q + ".out_proj.bias"])]

    def ffn(q):
        f1 = _mfma_frags(f64(sd[q + ".linear1.weight"]))
        f2 = _mfma_frags(f64(sd[q + ".linear2.weight"]))
        per_wave = []
        for w in range(NW):
            seq = []
            for c in range(FF // 128):
                for kt in range(8):
                    for j in range(2):
                        seq.append(f1[c * 8 + w * 2 + j, kt])
                for kt in range(4):
                    for j in range(4):
                        seq.append(f2[w * 4 + j, c * 4 + kt])
            per_wave.append(torch.stack(seq))
        return per_wave, [f64(sd[q + ".linear1.bias"]), f64(sd[q + ".linear2.bias"])]

    def ln(q):
        return [f64(sd[q + ".weight"]), f64(sd[q + ".bias"])]

    streams = [[] for _ in range(NW)]
    params = []

    def emit(block):
        pw, pr = block
        for w in range(NW):
            streams[w].append(pw[w])
        params.extend(pr)

    for i in range(n_enc):
        q = "%s.encoder.layers.%d" % (prefix, i)
        emit(attn(q + ".self_attn")); params.extend(ln(q + ".norm1"))
        emit(ffn(q)); params.extend(ln(q + ".norm2"))
    params.extend(ln(prefix + ".encoder.norm"))
    for i in range(n_dec):
        q = "%s.decoder.layers.%d" % (prefix, i)
        emit(attn(q + ".self_attn")); params.extend(ln(q + ".norm1"))
        emit(attn(q + ".multihead_attn")); params.extend(ln(q + ".norm2"))
        emit(ffn(q)); params.extend(ln(q + ".norm3"))
    params.extend(ln(prefix + ".decoder.norm"))
    fpw, npar = C.c_longlong(), C.c_longlong()
    check(lib().ccnet_transformer_fused_frags(n_enc, n_dec, C.byref(fpw), C.byref(npar)))
    per_wave = [torch.cat(s) if s else torch.zeros(0, 512, dtype=torch.float64) for s in streams]
    assert all(pw.shape[0] == fpw.value for pw in per_wave), (per_wave[0].shape, fpw.value)
    pad = torch.zeros(RING, 512, dtype=torch.float64)  # the ring prefetches RING fragments past the end
    stream = torch.stack([torch.cat([pw, pad]) for pw in per_wave]).to(torch.bfloat16).contiguous().to(device)
    pvec = torch.cat([t.reshape(-1) for t in params]).to(torch.float32).contiguous().to(device)
    assert pvec.numel() == npar.value, (pvec.numel(), npar.value)
    return stream, pvec, fpw.value


def transformer_fused(src, src_ld, out, out_ld, S, N, stream_w, fpw, params, n_enc, n_dec):
    check(lib().ccnet_transformer_fused(src.data_ptr(), src_ld, out.data_ptr(), out_ld, S, N, 256, 2, 1024, stream_w.data_ptr(), fpw,
                                        params.data_ptr(), n_enc, n_dec, stream_ptr()))


DECODE_SCORE_MAX_NCH = 64


def decode(heads, nc, in_h, in_w, out, scores=None):
    """heads: list of fp32 CUDA tensors viewed as (B, 5+nc, h, w) with arbitrary strides over (b, c) and a uniform
    pixel stride (NCHW contiguous or an NHWC buffer permuted to NCHW).  scores (optional, (B, A) fp32, 5 + nc <= 64): filled
    with obj * max class probability per anchor, which filter_candidates can read instead of the rows."""
    n = len(heads)
    B = heads[0].shape[0]
    ptrs = (C.c_void_p * n)(*[h.data_ptr() for h in heads])
    sb = (C.c_int64 * n)()
    sc = (C.c_int64 * n)()
    sp = (C.c_int64 * n)()
    hh = (C.c_int32 * n)()
    ww = (C.c_int32 * n)()
    for i, h in enumerate(heads):
        if h.dtype != torch.float32 or h.dim() != 4 or h.shape[1] != 5 + nc:
            raise ValueError("decode: head %d must be fp32 (B,%d,h,w)" % (i, 5 + nc))
        st = h.stride()
        if st[2] != h.shape[3] * st[3]:
            raise ValueError("decode: head %d rows are not uniformly strided" % i)
        sb[i], sc[i], sp[i] = st[0], st[1], st[3]
        hh[i], ww[i] = h.shape[2], h.shape[3]
    check(lib().ccnet_decode_scored(ptrs, sb, sc, sp, hh, ww, n, B, nc, float(in_h), float(in_w), out.data_ptr(),
                                    scores.data_ptr() if scores is not None else None, stream_ptr()))


def filter_candidates(pred, nc, conf, cand, counts, scores=None):
    B, A, _ = pred.shape
    nws = int(lib().ccnet_filter_workspace_ints(B, A))
    # many anchors per image: one workgroup per 32768 anchors after a counting pass
    ws = torch.empty(nws, dtype=torch.int32, device=pred.device) if nws else None
    check(lib().ccnet_filter_scored(pred.data_ptr(), scores.data_ptr() if scores is not None else None, B, A, nc, float(np.float32(conf)),
                                    cand.data_ptr(), counts.data_ptr(), ws.data_ptr() if nws else None, stream_ptr()))


def nms_workspace_bytes(B, nmax):
    return int(lib().ccnet_nms_workspace_bytes(B, nmax))


def nms(cand, counts, nmax, thr, image_hw, det, keep_idx, nkeep, workspace):
    B, A, _ = cand.shape
    check(lib().ccnet_nms(cand.data_ptr(), counts.data_ptr(), B, A, nmax, float(thr), image_hw.data_ptr(), det.data_ptr(),
                          keep_idx.data_ptr(), nkeep.data_ptr(), workspace.data_ptr(), stream_ptr()))


def plan_set_lane(lane):
    check(lib().ccnet_plan_set_lane(lane))


def plan_wait(dst_lane, src_lane):
    check(lib().ccnet_plan_wait(dst_lane, src_lane))


class Recorder:
    """with Recorder() as r: ...op wrappers...  -> r.plan is a replayable ccnet_plan handle."""

    def __init__(self):
        self.handle = C.c_void_p()

    def __enter__(self):
        check(lib().ccnet_plan_begin(C.byref(self.handle)))
        return self

    def __exit__(self, et, ev, tb):
        lib().ccnet_plan_end()
        if et is not None:
            lib().ccnet_plan_destroy(self.handle)
            self.handle = C.c_void_p()
        return False


class Plan:
    def __init__(self, handle, keepalive):
        self.handle = handle
        self.keepalive = keepalive  # tensors whose device pointers the plan has baked in

    @property
    def num_ops(self):
        return lib().ccnet_plan_num_ops(self.handle)

    def run(self, x=None):
        check(lib().ccnet_plan_run(self.handle, None if x is None else x.data_ptr(), stream_ptr()))

    def op_info(self):
        """[(kind, flops, bytes)] per recorded op (include/ccnet_hip.h CCNET_OP_*)."""
        out = []
        k, f, b = C.c_int32(), C.c_double(), C.c_double()
        for i in range(self.num_ops):
            check(lib().ccnet_plan_op_info(self.handle, i, C.byref(k), C.byref(f), C.byref(b)))
            out.append((k.value, f.value, b.value))
        return out

    def run_timed(self, x=None):
        """Run with a HIP event pair around every op on the current stream; returns ms per op (synchronises)."""
        n = self.num_ops
        ms = (C.c_float * n)()
        check(lib().ccnet_plan_run_timed(self.handle, None if x is None else x.data_ptr(), stream_ptr(), ms))
        return list(ms)

    def __del__(self):
        try:
            if self.handle:
                lib().ccnet_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
