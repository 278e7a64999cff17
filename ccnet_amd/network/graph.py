"""Architecture description of the four CCNet detectors, written once and walked twice:

  * `declare(kind, ...)` registers parameter holders (torch.nn leaf modules used ONLY as named parameter storage)
    so that `Net.state_dict()` has exactly the reference's keys and shapes and `load_model_weights`
    (utils/util.py:90-108) keeps working;
  * `compile_plan(...)` folds BN in fp64, packs weights for the MFMA kernels, lays activations out as NHWC buffers
    and records the whole forward as one libccnet_hip plan.  No ATen compute op is ever issued.

Dataflow follows network/net.py:194-275, net_deploy.py:189-270, net_amff.py:203-298, net_small_target.py:201-300;
block semantics follow network/model_block.py and network/attention_block.py (cited inline).
Fusions relative to the reference (all exact in real arithmetic):
  - BN folded into the preceding conv;  SpaceToDepth + 3x3 == 6x6 stride-2 conv on the un-shuffled tensor;
  - CSP conv1 || conv2, and the head's cls_convs[0] || reg_convs[0], run as one GEMM over the shared input;
  - torch.cat never copies (producers write channel slices; consumers read two segments);
  - nn.Upsample is an addressing mode of the consumer;  reg_preds || obj_preds is one GEMM;
  - residual adds (Bottleneck, Transformer) happen in the producing GEMM's epilogue.
"""
import math
import os

import torch
import torch.nn as nn

from .. import _lib, ops
from .._lib import BF16, F32

KINDS = ("net", "net_deploy", "net_amff", "net_small_target", "net_old")


# ============================================================================================ declaration
class _Holder(nn.Module):
    """Pure container; never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder modules are not callable; use Net.forward")


def _mount(root, path, module):
    parts = path.split(".")
    cur = root
    for p in parts[:-1]:
        if p not in cur._modules:
            cur.add_module(p, _Holder())
        cur = cur._modules[p]
    cur.add_module(parts[-1], module)


class Declarer:
    def __init__(self, root):
        self.root = root

    def base_conv(self, p, cin, cout, k, stride=1):
        _mount(self.root, p + ".conv", nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=False))
        _mount(self.root, p + ".bn", nn.BatchNorm2d(cout))

    def conv(self, p, cin, cout, k, bias, groups=1):
        _mount(self.root, p, nn.Conv2d(cin, cout, k, 1, (k - 1) // 2, groups=groups, bias=bias))

    def bn(self, p, c):
        _mount(self.root, p, nn.BatchNorm2d(c))

    def module(self, p, m):
        _mount(self.root, p, m)


def _cfg(kind, depth, width, num_classes):
    bc = int(width * 64)
    bd = max(round(depth * 3), 1)
    return dict(kind=kind, bc=bc, bd=bd, nn=round(3 * depth), nc=num_classes, width=width,
                amff=(kind == "net_amff"), s2d=(kind == "net"), vit=(kind in ("net", "net_old")),
                small=(kind == "net_small_target"))


def _csp_decl(d, p, cin, cout, n, amff):
    h = int(cout * 0.5)
    d.base_conv(p + ".conv1", cin, h, 1)
    d.base_conv(p + ".conv2", cin, h, 1)
    d.base_conv(p + ".conv3", 2 * h, cout, 1)
    for i in range(n):
        q = "%s.m.%d" % (p, i)
        if amff:
            d.conv(q + ".conv1", h, h, 1, False)
            d.bn(q + ".bn", h)
            d.conv(q + ".conv2", h, h, 3, False)
        else:
            d.base_conv(q + ".conv1", h, h, 1)
            d.base_conv(q + ".conv2", h, h, 3)


def _down_decl(d, p, cin, cout, s2d):
    if s2d:
        d.base_conv(p + ".conv", cin * 4, cout, 3)  # SpaceToDepth.conv (model_block.py:164)
    else:
        d.base_conv(p, cin, cout, 3, 2)


def _cbam_decl(d, p, c, ratio=8, k=7):
    d.conv(p + ".channel_attention.fc1", c, c // ratio, 1, False)
    d.conv(p + ".channel_attention.fc2", c // ratio, c, 1, False)
    d.module(p + ".spatial_attention.conv1", nn.Conv2d(2, 1, k, padding=k // 2, bias=False))


def _asff_decl(d, p, level, width):
    dims = [int(1024 * width), int(512 * width), int(256 * width)]
    inter = dims[level]
    if level == 0:
        d.base_conv(p + ".stride_level_1", dims[1], inter, 3, 2)
        d.base_conv(p + ".stride_level_2", dims[2], inter, 3, 2)
    elif level == 1:
        d.base_conv(p + ".compress_level_0", dims[0], inter, 1)
        d.base_conv(p + ".stride_level_2", dims[2], inter, 3, 2)
    else:
        d.base_conv(p + ".compress_level_0", dims[0], inter, 1)
        d.base_conv(p + ".compress_level_1", dims[1], inter, 1)
    d.base_conv(p + ".expand", inter, dims[level], 3)
    for i in range(3):
        d.base_conv(p + ".weight_level_%d" % i, inter, 16, 1)
    d.base_conv(p + ".weight_levels", 48, 3, 1)


def declare(root, kind, depth, width, num_classes):
    """Registers every parameter of network/<kind>.py's Net under the reference's names."""
    c = _cfg(kind, depth, width, num_classes)
    d = Declarer(root)
    bc, bd = c["bc"], c["bd"]
    if c["s2d"]:
        d.base_conv("stem.conv", 12, bc, 3)
    elif kind == "net_old":
        d.base_conv("stem.0", 3, bc, 3, 2)  # nn.Sequential(BaseConv(3, bc, 3, 2)), network/net_old.py:41-43
    else:
        d.module("stem.conv", nn.Conv2d(3, bc, 6, 2, 2, bias=False))
        d.bn("stem.bn", bc)
    n3 = bd if c["s2d"] else bd * 3
    for name, cin, n in (("dark2", bc, bd), ("dark3", bc * 2, n3), ("dark4", bc * 4, bd * 3)):
        _down_decl(d, name + ".0", cin, cin * 2, c["s2d"])
        _csp_decl(d, name + ".1", cin * 2, cin * 2, n, c["amff"])
    _down_decl(d, "dark5.0", bc * 8, bc * 16, c["s2d"])
    d.base_conv("dark5.1.conv1", bc * 16, bc * 8, 1)
    d.base_conv("dark5.1.conv2", bc * 32, bc * 16, 1)
    if c["vit"]:
        C5, dm = bc * 16, bc * 8
        d.conv("dark5.2.conv1", C5, C5, 3, True, groups=C5)
        d.conv("dark5.2.conv2", C5, dm, 1, True)
        d.module("dark5.2.transformer", nn.Transformer(d_model=dm, nhead=2, num_encoder_layers=6, num_decoder_layers=6,
                                                        dim_feedforward=bc * 32))
        d.conv("dark5.2.conv3", dm, C5, 1, True)
        d.conv("dark5.2.conv4", C5 * 2, C5, 3, True)
        _csp_decl(d, "dark5.3", C5, C5, bd, False)
    else:
        _csp_decl(d, "dark5.2", bc * 16, bc * 16, bd, c["amff"])

    w = width
    if c["small"]:
        ch = [int(128 * w), int(256 * w), int(512 * w), int(1024 * w)]
    else:
        ch = [int(256 * w), int(512 * w), int(1024 * w)]
    top = len(ch) - 1
    nn_ = c["nn"]
    d.base_conv("lateral_conv0", ch[top], ch[top - 1], 1)
    _csp_decl(d, "C3_p4", 2 * ch[top - 1], ch[top - 1], nn_, c["amff"])
    d.base_conv("reduce_conv1", ch[top - 1], ch[top - 2], 1)
    _csp_decl(d, "C3_p3", 2 * ch[top - 2], ch[top - 2], nn_, c["amff"])
    if c["small"]:
        d.base_conv("reduce_conv2", ch[1], ch[0], 1)
        _csp_decl(d, "C3_p2", 2 * ch[0], ch[0], nn_, False)
        _down_decl(d, "bu_conv3", ch[0], ch[0], False)
        _csp_decl(d, "C3_n2", 2 * ch[0], ch[1], nn_, False)
    _down_decl(d, "bu_conv2", ch[top - 2], ch[top - 2], c["s2d"])
    _csp_decl(d, "C3_n3", 2 * ch[top - 2], ch[top - 1], nn_, c["amff"])
    _down_decl(d, "bu_conv1", ch[top - 1], ch[top - 1], c["s2d"])
    _csp_decl(d, "C3_n4", 2 * ch[top - 1], ch[top], nn_, c["amff"])
    if c["amff"]:
        nk = [512, 256, 512, 1024]
        for name, cc in (("eca_1", ch[2]), ("eca_2", ch[1]), ("eca_3", ch[0]), ("eca_fp1", int(nk[0] * w)),
                         ("eca_pa1", int(nk[1] * w)), ("eca_pa2", int(nk[2] * w)), ("eca_pa3", int(nk[3] * w))):
            d.module(name + ".conv", nn.Conv1d(1, 1, 3, padding=1, bias=False))
        for i in range(3):
            _asff_decl(d, "asff_%d" % (i + 1), i, w)
    else:
        names = ["layer1_cbam", "layer2_cbam", "layer3_cbam", "layer4_cbam"][-len(ch):]
        for name, cc in zip(names, ch):
            _cbam_decl(d, name, cc)
    hc = int(256 * w)
    for i, cc in enumerate(ch):
        d.base_conv("stems.%d" % i, cc, hc, 1)
        for br in ("cls_convs", "reg_convs"):
            d.base_conv("%s.%d.0" % (br, i), hc, hc, 3)
            d.base_conv("%s.%d.1" % (br, i), hc, hc, 3)
        d.conv("cls_preds.%d" % i, hc, c["nc"], 1, True)
        d.conv("reg_preds.%d" % i, hc, 4, 1, True)
        d.conv("obj_preds.%d" % i, hc, 1, 1, True)
    return c


# ============================================================================================ compilation
class T:
    """A channel slice of an NHWC activation buffer; `shift` = pending nearest-neighbour upsampling."""

    def __init__(self, buf, coff, C, H, W, shift=0):
        self.buf, self.coff, self.C, self.H, self.W, self.shift = buf, coff, C, H, W, shift

    def sl(self):
        return _lib.Slice(self.buf.data_ptr(), self.buf.shape[-1], self.coff, self.C, self.shift)

    def part(self, off, C):
        return T(self.buf, self.coff + off, C, self.H, self.W, self.shift)

    def up(self, k=1):
        return T(self.buf, self.coff, self.C, self.H << k, self.W << k, self.shift + k)


class Compiler:
    def __init__(self, sd, cfg, B, H, W, dtype, device, act):
        self.sd, self.c, self.B, self.H, self.W, self.dtype, self.dev, self.act = sd, cfg, B, H, W, dtype, device, act
        self.tdt = _lib.torch_dtype(dtype)
        self.keep = []  # every device tensor the plan points into
        self.taps = {}  # name -> T, for debugging / tap-level parity tests

    # ---- storage
    def new(self, C, H, W, dtype=None):
        t = torch.empty(self.B, H, W, C, dtype=self.tdt if dtype is None else dtype, device=self.dev)
        self.keep.append(t)
        return T(t, 0, C, H, W)

    def dev_f32(self, t):
        t = t.detach().to(torch.float32).to(self.dev).contiguous()
        self.keep.append(t)
        return t

    # ---- weights
    def _bn(self, p):
        s = self.sd
        return (s[p + ".weight"], s[p + ".bias"], s[p + ".running_mean"], s[p + ".running_var"])

    def folded(self, p, kind="base"):
        """-> (weight OIHW fp64, bias fp64).  kind: base = <p>.conv + <p>.bn ; plain = <p>.weight (+ <p>.bias)."""
        if kind == "base":
            return ops.fold_bn(self.sd[p + ".conv.weight"], self._bn(p + ".bn"))
        return ops.fold_bn(self.sd[p + ".weight"], None, self.sd.get(p + ".bias"))

    def pack(self, w, b):
        pc = ops.PackedConv(w, b, self.dtype, self.dev)
        self.keep.append(pc)
        return pc

    # ---- emitters
    def conv3x3_halo(self, w, b, ins, s2d, act, out=None, residual=None):
        """3x3 s1 p1 (s2d False), SpaceToDepth + 3x3 with the un-shuffled input (True) or 3x3 s2 p1 (2) on ccnet_conv_halo when the
        shape qualifies, else None."""
        if self.dtype != BF16 or os.environ.get("CCNET_NO_HALO_CONV", "0") != "0":
            return None
        # stride 2 on this kernel (five two-tap passes, ten taps multiplied for nine) measured EQUAL to the implicit GEMM inside the
        # networks that have such layers (same-box A/B: net_deploy at 32 images 2.164 / 2.176 against 2.172 / 2.169 ms per step;
        # single layers +16 % ... -10 %), so the implicit GEMM keeps them; CCNET_HALO_S2=1 switches them over
        if s2d == 2 and os.environ.get("CCNET_HALO_S2", "0") == "0":
            return None
        x0 = ins[0]
        cout, cvirt = w.shape[0], w.shape[1]
        Ho, Wo = (x0.H // 2, x0.W // 2) if s2d else (x0.H, x0.W)
        if s2d and (x0.H % 2 or x0.W % 2 or len(ins) != 1):
            return None
        probe_out = out.sl() if out is not None else _lib.Slice(None, cout, 0, cout, 0)
        if not ops.conv_halo_ok(self.dtype, None, [t.sl() for t in ins], cvirt, cout, probe_out, None if residual is None else residual.sl()):
            return None
        # The choice must not depend on the batch size: the two kernels round alike but accumulate in different orders, and image i of
        # a batch has to equal image i run alone bit for bit (tests/test_forward_gpu.py: batch independence; sharded == unsharded).
        # The 20 x 20 maps with fewer than four column blocks were kept on ccnet_conv2d's 128-position tiles at first (a per-layer
        # microbenchmark had them equal); inside the network the halo kernel is the faster one there too (same-box A/B,
        # scripts/ab_env.sh: step 4.706 -> 4.643 ms for `net` at 64 images, 2.263 -> 2.255 for `net_deploy` at 32).
        # CCNET_HALO_SMALL_MAPS=0 brings the old rule back.
        if Ho * Wo <= 400 and cout < 512 and os.environ.get("CCNET_HALO_SMALL_MAPS", "1") == "0":
            return None
        pk = ops.PackedHaloS2(w, b, self.dev) if s2d == 2 else ops.PackedHalo(w, b, self.dev)
        self.keep.append(pk)
        if out is None:
            out = self.new(cout, Ho, Wo)
        assert (out.H, out.W, out.C) == (Ho, Wo, cout), "conv output slice mismatch"
        ops.conv_halo([t.sl() for t in ins], pk, int(s2d), self.B, x0.H, x0.W, out.sl(), act, None if residual is None else residual.sl())
        return out

    def conv1x1_pw(self, w, b, ins, act, out=None, residual=None):
        """1x1 s1 on ccnet_conv_pw (pixels in LDS, weights streamed into registers) when the shape qualifies, else None.  Its results
        are bit-identical to ccnet_conv2d's, so the rule below is a speed matter only: K >= 256 (CCNET_PW_MIN_K), where the implicit
        GEMM's short K loop is latency-bound; the K = 128 layers stay on the streaming kernel."""
        if self.dtype != BF16 or os.environ.get("CCNET_NO_PW_CONV", "0") != "0":
            return None
        x0 = ins[0]
        cout, cin = w.shape[0], w.shape[1]
        if cin < int(os.environ.get("CCNET_PW_MIN_K", "256")):
            return None
        probe_out = out.sl() if out is not None else _lib.Slice(None, cout, 0, cout, 0)
        if not ops.conv_pw_ok(self.dtype, None, [t.sl() for t in ins], cout, probe_out, None if residual is None else residual.sl()):
            return None
        pk = ops.PackedPw(w, b, self.dev)
        self.keep.append(pk)
        if out is None:
            out = self.new(cout, x0.H, x0.W)
        assert (out.H, out.W, out.C) == (x0.H, x0.W, cout), "conv output slice mismatch"
        ops.conv_pw([t.sl() for t in ins], pk, self.B, x0.H, x0.W, out.sl(), act, None if residual is None else residual.sl())
        return out

    def conv_w(self, w, b, ins, stride, pad, act, out=None, residual=None, out_dtype=None):
        if w.shape[-1] == 3 and w.shape[-2] == 3 and stride == 1 and pad == 1 and out_dtype is None:
            got = self.conv3x3_halo(w, b, ins, False, act, out, residual)
            if got is not None:
                return got
        if w.shape[-1] == 3 and w.shape[-2] == 3 and stride == 2 and pad == 1 and out_dtype is None:
            got = self.conv3x3_halo(w, b, ins, 2, act, out, residual)
            if got is not None:
                return got
        if w.shape[-1] == 1 and w.shape[-2] == 1 and stride == 1 and pad == 0 and out_dtype is None:
            got = self.conv1x1_pw(w, b, ins, act, out, residual)
            if got is not None:
                return got
        pc = self.pack(w, b)
        x0 = ins[0]
        H, W = x0.H, x0.W
        for t in ins:
            assert (t.H, t.W) == (H, W), "conv inputs disagree on size"
        Ho = (H + 2 * pad - pc.kh) // stride + 1
        Wo = (W + 2 * pad - pc.kw) // stride + 1
        if out is None:
            out = self.new(pc.cout, Ho, Wo, None if out_dtype is None else _lib.torch_dtype(out_dtype))
        assert (out.H, out.W, out.C) == (Ho, Wo, pc.cout), "conv output slice mismatch"
        ops.conv2d([t.sl() for t in ins], pc, self.B, H, W, stride, pad, out.sl(), act,
                   None if residual is None else residual.sl(), out_dtype)
        return out

    def base_conv(self, p, ins, stride=1, act=None, out=None, residual=None, pad=None):
        w, b = self.folded(p)
        k = w.shape[-1]
        return self.conv_w(w, b, ins, stride, (k - 1) // 2 if pad is None else pad, self.act if act is None else act, out, residual)

    def down(self, p, x):
        """SpaceToDepth+BaseConv k3 (net.py) as a 6x6 s2 p2 conv, or BaseConv k3 s2 (other nets)."""
        if self.c["s2d"]:
            w, b = self.folded(p + ".conv")
            got = self.conv3x3_halo(w, b, [x], True, self.act)
            if got is not None:
                return got
            return self.conv_w(ops.s2d_to_6x6(w), b, [x], 2, 2, self.act)
        return self.base_conv(p, [x], 2)

    def csp(self, p, ins, n, shortcut, amff=False):
        """CSPLayer / AMFFCSPLayer (model_block.py:86-149)."""
        w1, b1 = self.folded(p + ".conv1")
        w2, b2 = self.folded(p + ".conv2")
        h = w1.shape[0]
        x0 = ins[0]
        # hidden 32 (160^2 at 640: 392 -> 233 us at bs64) and hidden 64 (dark3.1 at 80^2: 198 -> 164 us).  With two input segments
        # (C3_p3) hidden 64 measured equal to the five launches when timed alone and was left out in round 2; inside the network it
        # wins (same-box A/B, scripts/ab_env.sh: step -0.6 ... -2 % on `net`, `net_deploy` and `net_small_target`), so it is taken
        # too.  CCNET_FUSED_CSP_ALL=0 brings the round-2 rule back.
        wide_ok = h == 64 and (len(ins) == 1 or os.environ.get("CCNET_FUSED_CSP_ALL", "1") != "0")
        if (self.dtype == BF16 and n == 1 and not amff and (h == 32 or wide_ok) and self.act == "silu" and x0.H % 8 == 0 and x0.W % 16 == 0
                and all(t.C % 64 == 0 and t.shift <= 1 for t in ins) and self.sd[p + ".conv3.conv.weight"].shape[0] == 2 * h
                and os.environ.get("CCNET_NO_FUSED_CSP", "0") == "0"):
            # the whole layer in one launch: x_1, x_2 and the bottleneck's hidden map never leave LDS (csp_fused.hip)
            q = p + ".m.0"
            wa, ba = self.folded(q + ".conv1")
            wb, bb = self.folded(q + ".conv2")
            w3, b3 = self.folded(p + ".conv3")
            pk = ops.PackedCsp(torch.cat([w1, w2], 0), torch.cat([b1, b2], 0), wa, ba, wb, bb, w3, b3, self.dev)
            self.keep.append(pk)
            out = self.new(2 * h, x0.H, x0.W)
            ops.csp_fused([t.sl() for t in ins], pk, self.B, x0.H, x0.W, shortcut, self.act, out.sl())
            return out
        cat = self.conv_w(torch.cat([w1, w2], 0), torch.cat([b1, b2], 0), ins, 1, 0, self.act)  # [x_1 | x_2]
        x1 = cat.part(0, h)
        if self._bneck_fused_wins(h, x0.H, x0.W, amff):
            # every Bottleneck (1x1 -> 3x3 [+ x]) in one launch with its hidden map in LDS (bneck_fused.hip).  Its results are
            # bit-identical to the two launches, so this choice (which looks at the batch size) never shows in an output.  The
            # kernel cannot work in place (a tile's halo is its neighbours' output): x_1 moves through scratch maps and, with two
            # or more bottlenecks, ends where the separate launches leave it; a single bottleneck leaves it in a scratch map
            # and conv3 reads [x_1 | x_2] as two segments -- the same k order as the concatenated buffer.
            pong = [None, None]
            cur = x1
            for i in range(n):
                q = "%s.m.%d" % (p, i)
                wa, ba = self.folded(q + ".conv1")
                wb, bb = self.folded(q + ".conv2")
                pk = ops.PackedBneck(wa, ba, wb, bb, self.dev)
                self.keep.append(pk)
                if i == n - 1 and n >= 2:
                    dst = x1
                else:
                    if pong[i & 1] is None:
                        pong[i & 1] = self.new(h, x0.H, x0.W)
                    dst = pong[i & 1]
                ops.bneck_fused(cur.sl(), pk, self.B, x0.H, x0.W, shortcut, self.act, dst.sl())
                cur = dst
            return self.base_conv(p + ".conv3", [cat] if n >= 2 else [cur, cat.part(h, h)])
        for i in range(n):
            q = "%s.m.%d" % (p, i)
            if amff:
                # conv1 -> BN (no act) -> conv2 (no BN) -> act -> (+x): BN folds into conv1, never conv2 (zero padding)
                wa, ba = ops.fold_bn(self.sd[q + ".conv1.weight"], self._bn(q + ".bn"))
                t = self.conv_w(wa, ba, [x1], 1, 0, "none")
                wb, bb = ops.fold_bn(self.sd[q + ".conv2.weight"])
                self.conv_w(wb, None, [t], 1, 1, self.act, out=x1, residual=x1 if shortcut else None)
            else:
                t = self.base_conv(q + ".conv1", [x1])
                self.base_conv(q + ".conv2", [t], out=x1, residual=x1 if shortcut else None)
        return self.base_conv(p + ".conv3", [cat])

    def _bneck_fused_wins(self, h, H, W, amff):
        """Where ccnet_bneck_fused beats conv1x1 + conv3x3 (same-box A/B, scripts/ab_env.sh and scripts/bneck_bench.py): 64 channels
        always (two workgroups per CU: 37 against 47 us at 32 x 80^2, 63 against 78 at 64 x 80^2); 128 and 256 channels (96 KB of
        LDS: one workgroup of four waves per CU, every SiLU exposed) while every tile finds a CU of its own (128: 31 against 36 us
        at 32 x 40^2, but 60 against 53 at 64 x 40^2, 431 tiles).  CCNET_NO_FUSED_BNECK=1 turns it off, CCNET_FUSED_BNECK_ALL=1 takes
        every shape the kernel has."""
        if (self.dtype != BF16 or amff or h not in (64, 128, 256) or self.act != "silu"
                or os.environ.get("CCNET_NO_FUSED_BNECK", "0") != "0"):
            return False
        if h == 256 and W + 2 > 23:        # bneck_fused.hip: 128-position linear tiles, maps up to 21 wide
            return False
        if h == 64 or os.environ.get("CCNET_FUSED_BNECK_ALL", "0") != "0":
            return True
        # bneck_fused.hip's geometry choice
        lin = h == 256 or (W + 2 <= 47 and (W + 2) * (H + 1) < ((W + 15) // 16) * ((H + 15) // 16) * 256)
        tpx = 128 if h == 256 else 256
        tiles = ((self.B * (H + 1) + 1) * (W + 2) + tpx - 1) // tpx if lin else self.B * ((W + 15) // 16) * ((H + 15) // 16)
        return tiles <= torch.cuda.get_device_properties(self.dev).multi_processor_count

    def sppf(self, p, x):
        w, b = self.folded(p + ".conv1")
        mid = w.shape[0]
        buf = self.new(4 * mid, x.H, x.W)
        self.conv_w(w, b, [x], 1, 0, self.act, out=buf.part(0, mid))
        ops.sppf_pool(buf.part(0, mid).sl(), self.dtype, self.B, x.H, x.W)
        return self.base_conv(p + ".conv2", [buf])

    def _scratch(self, x):
        s = torch.empty(ops.gate_scratch_floats(self.B, x.H, x.W, x.C), dtype=torch.float32, device=self.dev)
        self.keep.append(s)
        return s

    def cbam(self, p, x):
        fc1 = self.dev_f32(self.sd[p + ".channel_attention.fc1.weight"].reshape(-1, x.C))
        fc2 = self.dev_f32(self.sd[p + ".channel_attention.fc2.weight"].reshape(x.C, -1))
        wk = self.sd[p + ".spatial_attention.conv1.weight"]
        w7 = self.dev_f32(wk.reshape(2, wk.shape[-1], wk.shape[-1]))
        out = self.new(x.C, x.H, x.W)
        ops.cbam(x.sl(), self.dtype, self.B, x.H, x.W, fc1, fc2, w7, self._scratch(x), out.sl())
        return out

    def eca(self, p, x):
        w3 = self.dev_f32(self.sd[p + ".conv.weight"].reshape(3))
        out = self.new(x.C, x.H, x.W)
        ops.eca(x.sl(), self.dtype, self.B, x.H, x.W, w3, self._scratch(x), out.sl())
        return out

    # ---- Transformer (rows = B*HW tokens of E channels, stored as a 1 x rows "image")
    def linear(self, w, b, x, rows, act="none", residual=None, out=None):
        pc = self.pack(w.reshape(w.shape[0], w.shape[1], 1, 1), b)
        if out is None:
            t = torch.empty(rows, pc.cout, dtype=self.tdt, device=self.dev)
            self.keep.append(t)
            out = T(t, 0, pc.cout, 1, rows)
        ops.conv2d([x.sl()], pc, 1, 1, rows, 1, 0, out.sl(), act, None if residual is None else residual.sl())
        return out

    def layernorm(self, p, x, rows):
        E = x.C
        out = torch.empty(rows, E, dtype=self.tdt, device=self.dev)
        self.keep.append(out)
        ops.layernorm(x.buf, None, self.dev_f32(self.sd[p + ".weight"]), self.dev_f32(self.sd[p + ".bias"]), out, self.dtype, rows, E)
        return T(out, 0, E, 1, rows)

    def mha(self, p, xq, xkv, rows, N, nhead, residual):
        """nn.MultiheadAttention (packed in_proj rows [0:E]=Q, [E:2E]=K, [2E:3E]=V) + out_proj, residual added in the epilogue."""
        E = xq.C
        W = self.sd[p + ".in_proj_weight"].to(torch.float64)
        bvec = self.sd[p + ".in_proj_bias"].to(torch.float64)
        S = rows // N
        if xq is xkv:
            qkv = self.linear(W, bvec, xq, rows)
            q, k, v = qkv.part(0, E), qkv.part(E, E), qkv.part(2 * E, E)
        else:
            q = self.linear(W[:E], bvec[:E], xq, rows)
            kv = self.linear(W[E:], bvec[E:], xkv, rows)
            k, v = kv.part(0, E), kv.part(E, E)
        att = torch.empty(rows, E, dtype=self.tdt, device=self.dev)
        self.keep.append(att)
        att = T(att, 0, E, 1, rows)
        ops.attention(q.sl(), k.sl(), v.sl(), self.dtype, S, S, N, nhead, E // nhead, att.sl())
        return self.linear(self.sd[p + ".out_proj.weight"].to(torch.float64), self.sd[p + ".out_proj.bias"].to(torch.float64),
                           att, rows, residual=residual)

    def ffn(self, p, x, rows):
        h = self.linear(self.sd[p + ".linear1.weight"].to(torch.float64), self.sd[p + ".linear1.bias"].to(torch.float64), x, rows, act="relu")
        return self.linear(self.sd[p + ".linear2.weight"].to(torch.float64), self.sd[p + ".linear2.bias"].to(torch.float64), h, rows, residual=x)

    def transformer(self, p, src, rows, N, nhead=2, n_enc=6, n_dec=6):
        """nn.Transformer(src, src) in eval mode: post-norm, ReLU (model_block.py:270-271, 290)."""
        S = rows // N
        ff = self.sd[p + ".encoder.layers.0.linear1.weight"].shape[0] if n_enc else 0
        if (self.dtype == BF16 and src.C == 256 and nhead == 2 and ff == 1024 and S <= 64 and src.coff == 0
                and os.environ.get("CCNET_NO_FUSED_TRANSFORMER", "0") == "0"):
            # the whole stack in one launch (one workgroup per spatial position, transformer_fused.hip)
            wst, pvec, fpw = ops.pack_transformer(self.sd, p, n_enc, n_dec, self.dev)
            self.keep.extend([wst, pvec])
            out = torch.empty(rows, src.C, dtype=self.tdt, device=self.dev)
            self.keep.append(out)
            ops.transformer_fused(src.buf, src.buf.shape[-1], out, src.C, S, N, wst, fpw, pvec, n_enc, n_dec)
            return T(out, 0, src.C, 1, rows)
        x = src
        for i in range(n_enc):
            q = "%s.encoder.layers.%d" % (p, i)
            x = self.layernorm(q + ".norm1", self.mha(q + ".self_attn", x, x, rows, N, nhead, x), rows)
            x = self.layernorm(q + ".norm2", self.ffn(q, x, rows), rows)
        mem = self.layernorm(p + ".encoder.norm", x, rows)
        y = src
        for i in range(n_dec):
            q = "%s.decoder.layers.%d" % (p, i)
            y = self.layernorm(q + ".norm1", self.mha(q + ".self_attn", y, y, rows, N, nhead, y), rows)
            y = self.layernorm(q + ".norm2", self.mha(q + ".multihead_attn", y, mem, rows, N, nhead, y), rows)
            y = self.layernorm(q + ".norm3", self.ffn(q, y, rows), rows)
        return self.layernorm(p + ".decoder.norm", y, rows)

    def mobilevit(self, p, x):
        """MobileViT.forward (model_block.py:279-298).  Tokens are (B, HW, d) read as (S=B, N=HW, E=d)."""
        C5, H, W = x.C, x.H, x.W
        wd = self.sd[p + ".conv1.weight"].to(torch.float64)  # (C,1,3,3)
        y1 = self.new(C5, H, W)
        ops.dwconv3x3(x.sl(), self.dtype, self.B, H, W, self.dev_f32(wd.reshape(C5, 9).t()), self.dev_f32(self.sd[p + ".conv1.bias"]), y1.sl())
        w2, b2 = self.folded(p + ".conv2", "plain")
        tok = self.conv_w(w2, b2, [y1], 1, 0, "none")
        rows, N = self.B * H * W, H * W
        tok_rows = T(tok.buf, 0, tok.C, 1, rows)
        out_rows = self.transformer(p + ".transformer", tok_rows, rows, N)
        t_img = T(out_rows.buf, 0, out_rows.C, H, W)
        w3, b3 = self.folded(p + ".conv3", "plain")
        y = self.conv_w(w3, b3, [t_img], 1, 0, "none")
        w4, b4 = self.folded(p + ".conv4", "plain")
        return self.conv_w(w4, b4, [x, y], 1, 1, "none")

    def asff(self, p, level, pan):
        """ASFF.forward (model_block.py:364-412); pan = (pan_out2, pan_out1, pan_out0), level 0 = coarsest."""
        l0, l1, l2 = pan[2], pan[1], pan[0]
        silu = "silu"
        if level == 0:
            r0 = l0
            r1 = self.base_conv(p + ".stride_level_1", [l1], 2, silu, pad=1)
            pooled = self.new(l2.C, (l2.H - 1) // 2 + 1, (l2.W - 1) // 2 + 1)
            ops.maxpool3x3s2(l2.sl(), self.dtype, self.B, l2.H, l2.W, pooled.sl())
            r2 = self.base_conv(p + ".stride_level_2", [pooled], 2, silu, pad=1)
        elif level == 1:
            r0 = self.base_conv(p + ".compress_level_0", [l0], 1, silu).up(1)
            r1 = l1
            r2 = self.base_conv(p + ".stride_level_2", [l2], 2, silu, pad=1)
        else:
            r0 = self.base_conv(p + ".compress_level_0", [l0], 1, silu).up(2)
            r1 = self.base_conv(p + ".compress_level_1", [l1], 1, silu).up(1)
            r2 = l2
        H, W = r1.H, r1.W
        wcat = self.new(48, H, W)
        for i, r in enumerate((r0, r1, r2)):
            self.base_conv(p + ".weight_level_%d" % i, [r], 1, silu, out=wcat.part(16 * i, 16))
        lbuf = self.new(8, H, W)
        self.base_conv(p + ".weight_levels", [wcat], 1, silu, out=lbuf.part(0, 3))
        fused = self.new(r1.C, H, W)
        ops.asff_fuse(r0.sl(), r1.sl(), r2.sl(), lbuf.part(0, 3).sl(), self.dtype, self.B, H, W, fused.sl())
        return self.base_conv(p + ".expand", [fused], 1, silu)

    def head(self, feats):
        """Decoupled head (net.py:244-275).  Output per level: fp32 NHWC [B,h,w,5+nc] = [reg4 | obj1 | cls nc]."""
        outs = [None] * len(feats)
        # the levels are independent: the first (largest) stays on lane 0, the others run beside it on side lanes
        for k in range(1, len(feats)):
            ops.plan_wait(k, 0)
            ops.plan_set_lane(k)
            outs[k] = self.head_level(k, feats[k])
        ops.plan_set_lane(0)
        outs[0] = self.head_level(0, feats[0])
        for k in range(1, len(feats)):
            ops.plan_wait(0, k)
        return outs

    def head_level(self, k, x):
        nc = self.c["nc"]
        s = self.base_conv("stems.%d" % k, [x])
        wc, bcv = self.folded("cls_convs.%d.0" % k)
        wr, brv = self.folded("reg_convs.%d.0" % k)
        hc = wc.shape[0]
        both = self.conv_w(torch.cat([wc, wr], 0), torch.cat([bcv, brv], 0), [s], 1, 1, self.act)
        cls_f = self.base_conv("cls_convs.%d.1" % k, [both.part(0, hc)])
        reg_f = self.base_conv("reg_convs.%d.1" % k, [both.part(hc, hc)])
        out = self.new(5 + nc, x.H, x.W, torch.float32)
        wreg, breg = self.folded("reg_preds.%d" % k, "plain")
        wobj, bobj = self.folded("obj_preds.%d" % k, "plain")
        wcls, bcls = self.folded("cls_preds.%d" % k, "plain")
        if hc % 64 == 0 and os.environ.get("CCNET_SPLIT_PREDS", "0") == "0":
            # the three prediction convs of a level as ONE launch: a block-diagonal 1x1 over cat[reg_f | cls_f] (two input
            # segments, no copy).  The zero blocks add exact zeros to the fp32 accumulators, so the logits are unchanged.
            wp = torch.zeros(5 + nc, 2 * hc, 1, 1, dtype=wreg.dtype)
            wp[:4, :hc] = wreg
            wp[4:5, :hc] = wobj
            wp[5:, hc:] = wcls
            self.conv_w(wp, torch.cat([breg, bobj, bcls], 0), [reg_f, cls_f], 1, 0, "none", out=out, out_dtype=F32)
            return out
        self.conv_w(torch.cat([wreg, wobj], 0), torch.cat([breg, bobj], 0), [reg_f], 1, 0, "none", out=out.part(0, 5), out_dtype=F32)
        self.conv_w(wcls, bcls, [cls_f], 1, 0, "none", out=out.part(5, nc), out_dtype=F32)
        return out

    # ---- whole network
    def build(self, x_placeholder):
        c = self.c
        bc, bd, nn_ = c["bc"], c["bd"], c["nn"]
        H2, W2 = self.H // 2, self.W // 2
        l0 = self.new(bc, H2, W2)
        if c["s2d"]:
            w, b = self.folded("stem.conv")
            w6, stem_act = ops.s2d_to_6x6(w), self.act
        elif c["kind"] == "net_old":
            # 3x3 stride-2 pad-1 == the centre taps 1..3 of a 6x6 stride-2 pad-2 window (input offsets -1..+1)
            w, b = self.folded("stem.0")
            w6 = torch.zeros(w.shape[0], w.shape[1], 6, 6, dtype=w.dtype)
            w6[:, :, 1:4, 1:4] = w
            stem_act = self.act
        else:
            w6, b = ops.fold_bn(self.sd["stem.conv.weight"], self._bn("stem.bn"))
            stem_act = "silu"  # FocusReplaceConv is built without `act` (net_deploy.py:41)
        wp, bp = ops.pack_stem(w6, b, self.dev)
        self.keep += [wp, bp]
        ops.stem_conv(x_placeholder, wp, bp, bc, stem_act, l0.sl(), self.dtype)
        tp = self.taps
        tp["stem"] = l0

        amff = c["amff"]
        n3 = bd if c["s2d"] else bd * 3
        l1 = self.csp("dark2.1", [self.down("dark2.0", l0)], bd, True, amff)
        l2 = self.csp("dark3.1", [self.down("dark3.0", l1)], n3, True, amff)
        l3 = self.csp("dark4.1", [self.down("dark4.0", l2)], bd * 3, True, amff)
        # The gates of dark3 / dark4 (and dark2 in the four-level net) do not depend on dark5: on lane 3 they run beside it -- beside
        # the fused Transformer in `net` (400 positions on 256 CUs leave CUs idle in its second round), beside the under-filled 20 x 20
        # launches of dark5 in the other nets (three CBAMs one after the other were 7 % of net_deploy's step at 32 images).
        # CCNET_SIDE_GATES=0 brings the sequential order back.
        side_gates = os.environ.get("CCNET_SIDE_GATES", "1") != "0"
        g1 = None
        if side_gates:
            ops.plan_wait(3, 0)
            ops.plan_set_lane(3)
            if amff:
                g2, g3 = self.eca("eca_3", l2), self.eca("eca_2", l3)
            else:
                g2, g3 = self.cbam("layer2_cbam", l2), self.cbam("layer3_cbam", l3)
                if c["small"]:
                    g1 = self.cbam("layer1_cbam", l1)
            ops.plan_set_lane(0)
        t = self.sppf("dark5.1", self.down("dark5.0", l3))
        tp["dark2"], tp["dark3"], tp["dark4"], tp["sppf"] = l1, l2, l3, t
        if c["vit"]:
            t = self.mobilevit("dark5.2", t)
            tp["mobilevit"] = t
            l4 = self.csp("dark5.3", [t], bd, False)
        else:
            l4 = self.csp("dark5.2", [t], bd, False, amff)

        if side_gates:
            g4 = self.eca("eca_1", l4) if amff else self.cbam("layer4_cbam", l4)
            ops.plan_wait(0, 3)
        elif amff:
            g2, g3, g4 = self.eca("eca_3", l2), self.eca("eca_2", l3), self.eca("eca_1", l4)
        else:
            g2, g3, g4 = self.cbam("layer2_cbam", l2), self.cbam("layer3_cbam", l3), self.cbam("layer4_cbam", l4)

        tp["dark5"], tp["gate2"], tp["gate3"], tp["gate4"] = l4, g2, g3, g4
        fpn0 = self.base_conv("lateral_conv0", [g4])
        f0 = self.csp("C3_p4", [fpn0.up(), g3], nn_, False, amff)
        if amff:
            f0 = self.eca("eca_fp1", f0)
        fpn1 = self.base_conv("reduce_conv1", [f0])
        p2 = self.csp("C3_p3", [fpn1.up(), g2], nn_, False, amff)
        if amff:
            p2 = self.eca("eca_pa1", p2)
        p3 = None
        if c["small"]:
            if g1 is None:
                g1 = self.cbam("layer1_cbam", l1)
            fpn2 = self.base_conv("reduce_conv2", [p2])
            p3 = self.csp("C3_p2", [fpn2.up(), g1], nn_, False)
            p2 = self.csp("C3_n2", [self.base_conv("bu_conv3", [p3], 2), fpn2], nn_, False)
        early = not amff and not c["small"]
        outs = [None, None, None]
        if early:
            # pan_out2 is final here: its head level (the largest, 80x80 at 640) runs on lane 1 beside the bottom-up path
            ops.plan_wait(1, 0)
            ops.plan_set_lane(1)
            outs[0] = self.head_level(0, p2)
            ops.plan_set_lane(0)
        p1 = self.csp("C3_n3", [self.down("bu_conv2", p2), fpn1], nn_, False, amff)
        if amff:
            p1 = self.eca("eca_pa2", p1)
        if early:
            ops.plan_wait(2, 0)
            ops.plan_set_lane(2)
            outs[1] = self.head_level(1, p1)
            ops.plan_set_lane(0)
        p0 = self.csp("C3_n4", [self.down("bu_conv1", p1), fpn0], nn_, False, amff)
        if amff:
            p0 = self.eca("eca_pa3", p0)
            pan = (p2, p1, p0)
            p0, p1, p2 = self.asff("asff_1", 0, pan), self.asff("asff_2", 1, pan), self.asff("asff_3", 2, pan)
        tp["pan_out2"], tp["pan_out1"], tp["pan_out0"] = p2, p1, p0
        if early:
            outs[2] = self.head_level(2, p0)
            ops.plan_wait(0, 1)
            ops.plan_wait(0, 2)
            return outs
        feats = (p3, p2, p1, p0) if c["small"] else (p2, p1, p0)
        return self.head(feats)


def compile_plan(kind, sd, cfg, B, H, W, dtype, device, act):
    """Returns (ops.Plan, [head buffers (B,h,w,5+nc) fp32], static input tensor)."""
    if H % 32 or W % 32:
        raise ValueError("ccnet_amd: input height/width must be multiples of 32 (got %dx%d)" % (H, W))
    comp = Compiler(sd, cfg, B, H, W, dtype, device, act)
    x_static = torch.zeros(B, 3, H, W, dtype=torch.float32, device=device)
    comp.keep.append(x_static)
    with ops.Recorder() as rec:
        heads = comp.build(x_static)
    plan = ops.Plan(rec.handle, comp.keep)
    plan.taps = comp.taps
    return plan, [h.buf for h in heads], x_static
