"""Randomised parity sweep of ccnet_conv2d against torch (GPU box): shapes, strides, kernel sizes, one or two input segments
(with and without an upsampled first segment), residual, activations, output slices, both dtypes, several batch sizes --
so that every dispatch branch (halo-patch, LDS-DMA implicit GEMM, flat 1x1 path, generic fallback) meets inputs its tests
do not list.  Prints the worst relative error per kernel family and exits non-zero on a violation.

    python scripts/conv_fuzz.py [n_cases] [seed]
"""
import math, os, random, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops

DEV = "cuda"
ACTS = {"silu": F.silu, "relu": F.relu, "relu6": F.relu6, "lrelu": lambda t: F.leaky_relu(t, 0.1), "none": lambda t: t}


def tdt(code):
    return torch.bfloat16 if code == 1 else torch.float32


def q(x, code):
    return x.to(tdt(code)).float()


def nhwc(x, code, ldc, coff):
    B, C, H, W = x.shape
    buf = torch.zeros(B, H, W, ldc, dtype=tdt(code), device=DEV)
    buf[..., coff:coff + C] = x.permute(0, 2, 3, 1).to(DEV).to(tdt(code))
    return buf




def run(N, seed, verbose=True):
    """Returns (violations, {family: worst relative error})."""
    rnd = random.Random(seed)
    worst = {}
    bad = 0
    for it in range(N):
        code = rnd.choice([0, 1, 1])
        k, stride, pad = rnd.choice([(1, 1, 0), (1, 1, 0), (3, 1, 1), (3, 1, 1), (3, 2, 1), (6, 2, 2), (5, 1, 2), (1, 2, 0), (7, 1, 3)])
        B = rnd.choice([1, 2, 3, 5])
        H = rnd.choice([4, 7, 8, 13, 16, 20, 24, 40])
        W = rnd.choice([4, 8, 11, 16, 20, 24, 32, 40, 48])
        if k == 6:
            H, W = H + (H & 1), W + (W & 1)
        cin = rnd.choice([8, 16, 32, 64, 96, 128, 192, 256])
        cout = rnd.choice([5, 8, 16, 24, 32, 64, 80, 128, 192, 256])
        nseg = rnd.choice([1, 1, 1, 2])
        shift = rnd.choice([0, 0, 1]) if nseg == 2 and H % 2 == 0 and W % 2 == 0 else 0
        act = rnd.choice(list(ACTS))
        use_res = rnd.random() < 0.3
        g = torch.Generator().manual_seed(it)
        c0 = rnd.choice([c for c in (8, 16, 32, 64, 128) if c < cin] or [cin]) if nseg == 2 else cin
        if nseg == 2 and c0 >= cin:
            nseg = 1
        w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
        b = torch.randn(cout, generator=g) * 0.1
        pc = ops.PackedConv(w, b, code, DEV)
        if nseg == 1:
            x = torch.randn(B, cin, H, W, generator=g)
            off = rnd.choice([0, 8])
            xb = nhwc(x, code, cin + 16, off)
            ins = [ops.sl(xb, off, cin)]
            xr = q(x, code)
        else:
            a = torch.randn(B, c0, H >> shift, W >> shift, generator=g)
            bb = torch.randn(B, cin - c0, H, W, generator=g)
            ta, tb = nhwc(a, code, c0, 0), nhwc(bb, code, cin - c0 + 8, 8)
            ins = [ops.sl(ta, 0, c0, shift=shift), ops.sl(tb, 8, cin - c0)]
            au = F.interpolate(q(a, code), scale_factor=2, mode="nearest") if shift else q(a, code)
            xr = torch.cat([au, q(bb, code)], 1)
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        if Ho < 1 or Wo < 1:
            continue
        out_dt = 0 if cout == 5 else code
        ooff = 8 if cout % 8 == 0 else 0
        ldo = cout + 16 if cout % 8 == 0 else 11
        out = torch.zeros(B, Ho, Wo, ldo, dtype=tdt(out_dt), device=DEV)
        res, ref_res = None, 0
        if use_res:
            r = torch.randn(B, cout, Ho, Wo, generator=g)
            rb = nhwc(r, code, cout, 0)
            res, ref_res = ops.sl(rb), q(r, code)
        desc = "it%d code%d k%d s%d p%d B%d %dx%d cin%d(c0 %d shift %d) cout%d %s res%d" % (it, code, k, stride, pad, B, H, W, cin, c0, shift, cout, act, use_res)
        try:
            ops.conv2d(ins, pc, B, H, W, stride, pad, ops.sl(out, ooff, cout), act, res, out_dt)
            torch.cuda.synchronize()
        except RuntimeError as e:
            print("REJECTED", desc, str(e)[:120])
            bad += 1
            continue
        ref = ACTS[act](F.conv2d(xr, q(w, code), b, stride, pad)) + ref_res
        got = out[..., ooff:ooff + cout].float().cpu().permute(0, 3, 1, 2)
        err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
        tol = 2e-2 if code == 1 and out_dt == 1 else (6e-3 if code == 1 else 2e-4)
        fam = "k%d s%d %s" % (k, stride, "bf16" if code else "fp32")
        worst[fam] = max(worst.get(fam, 0.0), err)
        untouched = float(out[..., :ooff].float().abs().max()) if ooff else 0.0
        if not (err < tol) or untouched != 0.0:
            print("FAIL", desc, "err %.3e tol %.1e untouched %.3g" % (err, tol, untouched))
            bad += 1
    if verbose:
        for fam in sorted(worst):
            print("%-14s worst rel err %.2e" % (fam, worst[fam]))
        print("cases", N, "seed", seed, "violations", bad)
    return bad, worst


if __name__ == "__main__":
    n_ = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed_ = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sys.exit(1 if run(n_, seed_)[0] else 0)
