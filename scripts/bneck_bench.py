"""Times ccnet_bneck_fused against the two launches it replaces (run on the GPU box): python scripts/bneck_bench.py
Per-layer figures run at ramping clocks and say little about the network: a first look only; decisions come
from scripts/ab_env.sh CCNET_NO_FUSED_BNECK=1."""
import math, sys, torch
sys.path.insert(0, ".")
from ccnet_amd import ops
from ccnet_amd._lib import lib, sl

DEV = "cuda"
def bench(fn, n=60):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for B, C, H in [(64, 128, 40), (32, 128, 40), (32, 64, 80), (64, 64, 80), (16, 128, 80), (64, 256, 20), (32, 256, 20)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, H, C, generator=g).to(DEV).to(torch.bfloat16)
    w1 = torch.randn(C, C, 1, 1, generator=g) / math.sqrt(C); b1 = torch.randn(C, generator=g) * 0.1
    w2 = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C); b2 = torch.randn(C, generator=g) * 0.1
    pk = ops.PackedBneck(w1, b1, w2, b2, DEV)
    pc, ph = ops.PackedConv(w1, b1, 1, DEV), ops.PackedHalo(w2, b2, DEV)
    out = torch.empty_like(x); hid = torch.empty_like(x)
    f = lambda: ops.bneck_fused(sl(x), pk, B, H, H, True, "silu", sl(out))
    def u():
        ops.conv2d([sl(x)], pc, B, H, H, 1, 0, sl(hid), "silu")
        ops.conv_halo([sl(hid)], ph, 0, B, H, H, sl(out), "silu", sl(x))
    tf, tu = bench(f), bench(u)
    tf2, tu2 = bench(f), bench(u)
    print("B %3d C %3d %dx%d: fused %.1f / %.1f us   two launches %.1f / %.1f us" % (B, C, H, H, tf, tf2, tu, tu2), flush=True)
    L = lib()
    if hasattr(L, "ccnet_debug_bneck_stamps"):   # -DBK_STAMP build: ticks (10 ns) block 0 spends per phase, per launch
        import ctypes
        buf = (ctypes.c_ulonglong * 8)()
        L.ccnet_debug_bneck_stamps(buf, 1)
        for _ in range(20): f()
        L.ccnet_debug_bneck_stamps(buf, 1)
        names = ["wait x", "phase 1", "h epilogue", "phase 2", "out epilogue"]
        print("     block 0 per launch: " + "  ".join("%s %.2f us" % (n, buf[i] / 20 * 0.01) for i, n in enumerate(names)), flush=True)
