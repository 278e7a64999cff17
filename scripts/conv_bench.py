"""Micro-benchmark of single conv shapes through the C ABI (debug/perf aid; run on the GPU box)."""
import math, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops

SHAPES = {  # name: (B, H, W, cin, cout, k, stride, pad)
    "c3_128_80": (64, 80, 80, 128, 128, 3, 1, 1),
    "c3_128_80_n256": (64, 80, 80, 128, 256, 3, 1, 1),
    "c3_64_160": (64, 160, 160, 64, 64, 3, 1, 1),
    "s2d_32_320": (64, 320, 320, 32, 64, 6, 2, 2),
    "c1_64_160": (64, 160, 160, 64, 64, 1, 1, 0),
    "c1_128_80": (64, 80, 80, 128, 128, 1, 1, 0),
    "c1_256_40_n256": (64, 40, 40, 256, 256, 1, 1, 0),
    "lin_256_768": (1, 1, 25600, 256, 768, 1, 1, 0),
    "lin_256_1024": (1, 1, 25600, 256, 1024, 1, 1, 0),
    "lin_1024_256": (1, 1, 25600, 1024, 256, 1, 1, 0),
    "c3_512_20": (64, 20, 20, 512, 512, 3, 1, 1),
    "c3_1024_512_20": (64, 20, 20, 1024, 512, 3, 1, 1),
    "c3_128_20": (64, 20, 20, 128, 128, 3, 1, 1),
    "c1_512_20": (64, 20, 20, 512, 512, 1, 1, 0),
    "c1_1024_512_20": (64, 20, 20, 1024, 512, 1, 1, 0),
}
names = sys.argv[1:] or list(SHAPES)
for name in names:
    B, H, W, cin, cout, k, s, pad = SHAPES[name]
    g = torch.Generator().manual_seed(0)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    pc = ops.PackedConv(w, torch.zeros(cout), 1, "cuda")
    x = torch.randn(B, H, W, cin, device="cuda").to(torch.bfloat16)
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    out = torch.empty(B, Ho, Wo, cout, device="cuda", dtype=torch.bfloat16)
    def run():
        ops.conv2d([ops.sl(x)], pc, B, H, W, s, pad, ops.sl(out), "silu")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * Ho * Wo * cout * cin * k * k
    by = (B * H * W * cin + B * Ho * Wo * cout + cout * cin * k * k) * 2
    print("%-16s %8.3f ms  %7.1f TF/s  %7.1f GB/s" % (name, ms, fl / ms / 1e9, by / ms / 1e6), flush=True)
    from ccnet_amd import _lib
    if hasattr(_lib.lib(), "ccnet_debug_stamps"):  # diagnostic build (-DCC_STAMP): share of wave time per main-loop phase
        import ctypes
        st = (ctypes.c_ulonglong * 8)()
        _lib.lib().ccnet_debug_stamps(st, 1)
        tot = float(sum(st[:5])) or 1.0
        print("   phases: gather %.1f%%  epilogue %.1f%%  lds+mfma %.1f%%  wait-dma %.1f%%  barrier %.1f%%" % tuple(100.0 * st[i] / tot for i in range(5)))
