"""Cycle stamps of the fused CSP kernel (library built with EXTRA_csp_fused=-DCSP_STAMP): share of wave 0's time per phase."""
import ctypes as C, os, sys
os.environ["CCNET_FUSED_CSP_ALL"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops, _lib
from ccnet_amd._lib import BF16
from ccnet_amd.network import graph
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_ops_gpu as tg
names = ["load+sync", "mm1", "epi1+sync", "stage2+sync", "stage3+sync", "stage4"]
for h, segs, up, sc, B, H, W in ((32, (64,), False, True, 64, 160, 160), (64, (128,), False, True, 64, 80, 80), (64, (128, 128), True, False, 64, 80, 80)):
    gen = torch.Generator().manual_seed(1)
    sd = tg._csp_state("blk", sum(segs), h, gen)
    comp = graph.Compiler(sd, {}, B, H, W, BF16, torch.device("cuda"), "silu")
    ins = []
    for i, c in enumerate(segs):
        hh, ww = (H // 2, W // 2) if (up and i == 0) else (H, W)
        t = graph.T(torch.randn(B, hh, ww, c, device="cuda").to(torch.bfloat16), 0, c, hh, ww)
        ins.append(t.up() if (up and i == 0) else t)
    out8 = (C.c_ulonglong * 8)()
    comp.csp("blk", ins, 1, sc)
    _lib.lib().ccnet_debug_csp_stamps(out8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); comp.csp("blk", ins, 1, sc); e1.record(); torch.cuda.synchronize()
    _lib.lib().ccnet_debug_csp_stamps(out8)
    tot = sum(out8[:6])
    print("h=%d segs=%s: %.1f us; block 0: %.0f memtime ticks/launch: " % (h, segs, e0.elapsed_time(e1) * 1e3, tot) +
          "  ".join("%s %.1f%%" % (n, 100.0 * out8[i] / tot) for i, n in enumerate(names)), flush=True)
