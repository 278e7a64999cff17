"""Fused CSP kernel alone (GPU box): the three big-map CSP layers of `net` at bs64, fused against the five launches.
python scripts/csp_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CCNET_FUSED_CSP_ALL"] = "1"
from ccnet_amd import ops
from ccnet_amd._lib import BF16
from ccnet_amd.network import graph
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_ops_gpu as tg

for name, h, segs, up, sc, B, H, W in (("dark2.1", 32, (64,), False, True, 64, 160, 160), ("dark3.1", 64, (128,), False, True, 64, 80, 80),
                                       ("C3_p3", 64, (128, 128), True, False, 64, 80, 80)):
    gen = torch.Generator().manual_seed(1)
    sd = tg._csp_state("blk", sum(segs), h, gen)
    res = []
    for fused in (True, False):
        os.environ["CCNET_NO_FUSED_CSP"] = "0" if fused else "1"
        comp = graph.Compiler(sd, {}, B, H, W, BF16, torch.device("cuda"), "silu")
        ins = []
        for i, c in enumerate(segs):
            hh, ww = (H // 2, W // 2) if (up and i == 0) else (H, W)
            t = graph.T((torch.randn(B, hh, ww, c, device="cuda") * 0.8).to(torch.bfloat16), 0, c, hh, ww)
            ins.append(t.up() if (up and i == 0) else t)
        with ops.Recorder() as rec:
            comp.csp("blk", ins, 1, sc)
        plan = ops.Plan(rec.handle, comp.keep)
        for _ in range(3):
            plan.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print("%-8s hidden %3d: fused %7.1f us   five launches %7.1f us" % (name, h, res[0], res[1]), flush=True)
