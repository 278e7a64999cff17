"""Which tap differs between two forwards of the same input (GPU box)?  python scripts/det_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ccnet_amd import synth
net = bench.build_model("net", "bf16").to("cuda").set_precision("bf16")
B, S = 64, 640
x = torch.cat([synth.synth_images(1, S, S, seed=1000 + i) for i in range(B)], 0).contiguous().cuda()
net(x); torch.cuda.synchronize()
plan = net.plan_for(B, S, S, x.device)[0]
snap = None
for it in range(6):
    net(x); torch.cuda.synchronize()
    cur = {k: t.buf[..., t.coff:t.coff + t.C].clone() for k, t in plan.taps.items()}
    if snap is not None:
        bad = [(k, int((cur[k] != snap[k]).sum())) for k in cur if not torch.equal(cur[k], snap[k])]
        print("run", it, "differs:", bad[:6], flush=True)
    snap = cur
