"""Fused transformer vs per-op path (GPU box): same bf16 model, batch sizes 1, 2, 5, 64."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import synth

def run(B, S, fused):
    os.environ["CCNET_NO_FUSED_TRANSFORMER"] = "0" if fused else "1"
    net = importlib.import_module("ccnet_amd.network.net").Net(0.33, 0.5, 6, "silu")
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0, bn_calib="net"))
    net = net.to("cuda").eval().set_precision("bf16")
    x = synth.synth_images(B, S, S, seed=1).to("cuda")
    outs = [o.float().clone() for o in net(x)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        net(x)
    torch.cuda.synchronize()
    return outs, (time.perf_counter() - t0) / 5 * 1e3

for B, S in ((1, 224), (2, 224), (5, 224), (64, 224), (64, 640)):
    a, ta = run(B, S, True)
    b, tb = run(B, S, False)
    errs = []
    for u, v in zip(a, b):
        errs.append("%.2e/%.2e" % (float((u - v).abs().max() / v.abs().max()), float((u - v).pow(2).mean().sqrt() / v.pow(2).mean().sqrt())))
    print("B=%d %d^2: max/rms rel diff per level %s   forward %.3f ms fused vs %.3f ms per-op" % (B, S, " ".join(errs), ta, tb), flush=True)
