"""Per-tap error of the HIP path vs the CPU oracle (debug aid; run on the GPU box)."""
import importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import synth
from oracle import ccnet_oracle as orc

kind = sys.argv[1] if len(sys.argv) > 1 else "net_deploy"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
net = importlib.import_module("ccnet_amd.network." + kind).Net(0.33, 0.5, 6)
sd = synth.synth_state_dict(net.state_dict(), 0, kind)
net.load_state_dict(sd)
net.to("cuda")
x = synth.synth_images(B, 224, 224, 0)
taps = {}
ref = orc.forward(kind, sd, x, 0.33, 0.5, "silu", taps)
for prec in ("fp32", "bf16"):
    net.set_precision(prec)
    outs = net(x.cuda())
    torch.cuda.synchronize()
    plan = net.plan_for(B, 224, 224, x.cuda().device)[0]
    print("==", kind, prec)
    for name, t in plan.taps.items():
        if name not in taps:
            continue
        got = t.buf[..., t.coff:t.coff + t.C].float().cpu().permute(0, 3, 1, 2)
        r = taps[name]
        print("  %-10s rel-max %.3e  rms-rel %.3e  ref std %.3f" % (name, float((got - r).abs().max() / r.abs().max()),
              float((got - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()), float(r.std())))
    for k, (o, r) in enumerate(zip(outs, ref)):
        o = o.float().cpu()
        print("  head%d      rel-max %.3e  rms-rel %.3e" % (k, float((o - r).abs().max() / r.abs().max()),
              float((o - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt())))
