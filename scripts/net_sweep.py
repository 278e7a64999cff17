"""Whole-network parity sweep (GPU box): every net at odd batch sizes and non-square inputs, fp32 HIP path against the CPU
oracle (gate 1e-3 relative per level) and bf16 against fp32 (rms), plus decode + post_process against the oracle on the
device logits (bit-exact).  python scripts/net_sweep.py [quick]"""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import synth
from ccnet_amd.utils import bbox
from oracle import ccnet_oracle as orc

CASES = [("net", 3, 96, 160), ("net", 1, 160, 96), ("net", 5, 64, 64), ("net_old", 2, 128, 96), ("net_deploy", 3, 128, 224),
         ("net_deploy", 7, 64, 96), ("net_amff", 2, 96, 128), ("net_amff", 3, 64, 160), ("net_small_target", 2, 128, 192),
         ("net_small_target", 3, 96, 64), ("net", 2, 256, 512), ("net_deploy", 1, 320, 256)]


def run(cases, verbose=True):
    worst = 0.0
    for kind, B, H, W in cases:
        net = importlib.import_module("ccnet_amd.network." + kind).Net(0.33, 0.5, 6, "silu")
        sd = synth.synth_state_dict(net.state_dict(), seed=0, bn_calib=kind)
        net.load_state_dict(sd)
        net = net.to("cuda").eval()
        x = synth.synth_images(B, H, W, seed=B * 1000 + H + W)
        ref = orc.forward(kind, sd, x, 0.33, 0.5, "silu")
        o32 = [o.clone() for o in net.set_precision("fp32")(x.cuda())]
        o16 = [o.clone() for o in net.set_precision("bf16")(x.cuda())]
        torch.cuda.synchronize()
        e32 = max(float((o.cpu() - r).abs().max() / r.abs().max()) for o, r in zip(o32, ref))
        rms = max(float(((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())) for a, b in zip(o16, o32))
        dec = bbox.decode_outputs(o32, [H, W])
        rdec = orc.decode_outputs([o.cpu() for o in o32], [H, W])
        edec = float((dec.cpu() - rdec).abs().max() / rdec.abs().max())
        info = [[i, 2 * W + i, H + 3 * i] for i in range(B)]
        got = bbox.post_process(dec, 6, [H, W], info, 0.3, 0.6)
        exp = orc.post_process(dec.cpu(), 6, [H, W], info, 0.3, 0.6)
        same = all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(got, exp))
        if verbose:
            print("%-17s B%d %3dx%-3d fp32 vs oracle %.1e  bf16 vs fp32 rms %.3f  decode %.1e  post_process %s  (%d dets)" % (
                kind, B, H, W, e32, rms, edec, "bit-exact" if same else "DIFFERS", sum(g.shape[0] for g in got)), flush=True)
        assert e32 < 1e-3 and rms < 0.08 and edec < 3e-6 and same, (kind, B, H, W, e32, rms, edec, same)
        worst = max(worst, e32)
    return worst


if __name__ == "__main__":
    run(CASES[:4] if len(sys.argv) > 1 else CASES)
    print("sweep ok")
