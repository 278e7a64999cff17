"""Fused Transformer, full depth: is a launch with COLD caches (L2 / Infinity Cache flushed by a 1 GB write, fresh parameter
copies) different from warm launches?  Prints, per launch, the number of elements that differ from the majority.  (GPU box)
   python scripts/tf_cold_probe.py [enc,dec]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops, synth
ne, nd = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "6,6").split(","))
net = importlib.import_module("ccnet_amd.network.net").Net(0.33, 0.5, 6, "silu")
sd = synth.synth_state_dict(net.state_dict(), seed=0, bn_calib="net")
S, N = 64, 400
src = (torch.randn(S * N, 256, device="cuda") * 0.7).to(torch.bfloat16)
wst, pvec, fpw = ops.pack_transformer(dict(sd), "dark5.2.transformer", ne, nd, "cuda")
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
outs, kinds = [], []
for i in range(12):
    cold = i % 3 == 0
    if cold:
        big.fill_(float(i))          # 1 GB: everything the kernel reads is evicted from L2 and the Infinity Cache
    out = torch.full_like(src, 7.0)
    torch.cuda.synchronize()
    ops.transformer_fused(src, 256, out, 256, S, N, wst, fpw, pvec, ne, nd)
    torch.cuda.synchronize()
    outs.append(out.float().cpu()); kinds.append("cold" if cold else "warm")
st = torch.stack(outs)
med = st.median(dim=0).values
for i, k in enumerate(kinds):
    bad = (st[i] != med)
    toks = sorted(set((bad.any(dim=1).nonzero().flatten() // N).tolist()))
    print("launch %2d %s: %7d elements differ from the majority; tokens %s" % (i, k, int(bad.sum()), (toks[:3] + ["..."] + toks[-2:]) if len(toks) > 6 else toks))
