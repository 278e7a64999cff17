"""Which elements differ run to run in a TF_STOP build of the fused Transformer (GPU box): rows, columns, values."""
import importlib, os, sys
from collections import Counter
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops, synth
net = importlib.import_module("ccnet_amd.network.net").Net(0.33, 0.5, 6, "silu")
sd = synth.synth_state_dict(net.state_dict(), seed=0, bn_calib="net")
S, N = 64, 400
src = (torch.randn(S * N, 256, device="cuda") * 0.7).to(torch.bfloat16)
wst, pvec, fpw = ops.pack_transformer(dict(sd), "dark5.2.transformer", 0, 1, "cuda")
outs = []
for _ in range(8):
    out = torch.full_like(src, 7.0)
    ops.transformer_fused(src, 256, out, 256, S, N, wst, fpw, pvec, 0, 1)
    torch.cuda.synchronize()
    outs.append(out.float().cpu())
# majority value per element = the right one; report deviations
st = torch.stack(outs)                      # (8, rows, 256)
med = st.median(dim=0).values
cols, lanes, waves = Counter(), Counter(), Counter()
shown = 0
for r in range(8):
    bad = (st[r] != med).nonzero()
    for row, col in bad.tolist():
        cols[col % 64] += 1; waves[col // 64] += 1; lanes[row // N] += 1
        if shown < 40:
            print("run %d token %2d pos %3d col %3d (wave %d, q %d, elem %d): got %9.4f majority %9.4f  all runs %s" % (
                r, row // N, row % N, col, col // 64, (col % 64) // 8, col % 8, st[r, row, col], med[row, col],
                [round(float(v), 3) for v in st[:, row, col]]))
            shown += 1
print("column-within-wave histogram:", sorted(cols.items()))
print("wave histogram:", sorted(waves.items()))
print("token histogram:", sorted(lanes.items()))
