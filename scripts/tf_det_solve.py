"""Explain the wrong LayerNorm elements of a TF_STOP=2 build: which operand (gamma, beta, x) would have to be wrong?  (GPU box)
   CCNET_HIP_LIB=<stop build> python scripts/tf_det_solve.py dump <file>   -> saves 8 launches of the kernel
   python scripts/tf_det_solve.py solve <stop1 file> <stop2 file>"""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S, N = 64, 400
if sys.argv[1] == "dump":
    from ccnet_amd import ops, synth
    net = importlib.import_module("ccnet_amd.network.net").Net(0.33, 0.5, 6, "silu")
    sd = synth.synth_state_dict(net.state_dict(), seed=0, bn_calib="net")
    torch.manual_seed(3)
    src = (torch.randn(S * N, 256, device="cuda") * 0.7).to(torch.bfloat16)
    wst, pvec, fpw = ops.pack_transformer(dict(sd), "dark5.2.transformer", 0, 1, "cuda")
    outs = []
    for _ in range(8):
        out = torch.full_like(src, 7.0)
        ops.transformer_fused(src, 256, out, 256, S, N, wst, fpw, pvec, 0, 1)
        torch.cuda.synchronize()
        outs.append(out.float().cpu().numpy())
    g = sd["dark5.2.transformer.decoder.layers.0.norm1.weight"].float().numpy()
    b = sd["dark5.2.transformer.decoder.layers.0.norm1.bias"].float().numpy()
    np.savez(sys.argv[2], outs=np.stack(outs), g=g, b=b, src=src.float().cpu().numpy())
else:
    a, c = np.load(sys.argv[2]), np.load(sys.argv[3])
    x = a["outs"]; assert (x == x[0]).all(), "the pre-LayerNorm dump is not deterministic"
    x = x[0].astype(np.float64)
    y = c["outs"]; g, b = c["g"].astype(np.float64), c["b"].astype(np.float64)
    med = np.median(y, axis=0)
    mean = x.mean(1, keepdims=True); rs = 1.0 / np.sqrt(x.var(1, keepdims=True) + 1e-5)
    ref = (x - mean) * rs * g + b
    print("majority vs fp64 LayerNorm of the dumped input: max |diff| %.4f" % np.abs(med - ref).max())
    for r in range(y.shape[0]):
        bad = np.argwhere(y[r] != med)
        groups = {}
        for row, col in bad:
            groups.setdefault((row % N, col), []).append(row)
        for (pos, col), rows in sorted(groups.items()):
            gs, bs, xs = [], [], []
            for row in rows:
                yy = y[r, row, col]; xn = (x[row, col] - mean[row, 0]) * rs[row, 0]
                gs.append((yy - b[col]) / xn); bs.append(yy - xn * g[col]); xs.append((yy - b[col]) / g[col] / rs[row, 0] + mean[row, 0])
            gs, bs = np.array(gs), np.array(bs)
            # does the implied x equal the row's x at another column?
            hits = []
            for row, xv in zip(rows, xs):
                j = int(np.argmin(np.abs(x[row] - xv))); hits.append((j, round(float(x[row, j] - xv), 3)))
            src = c["src"].astype(np.float64)
            print("        mean of the row: %s" % [round(float(mean[row, 0]), 3) for row in rows][:6])
            print("        implied x - true x: %s ; implied x - src (pre-attention value): %s ; got - x (unnormalised?): %s" % (
                [round(float(xv - x[row, col]), 3) for row, xv in zip(rows, xs)][:6], [round(float(xv - src[row, col]), 3) for row, xv in zip(rows, xs)][:6],
                [round(float(y[r, row, col] - x[row, col]), 3) for row in rows][:6]))
            print("run %d pos %3d col %3d (%2d tokens): implied gamma %.3f +- %.3f (true %.3f)  implied beta %.3f +- %.3f (true %.3f)  nearest x column %s" % (
                r, pos, col, len(rows), gs.mean(), gs.std(), g[col], bs.mean(), bs.std(), b[col], hits[:4]))
            near_g = int(np.argmin(np.abs(g - gs.mean()))); near_b = int(np.argmin(np.abs(b - bs.mean())))
            print("        gamma nearest to implied: col %d (%.3f); beta nearest to implied: col %d (%.3f)" % (near_g, g[near_g], near_b, b[near_b]))
