"""Micro-benchmark of the fused transformer kernel alone (GPU box): S=64 tokens x N=400 positions, s-width weights."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
net = importlib.import_module("ccnet_amd.network.net").Net(0.33, 0.5, 6, "silu")
sd = synth.synth_state_dict(net.state_dict(), seed=0, bn_calib="net")
wst, pvec, fpw = ops.pack_transformer(sd, "dark5.2.transformer", 6, 6, "cuda")
src = (torch.randn(S * N, 256, device="cuda") * 0.7).to(torch.bfloat16)
out = torch.empty_like(src)
run = lambda: ops.transformer_fused(src, 256, out, 256, S, N, wst, fpw, pvec, 6, 6)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print("fused transformer S=%d N=%d: %.3f ms  (out rms %.3f)" % (S, N, e0.elapsed_time(e1) / 10, float(out.float().pow(2).mean().sqrt())), flush=True)

from ccnet_amd import _lib
if hasattr(_lib.lib(), "ccnet_debug_tf_stamps"):  # diagnostic build (-DTF_STAMP)
    import ctypes
    st = (ctypes.c_ulonglong * 8)()
    _lib.lib().ccnet_debug_tf_stamps(st)
    tot = float(sum(st[:5])) or 1.0
    print("   phases: gemm %.1f%%  epilogue %.1f%%  barrier %.1f%%  attention core %.1f%%  layernorm %.1f%%" % tuple(100.0 * st[i] / tot for i in range(5)))
