"""Stem kernel alone (GPU box): 64 x 3 x 640 x 640 fp32 -> 320 x 320 x 32 bf16, SiLU.  python scripts/stem_bench.py [iters]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, S, cout = 64, 640, 32
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, S, S, generator=g).cuda()
w6 = torch.randn(cout, 3, 6, 6, generator=g) / math.sqrt(108)
b = torch.randn(cout, generator=g) * 0.1
wp, bp = ops.pack_stem(w6, b, "cuda")
out = torch.zeros(B, S // 2, S // 2, cout, dtype=torch.bfloat16, device="cuda")
run = lambda: ops.stem_conv(x, wp, bp, cout, "silu", ops.sl(out), 1)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
by = x.numel() * 4 + out.numel() * 2
print("stem 64x640x640: %.4f ms, %.0f GB/s of %.0f MB" % (ms, by / ms / 1e6, by / 1e6), flush=True)
ref = torch.nn.functional.silu(torch.nn.functional.conv2d(x[:2], w6.cuda(), b.cuda(), 2, 2))
err = float((out[:2].float().permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max())
print("max rel err vs torch (2 images): %.2e" % err)
