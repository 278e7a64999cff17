"""CBAM / ECA timing at the three maps of net 640x640 bs64 (bf16): per-op milliseconds by HIP events; run under
rocprofv3 --kernel-trace --stats for the per-kernel split.  python scripts/gate_bench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
g = torch.Generator().manual_seed(0)
B = 64
tot = 0.0
for H, Cc in ((80, 128), (40, 256), (20, 512)):
    x = torch.randn(B, H, H, Cc, generator=g).to(torch.bfloat16).to(dev)
    cr = Cc // 16
    fc1 = (torch.randn(cr, Cc, generator=g) * 0.2).to(dev)
    fc2 = (torch.randn(Cc, cr, generator=g) * 0.5).to(dev)
    w7 = (torch.randn(2, 7, 7, generator=g) * 0.3).to(dev)
    w3 = torch.randn(3, generator=g).to(dev)
    scratch = torch.zeros(ops.gate_scratch_floats(B, H, H, Cc), dtype=torch.float32, device=dev)
    out = torch.empty_like(x)
    for name, fn in (("cbam", lambda: ops.cbam(ops.sl(x), 1, B, H, H, fc1, fc2, w7, scratch, ops.sl(out))),
                     ("eca", lambda: ops.eca(ops.sl(x), 1, B, H, H, w3, scratch, ops.sl(out)))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        if name == "cbam":
            tot += ms
        print("%s %dx%d C%d: %.4f ms  (x bytes %.1f MB)" % (name, H, H, Cc, ms, x.numel() * 2 / 1e6))
print("cbam total %.4f ms" % tot)
