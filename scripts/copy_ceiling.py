"""What a read+write stream reaches on this GPU at the sizes of the thin layers (torch's device copy as the yardstick)."""
import torch

for mb in (26, 52, 105, 210, 420, 840):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
    y = torch.empty_like(x)
    for _ in range(5):
        y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("copy %4d MB in + %4d MB out: %.4f ms  %.0f GB/s (read + write)" % (mb, mb, ms, 2 * mb / ms), flush=True)
