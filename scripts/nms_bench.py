"""SURVEY 8(d) NMS-only microbench (GPU box): n candidates/image x 16 images, log-normal boxes, 6 classes."""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd.utils import bbox
for n in (1000, 10000, 40000):
    rng = np.random.RandomState(0)
    B = 16
    cand = np.zeros((B, n, 7), np.float32)
    c = rng.uniform(0, 1, (B, n, 2)); wh = np.exp(rng.normal(math.log(0.05), 0.5, (B, n, 2)))
    cand[..., 0:2] = c - wh / 2; cand[..., 2:4] = c + wh / 2
    cand[..., 4] = rng.uniform(0, 1, (B, n)); cand[..., 5] = 1.0; cand[..., 6] = rng.randint(0, 6, (B, n))
    ct = torch.from_numpy(cand).cuda(); counts = torch.full((B,), n, dtype=torch.int32, device="cuda")
    hw = torch.ones(B, 2, dtype=torch.int32, device="cuda")
    bbox._nms(ct, counts, n, 0.6, hw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        det, keep, nk, _ = bbox._nms(ct, counts, n, 0.6, hw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    pairs = B * 6 * (n / 6) * (n / 6 - 1) / 2
    print("n=%6d x16: %8.3f ms  kept/img %.0f  %.2f G same-class pairs/s" % (n, ms, float(nk.float().mean()), pairs / ms / 1e6), flush=True)
