"""Micro-benchmark of the on-device image preparation (csrc/preprocess.hip): n source images already resident in HBM ->
(n, 3, H, W) fp32.  Prints achieved GB/s on the algorithmic bytes (source bytes read once + fp32 planes written once) and,
with --cpu, the oracle's (numpy restatement of Pillow + the reference's normalisation) rate on a few images.

    python scripts/preprocess_bench.py [--n 64] [--src 200 200] [--out 640 640] [--letterbox] [--cpu]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--src", type=int, nargs=2, default=[200, 200])
    ap.add_argument("--out", type=int, nargs=2, default=[640, 640])
    ap.add_argument("--letterbox", action="store_true")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    from ccnet_amd.utils import preprocess as pp
    ih, iw = a.src
    h, w = a.out
    rng = np.random.default_rng(0)
    host = rng.integers(0, 256, (a.n, ih, iw, 3), dtype=np.uint8)
    src = torch.from_numpy(host).cuda()
    out = torch.empty((a.n, 3, h, w), dtype=torch.float32, device="cuda")
    for _ in range(5):
        pp.preprocess_group(src, out, (h, w), a.letterbox)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        pp.preprocess_group(src, out, (h, w), a.letterbox)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    nbytes = a.n * (ih * iw * 3 + 3 * h * w * 4)
    line = {"n": a.n, "src": [ih, iw], "out": [h, w], "letterbox": a.letterbox, "ms": round(ms, 4), "images_per_s": round(a.n / ms * 1e3, 1),
            "algorithmic_GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000.0, 4)}
    t0 = time.perf_counter()
    for _ in range(3):
        pp.preprocess(host, (h, w), "cuda", a.letterbox)
    torch.cuda.synchronize()
    line["with_h2d_images_per_s"] = round(3 * a.n / (time.perf_counter() - t0), 1)
    if a.cpu:
        from oracle import preprocess_oracle as po
        k = min(a.n, 4)
        t0 = time.perf_counter()
        for i in range(k):
            (po.preprocess_letterbox if a.letterbox else po.preprocess_eval)(host[i], (h, w))
        line["oracle_images_per_s"] = round(k / (time.perf_counter() - t0), 2)
        try:
            from PIL import Image
            t0 = time.perf_counter()
            for i in range(a.n):
                d = np.array(Image.fromarray(host[i], "RGB").resize((w, h), Image.BICUBIC), dtype="float32")
                d /= 255.0
                d -= po.MEAN
                d /= po.STD
                np.transpose(d, (2, 0, 1))
            line["pillow_numpy_1core_images_per_s"] = round(a.n / (time.perf_counter() - t0), 1)
        except ImportError:
            pass
    print(line)


if __name__ == "__main__":
    main()
