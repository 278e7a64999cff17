"""Per-layer timing of ccnet_conv_halo against ccnet_conv2d's halo-patch kernel (GPU box):  python scripts/halo_bench.py"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops

def ev(fn, reps=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

CASES = [  # B, H, W, Cin, Cout, s2d, label
    (64, 80, 80, 128, 128, False, "head 3x3 128->128 @80"),
    (64, 80, 80, 128, 256, False, "head 3x3 128->256 @80"),
    (64, 40, 40, 128, 128, False, "3x3 128->128 @40"),
    (64, 20, 20, 256, 256, False, "3x3 256->256 @20"),
    (64, 20, 20, 1024, 512, False, "MobileViT conv4 1024->512 @20"),
    (64, 80, 80, 64, 64, False, "3x3 64->64 @80"),
    (64, 320, 320, 32, 64, True, "dark2 S2D 32->64 @160"),
    (64, 160, 160, 64, 128, True, "dark3 S2D 64->128 @80"),
    (64, 80, 80, 128, 256, True, "dark4 S2D 128->256 @40"),
    (64, 40, 40, 256, 512, True, "dark5 S2D 256->512 @20"),
    (32, 40, 40, 128, 128, False, "deploy bs32 3x3 128->128 @40"),
    (32, 20, 20, 256, 256, False, "deploy bs32 3x3 256->256 @20"),
]
only = sys.argv[1:]
for B, H, W, C, cout, s2d, label in CASES:
    if only and not any(o in label for o in only):
        continue
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).cuda()
    cv = 4 * C if s2d else C
    w = torch.randn(cout, cv, 3, 3, generator=g) / math.sqrt(cv * 9)
    b = torch.randn(cout, generator=g) * 0.1
    pk = ops.PackedHalo(w, b, "cuda")
    pc = ops.PackedConv(ops.s2d_to_6x6(w) if s2d else w, b, 1, "cuda")
    Ho, Wo = (H // 2, W // 2) if s2d else (H, W)
    out = torch.zeros(B, Ho, Wo, cout, dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * B * Ho * Wo * cout * 9 * cv
    res = {}
    for act in ("silu", "none"):
        res["halo " + act] = ev(lambda: ops.conv_halo([ops.sl(x)], pk, s2d, B, H, W, ops.sl(out), act))
        res["patch " + act] = ev(lambda: ops.conv2d([ops.sl(x)], pc, B, H, W, 2 if s2d else 1, 2 if s2d else 1, ops.sl(out), act))
    print("%-34s " % label + "  ".join("%s %.3f ms %4.0f TF/s" % (k, v, fl / v / 1e9) for k, v in res.items()), flush=True)
