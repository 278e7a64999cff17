import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
from ccnet_amd import _lib
from ccnet_amd.utils import bbox
g = torch.Generator().manual_seed(0)
B, A, n = 64, 8400, 100
cand = torch.zeros(B, A, 7)
c = torch.rand(B, n, 2, generator=g) * 0.6 + 0.2
wh = torch.rand(B, n, 2, generator=g) * 0.1 + 0.02
cand[:, :n, 0:2] = c - wh / 2; cand[:, :n, 2:4] = c + wh / 2
cand[:, :n, 4] = torch.rand(B, n, generator=g); cand[:, :n, 5] = 1.0
cand[:, :n, 6] = torch.randint(0, 6, (B, n), generator=g).float()
cd = cand.cuda(); ct = torch.full((B,), n, dtype=torch.int32).cuda(); hw = torch.ones(B, 2, dtype=torch.int32).cuda()
for _ in range(3): bbox._nms(cd, ct, 128, 0.6, hw)
torch.cuda.synchronize()
st = (ctypes.c_ulonglong * 8)()
_lib.lib().ccnet_debug_nms_stamps(st)
N = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): bbox._nms(cd, ct, 128, 0.6, hw)
e1.record(); torch.cuda.synchronize()
_lib.lib().ccnet_debug_nms_stamps(st)
print("per call %.1f us (incl. host); phases (100 MHz ticks -> us): " % (e0.elapsed_time(e1) / N * 1e3), [round(st[i] / N / 100.0, 2) for i in range(5)])
