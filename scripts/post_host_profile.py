"""Where post_process' time goes on the host (run on the GPU box): python scripts/post_host_profile.py [--batch 64]
Times, per call and with the GPU idle at the start: submission (filter + NMS launches, buffers, the copy), the wait for the copy,
the split into per-image arrays."""
import argparse, sys, time, importlib
import torch
sys.path.insert(0, ".")
from ccnet_amd import synth
from ccnet_amd.utils import bbox

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--kind", default="net")
a = ap.parse_args()
net = importlib.import_module("ccnet_amd.network." + a.kind).Net(0.33, 0.5, 6, "silu")
net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0, bn_calib=a.kind))
net = net.to("cuda").eval().set_precision("bf16")
x = synth.synth_images(a.batch, 640, 640, seed=0).to("cuda")
info = [["im%d" % i, 640, 640] for i in range(a.batch)]
outs = net(x)
pred = bbox.decode_outputs(outs, (640, 640))
for _ in range(5):
    bbox.post_process(pred, 6, (640, 640), info, 0.5, 0.6)
n = 200
t_sub = t_wait = t_split = 0.0
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p = bbox.post_process_async(pred, 6, (640, 640), info, 0.5, 0.6)
    t1 = time.perf_counter()
    p.event.synchronize()
    t2 = time.perf_counter()
    r = p.result()
    t3 = time.perf_counter()
    t_sub += t1 - t0; t_wait += t2 - t1; t_split += t3 - t2
print("per call: submit %.1f us   wait for the copy %.1f us   split %.1f us   (detections %d)" % (t_sub / n * 1e6, t_wait / n * 1e6, t_split / n * 1e6, sum(len(d) for d in r)))
