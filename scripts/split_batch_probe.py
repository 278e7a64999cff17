"""Probe: does running two half-batches as independent chains on two streams beat one full batch on one stream?
(Dependent launches leave ~5 us gaps and small-M layers leave CUs idle; a second chain can fill both.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
from ccnet_amd import synth

kind = sys.argv[1] if len(sys.argv) > 1 else "net_deploy"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = 640
mod = importlib.import_module("ccnet_amd.network." + kind)


def mk():
    net = mod.Net(0.33, 0.5, 6, "silu")
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0, bn_calib=kind))
    net = net.to("cuda").set_precision("bf16")
    net.alias_outputs = True
    return net

x = synth.synth_images(8, S, S, seed=1).repeat(B // 8, 1, 1, 1).contiguous().cuda()
full = mk()
h = B // 2
a, b = mk(), mk()
xa, xb = x[:h].contiguous(), x[h:].contiguous()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run_full():
    full(x)


def run_split():
    with torch.cuda.stream(sa):
        a(xa)
    with torch.cuda.stream(sb):
        b(xb)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

for rep in range(2):
    print(kind, "B", B, "full %.3f ms   split 2x%d %.3f ms" % (timeit(run_full), h, timeit(run_split)))
