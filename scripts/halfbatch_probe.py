"""Does a forward run faster as two half batches on two streams (one kernel's tail rounds filled by the other chain's kernels)
than as one batch?  Timing probe only: for `net` the Transformer attends over the batch axis, so two halves are NOT the same
function there -- read its numbers as a bound on what the conv stages could gain.
    python scripts/halfbatch_probe.py [kind] [batch] [size]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from ccnet_amd import synth


def build(kind, dev):
    net = importlib.import_module("ccnet_amd.network." + kind).Net(0.33, 0.5, 6, "silu")
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0, bn_calib=kind))
    return net.to(dev).set_precision("bf16").eval()


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "net_deploy"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 640
    dev = torch.device("cuda:0")
    x = torch.cat([synth.synth_images(1, S, S, seed=1000 + i) for i in range(B)], 0).contiguous().to(dev)
    whole = build(kind, dev)
    halves = [build(kind, dev), build(kind, dev)]
    xs = [x[:B // 2].contiguous(), x[B // 2:].contiguous()]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def one():
        whole(x)

    def two():
        cur = torch.cuda.current_stream()
        for h, xh, st in zip(halves, xs, streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                h(xh)
        for st in streams:
            cur.wait_stream(st)

    def serial_halves():
        for h, xh in zip(halves, xs):
            h(xh)

    with torch.no_grad():
        for name, fn in (("one batch of %d" % B, one), ("two halves, two streams", two), ("two halves, one stream", serial_halves),
                         ("one batch of %d" % B, one), ("two halves, two streams", two)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            print("%-28s %.3f ms per %d images" % (name, (time.perf_counter() - t0) / 20 * 1e3, B), flush=True)


main()
