import sys, time, torch
sys.path.insert(0, "/root/repo")
from ccnet_amd import synth
from ccnet_amd.network.net import Net
from ccnet_amd.utils import bbox
for (d, w, nc, B, S) in [(1.0, 1.0, 80, 8, 640), (0.33, 0.5, 6, 160, 640)]:
    net = Net(d, w, nc, "silu")
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=1))
    net = net.to("cuda").eval().set_precision("bf16")
    x = synth.synth_images(8, S, S, seed=3).repeat(B // 8, 1, 1, 1).contiguous().cuda()
    for _ in range(2):
        outs = net(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        outs = net(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 3 * 1e3
    ok = all(bool(torch.isfinite(o).all()) for o in outs)
    dec = bbox.decode_outputs(outs, [S, S])
    dets = bbox.post_process(dec, nc, [S, S], [[i, S, S] for i in range(B)], 0.5, 0.6)
    print("depth %.2f width %.2f nc %d B %d: %.2f ms/forward (%.0f img/s) finite %s dets %d mem %.1f GB" % (
        d, w, nc, B, ms, B / ms * 1e3, ok, sum(x_.shape[0] for x_ in dets), torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del net, x, outs, dec
    torch.cuda.empty_cache()
