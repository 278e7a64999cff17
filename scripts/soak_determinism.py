import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import synth
from ccnet_amd.network.net import Net
from ccnet_amd.network.net_deploy import Net as NetD
from ccnet_amd.utils import bbox
for cls, kind in ((Net, "net"), (NetD, "net_deploy")):
    net = cls(0.33, 0.5, 6, "silu")
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0, bn_calib=kind))
    net = net.to("cuda").eval().set_precision("bf16")
    xs = [torch.cat([synth.synth_images(8, 640, 640, seed=100 * k + i) for i in range(8)], 0).cuda() for k in range(3)]
    info = [[i, 640, 640] for i in range(64)]
    ref = []
    for x in xs:
        o = [t.clone() for t in net(x)]
        d = bbox.post_process(bbox.decode_outputs(o, [640, 640]), 6, [640, 640], info, 0.5, 0.6)
        ref.append((o, d))
    bad = 0
    for it in range(60):
        k = it % 3
        o = net(xs[k])
        d = bbox.post_process(bbox.decode_outputs(o, [640, 640]), 6, [640, 640], info, 0.5, 0.6)
        same = all(torch.equal(a, b) for a, b in zip(o, ref[k][0])) and all(np.array_equal(a, b) for a, b in zip(d, ref[k][1]))
        bad += 0 if same else 1
    print(kind, "60 repeated steps over 3 alternating batches: mismatching steps =", bad, flush=True)
    assert bad == 0
print("soak ok")
