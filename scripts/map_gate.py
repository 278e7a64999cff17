"""bf16 vs fp32 at detection level (run on the GPU box): mAP@[.5:.95] of each precision's detections (conf 0.001,
tools/eval.py:49) against the fp32 detections at conf 0.5 (demo.py:89) taken as ground truth."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import synth
from ccnet_amd.utils import bbox, map as cmap

def dets(kind, prec, x, conf):
    net = importlib.import_module("ccnet_amd.network." + kind).Net(0.33, 0.5, 6, "silu")
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0, bn_calib=kind))
    net = net.to("cuda").eval().set_precision(prec)
    B, _, H, W = x.shape
    dec = bbox.decode_outputs(net(x), [H, W])
    info = [[i, W, H] for i in range(B)]
    return [bbox.post_process(dec.clone(), 6, [H, W], info, c, 0.6) for c in conf]

for kind, B, S in (("net_deploy", 8, 224), ("net_deploy", 4, 640), ("net", 4, 224), ("net_amff", 4, 224), ("net_small_target", 2, 224)):
    x = synth.synth_images(B, S, S, seed=3).to("cuda")
    f_hi, f_lo = dets(kind, "fp32", x, (0.5, 0.001))
    b_hi, b_lo = dets(kind, "bf16", x, (0.5, 0.001))
    gt = [[(r[0],) + tuple(r[2:]) for r in img] for img in cmap.detections_to_records(f_hi)]
    m_f, v_f = cmap.map_50_95(gt, cmap.detections_to_records(f_lo), 6)
    m_b, v_b = cmap.map_50_95(gt, cmap.detections_to_records(b_lo), 6)
    print("%-17s B=%d %d^2: gt %d boxes; dets fp32 %d bf16 %d (conf>=.5: %d vs %d); mAP50-95 fp32 %.4f bf16 %.4f; mAP50 %.4f / %.4f; mAP75 %.4f / %.4f; mAP95 %.4f / %.4f" % (
        kind, B, S, sum(len(g) for g in gt), sum(len(d) for d in f_lo), sum(len(d) for d in b_lo), sum(len(d) for d in f_hi), sum(len(d) for d in b_hi),
        m_f, m_b, v_f[0], v_b[0], v_f[5], v_b[5], v_f[9], v_b[9]), flush=True)
