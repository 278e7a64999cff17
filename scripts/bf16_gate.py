"""bf16 at detection level and per tap (run on the GPU box):  python scripts/bf16_gate.py [quick]

For each net: detections of the bf16 path against the fp32 path at conf 0.5 / nms 0.6 by box matching
(ccnet_amd/utils/map.py:match_boxes: same image, same class, IoU >= 0.9 / 0.75 / 0.5), the count delta, and the rms error
of every tap and head level against the fp32 CPU oracle.  Three arms of the synthetic-weight recipe are printed: BN biases with
mean 0.8 and weight seed 0 (ccnet_amd/synth.py, the committed fixtures), mean 0.4 with weight seed 1 (a second, independent
recipe: other weights, other bias level, BN statistics recalibrated by oracle.calibrate_bn) and mean 0.0 (seed 0, recalibrated),
so that the dependence of the bf16 numbers on the recipe is visible."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import synth
from ccnet_amd.utils import bbox, map as cmap
from oracle import ccnet_oracle as orc


def state_dict_for(kind, bias_mean, wseed=0):
    net = importlib.import_module("ccnet_amd.network." + kind).Net(0.33, 0.5, 6, "silu")
    if bias_mean == synth.BN_BIAS_MEAN and wseed == 0:
        return synth.synth_state_dict(net.state_dict(), seed=0, bn_calib=kind)
    old = synth.BN_BIAS_MEAN
    synth.BN_BIAS_MEAN = bias_mean
    try:
        sd = synth.synth_state_dict(net.state_dict(), seed=wseed)
    finally:
        synth.BN_BIAS_MEAN = old
    return orc.calibrate_bn(kind, sd, synth.synth_images(4, 224, 224, seed=1), 0.33, 0.5, "silu")


def run(kind, B, S, bias_mean, seed=3, conf=0.5, nms=0.6, verbose=True, wseed=0):
    sd = state_dict_for(kind, bias_mean, wseed)
    x = synth.synth_images(B, S, S, seed=seed)
    rtaps = {}
    rheads = orc.forward(kind, sd, x, 0.33, 0.5, "silu", rtaps)
    info = [[i, S, S] for i in range(B)]
    res, rms = {}, {}
    for prec in ("fp32", "bf16"):
        net = importlib.import_module("ccnet_amd.network." + kind).Net(0.33, 0.5, 6, "silu")
        net.load_state_dict(sd)
        net = net.to("cuda").eval().set_precision(prec)
        outs = [o.clone() for o in net(x.to("cuda"))]
        torch.cuda.synchronize()
        plan = net.plan_for(B, S, S, torch.device("cuda", torch.cuda.current_device()))[0]
        r = {}
        for name, t in plan.taps.items():
            if name in rtaps:
                got = t.buf[..., t.coff:t.coff + t.C].float().cpu().permute(0, 3, 1, 2)
                r[name] = float((got - rtaps[name]).pow(2).mean().sqrt() / rtaps[name].pow(2).mean().sqrt())
        for k, (o, h) in enumerate(zip(outs, rheads)):
            r["head%d" % k] = float((o.float().cpu() - h).pow(2).mean().sqrt() / h.pow(2).mean().sqrt())
        rms[prec] = r
        res[prec] = bbox.post_process(bbox.decode_outputs(outs, [S, S]), 6, [S, S], info, conf, nms)
    m = {thr: cmap.match_boxes(res["fp32"], res["bf16"], thr) for thr in (0.9, 0.75, 0.5)}
    n_ref, n_got = m[0.9]["n_ref"], m[0.9]["n_got"]
    out = {"kind": kind, "B": B, "S": S, "bias_mean": bias_mean, "wseed": wseed, "n_fp32": n_ref, "n_bf16": n_got,
           "frac_iou90": m[0.9]["matched"] / max(n_ref, 1), "frac_iou75": m[0.75]["matched"] / max(n_ref, 1),
           "frac_iou50": m[0.5]["matched"] / max(n_ref, 1), "mean_iou": m[0.5]["mean_iou"],
           "max_score_diff": m[0.5]["max_score_diff"], "rms_bf16": rms["bf16"], "rms_fp32": rms["fp32"]}
    if verbose:
        print("%-17s B=%d %d^2 bn_bias_mean %.1f weight seed %d: fp32 %d / bf16 %d detections (delta %+.2f %%); matched IoU>=.9 %.2f %%  >=.75 %.2f %%  >=.5 %.2f %%; "
              "mean IoU %.4f; max |score diff| %.4f" % (kind, B, S, bias_mean, wseed, n_ref, n_got, 100.0 * (n_got - n_ref) / max(n_ref, 1),
                                                      100 * out["frac_iou90"], 100 * out["frac_iou75"], 100 * out["frac_iou50"], out["mean_iou"],
                                                      out["max_score_diff"]), flush=True)
        print("    bf16 rms vs oracle per tap: " + "  ".join("%s %.4f" % (k, v) for k, v in rms["bf16"].items()), flush=True)
    return out


if __name__ == "__main__":
    quick = len(sys.argv) > 1
    cases = [("net_deploy", 8, 640), ("net", 8, 640), ("net_amff", 8, 640), ("net_small_target", 4, 640)]
    if quick:
        cases = cases[:2]
    for kind, B, S in cases:
        for bm, ws in ((0.8, 0), (0.4, 1), (0.0, 0)):
            run(kind, B, S, bm, wseed=ws)
