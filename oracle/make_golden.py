"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference, which only
exists in the build container) on synthetic weights/images from ccnet_amd.synth, and check the oracle
restatement against it on the way.  TEST INFRASTRUCTURE ONLY; never imported by the product.

    python oracle/make_golden.py            # regenerate all fixtures
    python oracle/make_golden.py --check    # only compare oracle vs reference, write nothing

torchvision is not installed here: `utils/bbox.py:10` is satisfied with an injected stub module whose
`boxes.batched_nms` is the oracle's restated NMS (the NMS arithmetic is third-party and parity-unpinned;
everything around it in post_process -- xyxy, mask, gather, numpy rescale -- is the reference's own code).
"""
import argparse
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, ROOT)

from ccnet_amd import synth  # noqa: E402
from oracle import ccnet_oracle as orc  # noqa: E402

KINDS = ("net", "net_deploy", "net_amff", "net_small_target", "net_old")
DEPTH, WIDTH, NC = 0.33, 0.5, 6


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be regenerated in the build container")
    sys.path.insert(0, REF)
    tv = types.ModuleType("torchvision")
    ops = types.ModuleType("torchvision.ops")
    boxes = types.ModuleType("torchvision.ops.boxes")

    def batched_nms(b, s, idxs, thr):
        keep = orc.batched_nms_reference(b.cpu().numpy(), s.cpu().numpy(), idxs.cpu().numpy(), thr)
        return torch.as_tensor(keep, dtype=torch.long)

    boxes.batched_nms = batched_nms
    ops.boxes = boxes
    ops.nms = None
    tv.ops = ops
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.ops", ops)
    sys.modules.setdefault("torchvision.ops.boxes", boxes)
    mods = {k: importlib.import_module("network." + k) for k in KINDS}
    bbox = importlib.import_module("utils.bbox")
    return mods, bbox


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def calibrate_bn(mods, kind, write):
    """BN running stats that keep activations O(1): every BatchNorm2d in train mode with
    momentum 1.0 (everything else eval, so Transformer dropout stays off), one forward of the REFERENCE model on
    4 synthetic 224x224 images.  The stats are committed as tests/golden/bn_calib_<kind>.npz so that both the
    build container and the GPU box regenerate bit-identical weights (hash values + these stats)."""
    model = mods[kind].Net(DEPTH, WIDTH, NC, "silu")
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed=0))
    model.eval()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.train()
            m.momentum = 1.0
    with torch.no_grad():
        model(synth.synth_images(4, 224, 224, seed=1))
    stats = {k: v.numpy().copy() for k, v in model.state_dict().items()
             if k.endswith("running_mean") or k.endswith("running_var")}
    if write:
        np.savez_compressed(synth.bn_calib_path(kind), **stats)
    return stats


def run_case(mods, bbox, kind, B, H, W, conf, nms, check_only, taps_wanted=()):
    torch.manual_seed(0)
    model = mods[kind].Net(DEPTH, WIDTH, NC, "silu")
    sd = synth.synth_state_dict(model.state_dict(), seed=0, bn_calib=kind)
    model.load_state_dict(sd)
    model.eval()
    x = synth.synth_images(B, H, W, seed=0)
    with torch.no_grad():
        ref_heads = model(x)
    taps = {}
    my_heads = orc.forward(kind, sd, x, DEPTH, WIDTH, "silu", taps)
    errs = [rel_err(a, b) for a, b in zip(my_heads, ref_heads)]
    stds = [float(h.std()) for h in ref_heads]
    tap_std = {k: round(float(v.std()), 3) for k, v in taps.items()}
    print("%-16s B=%d %dx%d  oracle-vs-reference head rel err %s  head std %s" % (
        kind, B, H, W, ["%.1e" % e for e in errs], ["%.2f" % s for s in stds]))
    print("    tap std:", tap_std)
    assert max(errs) < 2e-5, "oracle forward diverges from the reference"

    ref_dec = bbox.decode_outputs([h.clone() for h in ref_heads], [H, W])
    my_dec = orc.decode_outputs(ref_heads, [H, W])
    assert torch.equal(ref_dec.contiguous(), my_dec), "oracle decode is not bit-identical to the reference"
    info = [[i, W + 37 * i + 5, H + 11 * i + 3] for i in range(B)]
    ref_pp = bbox.post_process(ref_dec.clone(), NC, [H, W], info, conf_thres=conf, nms_thres=nms)
    my_pp = orc.post_process(my_dec, NC, [H, W], info, conf_thres=conf, nms_thres=nms)
    for a, b in zip(ref_pp, my_pp):
        assert a.dtype == np.float32 and a.shape == b.shape and np.array_equal(a, b), "oracle post_process mismatch"
    print("    decode bit-identical; post_process identical; detections/img:", [int(a.shape[0]) for a in ref_pp])
    if check_only:
        return
    out = {"kind": kind, "B": B, "H": H, "W": W, "conf": conf, "nms": nms, "depth": DEPTH, "width": WIDTH, "nc": NC,
           "image_info": np.asarray(info, dtype=np.int64),
           "decoded": ref_dec.contiguous().numpy(),
           "det_counts": np.asarray([a.shape[0] for a in ref_pp], dtype=np.int64),
           "dets": np.concatenate(ref_pp, 0) if ref_pp else np.zeros((0, 7), np.float32),
           "weight_checksum": np.float64(sum(synth.tensor_checksum(v) for v in sd.values())),
           "image_checksum": np.float64(synth.tensor_checksum(x))}
    for i, h in enumerate(ref_heads):
        out["head%d" % i] = h.numpy()
    for name in taps_wanted:
        t = taps[name]
        # strided subsample keeps fixtures small: every 3rd channel, every 2nd pixel
        out["tap_" + name] = t[:, ::3, ::2, ::2].contiguous().numpy()
    path = os.path.join(ROOT, "tests", "golden", "%s_b%d_%d.npz" % (kind, B, H))
    np.savez_compressed(path, **out)
    print("    wrote", os.path.relpath(path, ROOT), "%.0f KB" % (os.path.getsize(path) / 1024))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(8)
    mods, bbox = _import_reference()
    taps = ("stem", "dark3", "dark5", "gate3", "pan_out1")
    for kind in KINDS:
        if not args.check or not os.path.exists(synth.bn_calib_path(kind)):
            calibrate_bn(mods, kind, True)
        run_case(mods, bbox, kind, 1, 224, 224, 0.5, 0.6, args.check, taps)
        run_case(mods, bbox, kind, 2, 224, 224, 0.001, 0.6, args.check, taps)
    # rectangular input: pins the "stride = input_shape[0] / h for both axes" quirk (bbox.py:57)
    run_case(mods, bbox, "net_deploy", 1, 160, 256, 0.3, 0.4, args.check, ())


if __name__ == "__main__":
    main()
