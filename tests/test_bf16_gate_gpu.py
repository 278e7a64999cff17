"""Detection-level gate of the bf16 path: box matching against the fp32 path at the demo thresholds (conf 0.5 / nms 0.6,
tools/demo.py:89-91) on 8 distinct 640x640 images.  north_star asks for box-level mAP within 0.1 pt; with neither a trained
checkpoint nor a dataset offline that gate itself is UNPINNED, and this test pins what can be measured: every fp32 detection
is matched to a same-image, same-class bf16 detection (ccnet_amd/utils/map.py:match_boxes) and the matched fractions, the
count delta and the score differences are bounded with about one third of margin to the values measured on MI355X
(scripts/bf16_gate.py):

    net_deploy: 2441 fp32 / 2390 bf16 boxes; IoU >= 0.5: 92.1 %, >= 0.75: 79.4 %, >= 0.9: 28.1 %; mean IoU 0.845; |score diff| <= 0.028
    net       :  542 fp32 /  546 bf16 boxes; IoU >= 0.5: 93.9 %, >= 0.75: 84.0 %, >= 0.9: 31.2 %; mean IoU 0.853; |score diff| <= 0.030

A doubling of the bf16 logit error (a dropped fp32 accumulation, a second rounding per layer) moves the IoU >= 0.75 fraction
below 60 % and fails here.  The numbers hold for the committed synthetic-weight recipe (BN biases of mean 0.8); with zero-mean
BN biases the random net is chaotic and NO 8-bit-mantissa implementation agrees with fp32 (script output in DESIGN.md).

A second, independent recipe (weight seed 1, BN-bias mean 0.4, BN statistics recalibrated by oracle.calibrate_bn) is gated
too, with the bounds ITS measurement allows (profiles/r03_bf16_gate.txt): far fewer boxes clear conf 0.5 there (35 / 124), the
bf16 logits sit at 4.8-5.9 % rms instead of 2.6-3.7 %, and 86 % of the boxes match at IoU 0.5:

    net_deploy:   35 fp32 /  37 bf16 boxes; IoU >= 0.5: 85.7 %, >= 0.75: 68.6 %; mean IoU 0.821; |score diff| <= 0.045
    net       :  124 fp32 / 124 bf16 boxes; IoU >= 0.5: 86.3 %, >= 0.75: 79.8 %; mean IoU 0.846; |score diff| <= 0.040"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _gate():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "bf16_gate.py")
    spec = importlib.util.spec_from_file_location("bf16_gate", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("kind", ["net_deploy", "net"])
def test_bf16_boxes_match_fp32_boxes(kind):
    r = _gate().run(kind, 8, 640, 0.8, verbose=False)
    assert r["n_fp32"] > 300
    assert abs(r["n_bf16"] - r["n_fp32"]) <= 0.03 * r["n_fp32"], r
    assert r["frac_iou50"] >= 0.88 and r["frac_iou75"] >= 0.70 and r["frac_iou90"] >= 0.18, r
    assert r["mean_iou"] >= 0.80 and r["max_score_diff"] <= 0.05, r
    # rms error of the bf16 logits against the fp32 CPU oracle, per level
    assert all(v < 0.06 for k, v in r["rms_bf16"].items() if k.startswith("head")), r["rms_bf16"]
    # and the fp32 path sits at fp32 rounding level on the same inputs
    assert all(v < 1e-4 for v in r["rms_fp32"].values()), r["rms_fp32"]


@pytest.mark.parametrize("kind", ["net_deploy", "net"])
def test_bf16_boxes_match_fp32_boxes_second_recipe(kind):
    r = _gate().run(kind, 8, 640, 0.4, verbose=False, wseed=1)
    assert r["n_fp32"] >= 25
    assert abs(r["n_bf16"] - r["n_fp32"]) <= max(3, 0.08 * r["n_fp32"]), r
    assert r["frac_iou50"] >= 0.78 and r["frac_iou75"] >= 0.55, r
    assert r["mean_iou"] >= 0.78 and r["max_score_diff"] <= 0.08, r
    assert all(v < 0.09 for k, v in r["rms_bf16"].items() if k.startswith("head")), r["rms_bf16"]
    assert all(v < 1e-4 for v in r["rms_fp32"].values()), r["rms_fp32"]
