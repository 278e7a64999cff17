"""Which layers make the bf16 error?  CPU experiment on the oracle (diagnostic).

The HIP bf16 path rounds every GEMM operand (activations and folded weights) to bf16 and accumulates in fp32.  Storing a
layer's OUTPUT in fp32 would change nothing for a convolution that consumes it -- the MFMA operand is rounded at the load all
the same -- so the question "which outputs should stay fp32" is really "which layers would have to run on fp32 operands".
This script emulates the operand rounding inside the oracle (conv2d, linear and bmm inputs rounded to bf16, fp32 accumulate)
and reports the logit error against the plain fp32 oracle with (a) everything rounded, (b) everything rounded except one
stage, (c) only one stage rounded.
    python scripts/bf16_layer_ablation.py [kind] [size] [batch]"""
import importlib
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F

from ccnet_amd import synth
from oracle import ccnet_oracle as orc

STATE = {"stage": "?", "round": lambda stage: False}


def r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


class FProxy:
    """stands in for torch.nn.functional inside the oracle module"""

    def __getattr__(self, name):
        return getattr(F, name)

    @staticmethod
    def conv2d(x, w, b=None, *a, **k):
        if STATE["round"](STATE["stage"]):
            x, w = r16(x), r16(w)
        return F.conv2d(x, w, b, *a, **k)

    @staticmethod
    def linear(x, w, b=None):
        if STATE["round"](STATE["stage"]):
            x, w = r16(x), r16(w)
        return F.linear(x, w, b)


def stage_of(prefix):
    m = re.search(r"(stem|dark\d|mobilevit|sppf|cbam\d?|gate\d?|lateral\w*|reduce\w*|bu_conv\d|C3_\w\d|asff\w*|stems|cls_convs|reg_convs|cls_preds|reg_preds|obj_preds)", prefix)
    return m.group(1) if m else prefix.split(".")[0]


def wrap(name):
    orig = getattr(orc, name)

    def inner(sd, p, *a, **k):
        old = STATE["stage"]
        STATE["stage"] = stage_of(p) if isinstance(p, str) else old
        try:
            return orig(sd, p, *a, **k)
        finally:
            STATE["stage"] = old
    setattr(orc, name, inner)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "net"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    net = importlib.import_module("ccnet_amd.network." + kind).Net(0.33, 0.5, 6, "silu")
    sd = synth.synth_state_dict(net.state_dict(), seed=0, bn_calib=kind)
    x = torch.cat([synth.synth_images(1, S, S, seed=1000 + i) for i in range(B)], 0)
    orc.F = FProxy()
    for name in ("conv_bn_act", "s2d_conv", "mobilevit", "cbam", "sppf", "csp"):
        if hasattr(orc, name):
            wrap(name)
    seen = []
    real_bmm = torch.bmm

    def bmm(a, b):
        if STATE["round"](STATE["stage"]):
            a, b = r16(a), r16(b)
        return real_bmm(a, b)
    orc.torch = type("T", (), {"__getattr__": lambda self, n: bmm if n == "bmm" else getattr(torch, n)})()

    def logits(rule):
        STATE["round"] = rule
        with torch.no_grad():
            outs = orc.forward(kind, sd, x, 0.33, 0.5, "silu")
        return torch.cat([o.flatten(2) for o in outs], 2)

    def note(stage):
        if stage not in seen:
            seen.append(stage)
        return False
    ref = logits(note)

    def rms(y):
        return float(((y - ref).pow(2).mean() / ref.pow(2).mean()).sqrt()) * 100

    print("%s %dx%d batch %d; stages: %s" % (kind, S, S, B, " ".join(seen)))
    allr = rms(logits(lambda s: True))
    print("all operands rounded to bf16:            logit rms error %.2f %%" % allr)
    print("%-12s %28s %28s" % ("stage", "all rounded but this one (%)", "only this one rounded (%)"))
    for st in seen:
        but = rms(logits(lambda s, st=st: s != st))
        only = rms(logits(lambda s, st=st: s == st))
        print("%-12s %28.2f %28.2f" % (st, but, only), flush=True)


main()
