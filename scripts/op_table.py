"""Per-op time table of one forward (GPU box):  python scripts/op_table.py [kind] [batch] [size] [precision]
Wraps the op emitters of ccnet_amd.ops while the plan is compiled to label every recorded op, then times the plan with an event
pair around each op on one stream (ccnet_plan_run_timed, min of 5)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops, synth, _lib

kind = sys.argv[1] if len(sys.argv) > 1 else "net"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = int(sys.argv[3]) if len(sys.argv) > 3 else 640
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
os.environ.setdefault("CCNET_LANES", "0")

log, state = [], {}
_enter = ops.Recorder.__enter__
def enter(self):
    r = _enter(self)
    state["h"] = self.handle
    return r
ops.Recorder.__enter__ = enter
cur = lambda: _lib.lib().ccnet_plan_num_ops(state["h"])

def desc(name, a):
    if name == "conv2d":
        ins, pc, b, h, w, stride, pad, out, act = a[:9]
        return "%dx%d s%d  %s -> %d @%dx%d %s%s" % (pc.kh, pc.kw, stride, "+".join(str(s.C) for s in ins), pc.cout, h, w, act,
                                                      " +res" if len(a) > 9 and a[9] is not None else "")
    return ""

for name in ("conv2d", "csp_fused", "stem_conv", "sppf_pool", "cbam", "eca", "dwconv3x3", "transformer_fused", "layernorm", "attention", "asff_fuse", "maxpool3x3s2"):
    f = getattr(ops, name)
    def g(*a, _f=f, _n=name, **k):
        n0 = cur(); r = _f(*a, **k); log.append((_n, n0, cur(), desc(_n, a))); return r
    setattr(ops, name, g)

net = importlib.import_module("ccnet_amd.network." + kind).Net(0.33, 0.5, 6, "silu")
net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0, bn_calib=kind))
net = net.to("cuda").eval().set_precision(prec)
x = synth.synth_images(B, S, S, seed=1).cuda()
net(x)
plan = net.plan_for(B, S, S, x.device)[0]
ms = None
for _ in range(5):
    c = plan.run_timed(x)
    ms = c if ms is None else [min(p, q) for p, q in zip(ms, c)]
info = plan.op_info()
tot = 0.0
for name, n0, n1, d in log:
    t = sum(ms[n0:n1]); fl = sum(i[1] for i in info[n0:n1]); by = sum(i[2] for i in info[n0:n1])
    tot += t
    print("%3d-%3d %-18s %-42s %7.1f us  %7.2f GF %7.1f MB  %7.1f TF/s %7.1f GB/s" % (n0, n1 - 1, name, d, t * 1e3, fl / 1e9, by / 1e6,
          fl / t / 1e9 if t > 0 else 0, by / t / 1e6 if t > 0 else 0))
print("sum of ops %.3f ms over %d ops" % (tot, len(ms)))
