"""nn.Module front of the MI355X path: same constructor, state_dict and forward contract as the reference's Net
classes (network/net.py:16-23,194-275), with forward executed as one recorded libccnet_hip plan."""
import collections
import os
import threading

import torch
import torch.nn as nn

from .. import _lib
from . import graph


class NetBase(nn.Module):
    KIND = None

    def __init__(self, depth=1.0, width=1.0, num_classes=80, act="silu"):
        super().__init__()
        if act not in ("silu", "relu", "relu6", "lrelu"):
            # same failure as get_activation_function (network/model_block.py:32-33)
            raise AttributeError("Unsupported activation function type: {}".format(act))
        self._cfg = graph.declare(self, self.KIND, depth, width, num_classes)
        self._act = act
        self._plans = collections.OrderedDict()  # LRU of compiled plans, newest last
        self._version = 0
        # fp32 (the reference's only defined semantics, gate 1e-3) unless the caller opts into bf16 storage: set_precision("bf16")
        # or CCNET_PRECISION=bf16.  bench.py and the tests always pass the precision explicitly.
        self.precision = os.environ.get("CCNET_PRECISION", "fp32")
        # compiled plans kept per instance; each holds a full set of activation buffers and packed weights for one
        # (B, H, W, precision, device), so the cache is bounded and the least recently used plan is destroyed
        self.max_plans = max(int(os.environ.get("CCNET_MAX_PLANS", "4")), 1)
        self.alias_outputs = False  # True: return views of the plan's own head buffers (overwritten by the next call)
        # one forward at a time per instance: the plan's buffers are shared, and a second thread's launches between this
        # thread's launches and its copy of the outputs would hand it the other thread's results (the socket server keeps
        # one loaded model for all request threads)
        self._fwd_lock = threading.Lock()
        self.eval()

    # ---- weights changed -> recorded plans hold stale packed copies
    def _invalidate(self):
        self._plans = collections.OrderedDict()
        self._version += 1

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._invalidate()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._invalidate()
        return r

    def set_precision(self, precision):
        """"bf16": bf16 storage / fp32 accumulate (head logits and decode stay fp32); "fp32": exact-fp32 MFMA everywhere."""
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.precision = precision
        return self

    def train(self, mode=True):
        if mode:
            raise RuntimeError("ccnet_amd Net is inference-only (BN is folded with its running statistics); use the reference to train")
        return super().train(False)

    def plan_for(self, B, H, W, device):
        dtype = _lib.BF16 if self.precision == "bf16" else _lib.F32
        key = (B, H, W, dtype, str(device))
        ent = self._plans.get(key)
        if ent is None:
            while len(self._plans) >= self.max_plans:
                # evict the least recently used plan: its launches are drained before its buffers go back to the allocator
                _, old = self._plans.popitem(last=False)
                with torch.cuda.device(old[2].device):
                    torch.cuda.synchronize()
                del old
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            with torch.cuda.device(device):
                ent = graph.compile_plan(self.KIND, sd, self._cfg, B, H, W, dtype, device, self._act)
            self._plans[key] = ent
        else:
            self._plans.move_to_end(key)
        return ent

    def forward(self, x):
        """x: float32 CUDA (B,3,H,W), H and W multiples of 32.  Returns the list of raw head maps (B, 5+nc, H/s, W/s),
        fp32, channel order [reg4, obj1, cls nc] -- same values/shapes as the reference; memory is NHWC underneath."""
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("Net.forward expects a (B,3,H,W) tensor")
        if not x.is_cuda:
            raise RuntimeError("ccnet_amd Net runs on the GPU only (libccnet_hip has no CPU path); move the input to 'cuda'")
        x = x.float().contiguous()
        B, _, H, W = x.shape
        with self._fwd_lock:
            plan, heads, _ = self.plan_for(B, H, W, x.device)
            with torch.cuda.device(x.device):
                # the plan's buffers are shared by every caller: a forward launched from another stream (torch streams are
                # per thread) must not start before the previous caller's copy of the outputs has finished on ITS stream
                last = getattr(plan, "last_use", None)
                if last is not None:
                    torch.cuda.current_stream().wait_event(last)
                plan.run(x)
                outs = [h.permute(0, 3, 1, 2) for h in heads]
                if not self.alias_outputs:
                    outs = [o.clone(memory_format=torch.preserve_format) for o in outs]
                ev = torch.cuda.Event()
                ev.record()
                plan.last_use = ev
            return outs

    # copies / pickles of the module carry weights, not compiled plans or the lock
    def __getstate__(self):
        d = self.__dict__.copy()
        d["_plans"] = collections.OrderedDict()
        d.pop("_fwd_lock", None)
        return d

    def __setstate__(self, d):
        super().__setstate__(d)
        self._fwd_lock = threading.Lock()
