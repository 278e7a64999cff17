"""Deterministic synthetic weights and images for CCNet parity tests and benchmarks.

The reference ships no checkpoints (weights/README.md is a placeholder) and PyTorch's default init
collapses activations to ~1e-7 by dark5, so every parity run uses weights produced
here.  Values depend only on (seed, state_dict key, element index) through an integer hash, and are
built from integer->float conversions, adds and multiplies alone (no log/cos), so they are bit-identical
on every machine and numpy build.  Nothing here touches the GPU.
"""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(text: str) -> int:
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _bits(key: str, n: int, seed: int, stream: int = 0) -> np.ndarray:
    base = np.uint64((_fnv1a(key) ^ (seed * 0x2545F4914F6CDD1D) ^ (stream * 0xD6E8FEB86659FD93)) & 0xFFFFFFFFFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _splitmix(_splitmix(idx * np.uint64(0xA0761D6478BD642F) + base))


def uniform(key: str, n: int, seed: int = 0, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    """n float64 values in [lo, hi) with 32 bits of resolution."""
    u = (_bits(key, n, seed) >> np.uint64(32)).astype(np.float64) * (1.0 / 4294967296.0)
    return lo + (hi - lo) * u


def normal(key: str, n: int, seed: int = 0, std: float = 1.0) -> np.ndarray:
    """Approximately N(0, std^2): centred sum of four 16-bit uniforms (Irwin-Hall), exact arithmetic."""
    b = _bits(key, n, seed, stream=1)
    m = np.uint64(0xFFFF)
    s = ((b & m) + ((b >> np.uint64(16)) & m) + ((b >> np.uint64(32)) & m) + ((b >> np.uint64(48)) & m)).astype(np.float64)
    z = ((s + 2.0) * (1.0 / 65536.0) - 2.0) * 1.7320508075688772  # var(sum of 4 U) = 1/3
    return z * std


def _fill(key, shape, seed, kind, a=0.0, b=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    if kind == "normal":
        v = normal(key, n, seed, std=b)
    else:
        v = uniform(key, n, seed, lo=a, hi=b)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


# gain that keeps activation std ~O(1) through conv -> BN(~identity) -> SiLU chains
_CONV_GAIN = 1.7
# Mean of the BN biases.  With zero-mean pre-activations a random SiLU network sits past the edge of chaos
# (a perturbation grows ~1.07x per layer, ~50x over the net), which would turn bf16 rounding noise into 20 %
# logit error -- a property of random weights, not of any implementation.  A positive mean keeps more units in
# SiLU's linear region and brings the per-layer growth to ~1, like a trained detector.
BN_BIAS_MEAN = 0.8


def synth_tensor(key: str, shape, seed: int = 0) -> torch.Tensor:
    """Variance-preserving value for one state_dict entry, chosen from its key name and shape."""
    shape = tuple(int(s) for s in shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_mean":
        return _fill(key, shape, seed, "normal", b=0.1)
    if leaf == "running_var":
        return _fill(key, shape, seed, "uniform", 0.5, 1.5)
    is_norm = ".bn." in key or key.endswith("bn.weight") or key.endswith("bn.bias") or ".norm" in key
    if is_norm and leaf == "weight":
        return _fill(key, shape, seed, "uniform", 0.8, 1.2)
    if is_norm and leaf == "bias":
        t = _fill(key, shape, seed, "normal", b=0.1)
        return t + BN_BIAS_MEAN if (".bn." in key or key.endswith("bn.bias")) else t
    head_pred = any(t in key for t in ("cls_preds", "reg_preds", "obj_preds"))
    if leaf in ("bias", "in_proj_bias"):
        if head_pred:
            # wide biases so obj*cls straddles the thresholds; objectness centred low so that, as in a
            # trained detector, only a few percent of the anchors survive conf_thres=0.5
            t = _fill(key, shape, seed, "normal", b=1.0)
            return t - 1.6 if "obj_preds" in key else t
        return _fill(key, shape, seed, "normal", b=0.05)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        gain = _CONV_GAIN
        if head_pred:
            gain = 0.6
        elif "transformer" in key:
            gain = 1.0 if "attn" in key else 1.4
        elif "channel_attention" in key:
            gain = 1.5
        elif "spatial_attention" in key:
            gain = 0.5
        elif len(shape) == 3:  # ECA Conv1d (1,1,3)
            gain = 1.5
        elif "dark5.2.conv" in key:  # MobileViT plain convs: no BN, no activation
            gain = 1.0
        return _fill(key, shape, seed, "normal", b=gain / np.sqrt(fan_in))
    return _fill(key, shape, seed, "normal", b=0.05)


def bn_calib_path(kind: str) -> str:
    """BN running statistics calibrated once with the reference model (oracle/make_golden.py:calibrate_bn)
    for the s-size / 6-class nets; committed fixtures, because a pure hash init either collapses or blows up
    through ~60 unnormalised layers."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.path.join(root, "tests", "golden", "bn_calib_%s.npz" % kind)


def synth_state_dict(template, seed: int = 0, bn_calib=None):
    """template: mapping key -> tensor (only shapes are read). Returns an OrderedDict of new tensors.
    bn_calib: net kind whose committed BN running stats replace the hashed ones where shapes match."""
    from collections import OrderedDict
    import os
    out = OrderedDict()
    for k, v in template.items():
        out[k] = synth_tensor(k, v.shape, seed)
    if bn_calib is not None and os.path.exists(bn_calib_path(bn_calib)):
        with np.load(bn_calib_path(bn_calib)) as z:
            for k in z.files:
                if k in out and tuple(out[k].shape) == tuple(z[k].shape):
                    out[k] = torch.from_numpy(z[k].astype(np.float32))
    return out


def synth_images(batch: int, height: int, width: int, seed: int = 0) -> torch.Tensor:
    """fp32 (B,3,H,W), roughly N(0,1) like mean/std-normalised images (tools/demo.py:56-59).
    Image i depends only on (seed, i, H, W), so any batch sharding sees the same pictures."""
    imgs = []
    for i in range(batch):
        v = normal("image/%d/%dx%d" % (i, height, width), 3 * height * width, seed)
        imgs.append(torch.from_numpy(v.astype(np.float32).reshape(3, height, width)))
    return torch.stack(imgs, 0)


def tensor_checksum(t: torch.Tensor) -> float:
    """Order-independent-ish fp64 checksum used to pin generated weights in fixtures."""
    a = t.detach().to("cpu", torch.float64).reshape(-1)
    if a.numel() == 0:
        return 0.0
    w = torch.arange(1, a.numel() + 1, dtype=torch.float64) % 251 + 1
    return float((a * w).sum())
