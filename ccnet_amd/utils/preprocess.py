"""Image preparation of the reference's callers on device, host side.

`preprocess(images, input_shape)` is tools/eval.py:134-140 for a batch of decoded 8-bit images: Pillow-exact bicubic resize to
(W, H), float32, `/= 255.0`, `-= mean`, `/= std`, HWC -> CHW -- resize, normalisation and transpose in two HIP kernels of
libccnet_hip (csrc/preprocess.hip); `letterbox=True` is the `random=False` branch of utils/datasets.py:123-134 (+ :73-77).
Only the file decode (PIL / cv2 `imread`) stays with the caller.  The 8-bit results of the resize pass through a 3x256 table
that is filled here with the very numpy statements of the reference, so the float32 output is bit-identical to the
reference's, not merely close.  The cv2.INTER_CUBIC resize of tools/demo.py:53 is a different filter and stays on the
caller's OpenCV (ccnet_amd/tools/demo.py:preprocess).
"""
import ctypes as C
import threading

import numpy as np
import torch

from .. import _lib

_MEAN = np.array([0.485, 0.456, 0.406])
_STD = np.array([0.229, 0.224, 0.225])
_cache = {}
_lock = threading.Lock()


def resample_tables(in_size, out_size):
    """(bounds int32 (out, 2), kk int32 (out, ksize)) of Pillow's bicubic resampling for one axis; host only."""
    L = _lib.lib()
    ks = L.ccnet_resample_ksize(int(in_size), int(out_size))
    if ks < 0:
        _lib.check(ks)
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ks), np.int32)
    _lib.check(L.ccnet_resample_coeffs(int(in_size), int(out_size), bounds.ctypes.data_as(C.c_void_p), kk.ctypes.data_as(C.c_void_p), ks))
    return bounds, kk


def normalisation_lut():
    """lut[c, v] = what tools/eval.py:136-139 makes of the 8-bit value v in channel c (same statements, same rounding)."""
    data = np.array(np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, 2), dtype="float32")
    data /= 255.0
    data -= _MEAN
    data /= _STD
    return np.ascontiguousarray(data[0].T)


def letterbox_geometry(iw, ih, input_shape):
    """utils/datasets.py:124-128 -> (nw, nh, dx, dy)."""
    h, w = input_shape
    scale = min(w / iw, h / ih)
    nw = int(iw * scale)
    nh = int(ih * scale)
    return nw, nh, (w - nw) // 2, (h - nh) // 2


def _cached(key, make):
    with _lock:
        v = _cache.get(key)
        if v is None:
            if len(_cache) > 256:
                _cache.clear()
            v = _cache[key] = make()
        return v


def _tables(in_size, out_size, device):
    def make():
        b, k = resample_tables(in_size, out_size)
        return torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), k.shape[1]
    return _cached(("t", in_size, out_size, str(device)), make)


def _lut(device):
    return _cached(("lut", str(device)), lambda: torch.from_numpy(normalisation_lut()).to(device))


def preprocess_group(src, out, input_shape, letterbox=False):
    """src: uint8 CUDA tensor (n, ih, iw, 3) or (n, ih, iw) (grey), contiguous; out: float32 CUDA tensor (n, 3, H, W) (a
    contiguous slice of the batch).  Runs on torch's current stream."""
    L = _lib.lib()
    if not (src.is_cuda and out.is_cuda and src.dtype == torch.uint8 and out.dtype == torch.float32):
        raise TypeError("preprocess_group: uint8 CUDA source and float32 CUDA destination expected")
    if src.dim() not in (3, 4) or (src.dim() == 4 and src.shape[3] != 3):
        raise ValueError("preprocess_group: source must be (n, ih, iw, 3) or (n, ih, iw), got %s" % (tuple(src.shape),))
    if src.device != out.device:
        raise ValueError("preprocess_group: source on %s, destination on %s" % (src.device, out.device))
    src = src.contiguous()
    n, ih, iw = src.shape[:3]
    ch = 3 if src.dim() == 4 else 1
    h, w = int(input_shape[0]), int(input_shape[1])
    if tuple(out.shape) != (n, 3, h, w) or not out.is_contiguous():
        raise ValueError("preprocess_group: destination must be a contiguous (%d, 3, %d, %d) tensor" % (n, h, w))
    nw, nh, dx, dy = letterbox_geometry(iw, ih, (h, w)) if letterbox else (w, h, 0, 0)
    hb, hk, hks = _tables(iw, nw, src.device)
    vb, vk, vks = _tables(ih, nh, src.device)
    lut = _lut(src.device)
    ws_bytes = L.ccnet_preprocess_workspace_bytes(n, ih, nw)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(L.ccnet_preprocess(src.data_ptr(), n, ih, iw, ch, ih * iw * ch, iw * ch, ch, out.data_ptr(), h, w, 3 * h * w, h * w, dx, dy, nw, nh,
                                      hb.data_ptr(), hk.data_ptr(), hks, vb.data_ptr(), vk.data_ptr(), vks, lut.data_ptr(), 128, ws.data_ptr(),
                                      ws_bytes, _lib.stream_ptr()))
    return out


def preprocess(images, input_shape, device="cuda", letterbox=False, names=None):
    """images: sequence of decoded RGB uint8 arrays (ih, iw, 3) or grey (ih, iw) of any sizes (or one stacked array / uint8
    tensor (n, ih, iw[, 3])).  Returns (float32 CUDA tensor (B, 3, H, W), image_info) with image_info[i] = [name or 0, iw, ih]
    as tools/eval.py:133 / tools/demo.py:52 build it.  Images of equal size travel and run together."""
    h, w = int(input_shape[0]), int(input_shape[1])
    if isinstance(images, torch.Tensor):
        groups = {tuple(images.shape[1:]): (list(range(images.shape[0])), images)}
        shapes = [tuple(images.shape[1:])] * images.shape[0]
    else:
        arrs = [np.asarray(im) for im in images]
        for a in arrs:
            if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
                raise ValueError("preprocess: decoded 8-bit RGB (ih, iw, 3) or grey (ih, iw) images expected, got %s %s" % (a.dtype, a.shape))
        shapes = [a.shape for a in arrs]
        groups = {}
        for i, a in enumerate(arrs):
            groups.setdefault(a.shape, ([], []))
            groups[a.shape][0].append(i)
            groups[a.shape][1].append(a)
        groups = {s: (idx, torch.from_numpy(np.stack(lst))) for s, (idx, lst) in groups.items()}
    B = len(shapes)
    out = torch.empty((B, 3, h, w), dtype=torch.float32, device=device)
    for shape, (idx, stacked) in groups.items():
        src = stacked.to(out.device, non_blocking=True)
        if idx == list(range(idx[0], idx[0] + len(idx))):
            preprocess_group(src, out[idx[0]:idx[0] + len(idx)], (h, w), letterbox)
        else:
            tmp = torch.empty((len(idx), 3, h, w), dtype=torch.float32, device=out.device)
            preprocess_group(src, tmp, (h, w), letterbox)
            out[torch.as_tensor(idx, device=out.device)] = tmp
    info = [[names[i] if names is not None else 0, int(shapes[i][1]), int(shapes[i][0])] for i in range(B)]
    return out, info
