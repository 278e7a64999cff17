"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the image preparation that precedes the hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path
(ccnet_amd/utils/preprocess.py -> libccnet_hip) never does.

What is restated
  * tools/eval.py:134-140 -- `image.convert("RGB")`, `image.resize((W, H), Image.BICUBIC)`, float32, `/= 255.0`,
    `-= [0.485, 0.456, 0.406]`, `/= [0.229, 0.224, 0.225]`, HWC -> CHW.
  * utils/datasets.py:123-134 (+ :73-77) -- the `random=False` letterbox: scale = min(w/iw, h/ih), nw = int(iw*scale),
    nh = int(ih*scale), bicubic resize to (nw, nh), paste at ((w-nw)//2, (h-nh)//2) on a (128,128,128) canvas, then the same
    normalisation.
  * The resize itself is third-party: Pillow's ImagingResample (src/libImaging/Resample.c; the reference pins no Pillow
    version, 12.2.0 is installed in this image).  Its published algorithm for 8-bit images: per axis a table of
    (first tap, tap count) and double-precision bicubic (a = -0.5) weights over a support of 2*max(scale,1), normalised to sum
    1, converted to 22-bit fixed point with round-half-away; a horizontal pass then a vertical pass, each
    `clip8((2^21 + sum(pixel*k)) >> 22)` with an 8-bit intermediate image.

Pinned by: tests/golden/preprocess_*.npz, produced by oracle/make_golden_preprocess.py from the reference's own
`evaluate_one_img` and `Datasets.__getitem__` running in this container (with Pillow doing the resize), and live against
Pillow for random sizes in tests/test_cpu.py.  The cv2.INTER_CUBIC flavour of tools/demo.py:53 is NOT restated (OpenCV is
absent: parity unpinned, left with the caller's cv2).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = np.array([0.485, 0.456, 0.406])
STD = np.array([0.229, 0.224, 0.225])


def bicubic_filter(x):
    """Keys cubic with a = -0.5 (Resample.c: bicubic_filter)."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size, support=2.0, filt=bicubic_filter):
    """Resample.c: precompute_coeffs + normalize_coeffs_8bpc for the full box (0, in_size).
    Returns bounds int32 (out_size, 2) = (first tap, tap count), kk int32 (out_size, ksize), ksize."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            v = w * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(img, bounds, kk, axis):
    """One resampling pass over `axis` (0 = vertical, 1 = horizontal) of an (H, W, C) uint8 image."""
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for xx in range(bounds.shape[0]):
        xmin, cnt = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for j in range(cnt):
            acc += src[xmin + j] * int(kk[xx, j])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img, out_w, out_h):
    """PIL `Image.resize((out_w, out_h), Image.BICUBIC)` on an (H, W, C) uint8 array: horizontal pass first (only when the
    width changes), then vertical (only when the height changes) -- Resample.c: ImagingResample."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    ih, iw = img.shape[:2]
    if iw != out_w:
        b, k, _ = precompute_coeffs(iw, out_w)
        img = _pass(img, b, k, 1)
    if ih != out_h:
        b, k, _ = precompute_coeffs(ih, out_h)
        img = _pass(img, b, k, 0)
    return img


def normalise(img_u8):
    """float32, /= 255.0, -= mean (float64 operand, result cast back), /= std, HWC -> CHW (tools/eval.py:136-140)."""
    data = np.array(img_u8, dtype="float32")
    data /= 255.0
    data -= MEAN
    data /= STD
    return np.transpose(data, (2, 0, 1))


def preprocess_eval(img_u8, input_shape):
    """tools/eval.py:134-140 for an RGB (ih, iw, 3) uint8 image -> (3, H, W) float32."""
    h, w = input_shape
    return normalise(resize_bicubic_u8(img_u8, w, h))


def letterbox_geometry(iw, ih, input_shape):
    """utils/datasets.py:124-128 -> (nw, nh, dx, dy)."""
    h, w = input_shape
    scale = min(w / iw, h / ih)
    nw = int(iw * scale)
    nh = int(ih * scale)
    return nw, nh, (w - nw) // 2, (h - nh) // 2


def preprocess_letterbox(img_u8, input_shape):
    """utils/datasets.py:123-134 + :73-77 -> (3, H, W) float32."""
    h, w = input_shape
    ih, iw = img_u8.shape[:2]
    nw, nh, dx, dy = letterbox_geometry(iw, ih, input_shape)
    canvas = np.full((h, w, 3), 128, np.uint8)
    canvas[dy:dy + nh, dx:dx + nw] = resize_bicubic_u8(img_u8, nw, nh)
    return normalise(canvas)
