"""Golden vectors for the image preparation in front of the hot path, produced by the REAL reference in this container:
`tools.eval.evaluate_one_img` (tools/eval.py:127-145; the tensor it hands to the model is captured by a stand-in model) and
`utils.datasets.Datasets.__getitem__` with train=False (utils/datasets.py:48-82, :123-134), reading synthetic PNG files from
a temporary JPEGImages/ directory.  Pillow (installed, third-party) does the bicubic resize inside the reference code.
Test infrastructure only; nothing here runs on the GPU box.

    python oracle/make_golden_preprocess.py      -> tests/golden/preprocess_cases.npz

cv2 / coloredlogs / tensorboardX / torchvision / matplotlib / pycocotools are not installed; the lines exercised here never
touch them, so empty stand-in modules satisfy the module-level imports (as in oracle/make_golden_map.py).
"""
import os
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "preprocess_cases.npz")

# (name, source (ih, iw), mode, input_shape (h, w))
CASES = [
    ("neudet_up", (200, 200), "RGB", (224, 224)),       # the NEU-DET case of BASELINE configs[0]: 200x200 -> 224x224
    ("down_mixed", (97, 131), "RGB", (64, 96)),         # both axes shrink, different ratios (support > 2)
    ("up_down", (50, 40), "RGB", (32, 64)),             # height shrinks, width grows
    ("same_width", (70, 96), "RGB", (32, 96)),          # horizontal pass skipped by Pillow
    ("gray", (33, 47), "L", (64, 64)),                  # convert("RGB") of a single-channel file
]
LETTERBOX = [
    ("lb_wide", (80, 120), "RGB", (96, 96)),            # bars above and below
    ("lb_tall", (150, 60), "RGB", (64, 96)),            # bars left and right, odd offsets
    ("lb_exact", (48, 48), "RGB", (96, 96)),            # no bars
]


def _stub():
    for name in ("cv2", "coloredlogs", "tensorboardX", "torchvision", "torchvision.ops", "matplotlib", "matplotlib.pyplot", "pycocotools",
                 "pycocotools.coco", "pycocotools.cocoeval"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["coloredlogs"].install = lambda *a, **k: None
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.modules["torchvision"].ops = sys.modules["torchvision.ops"]
    sys.modules["torchvision.ops"].nms = None
    sys.modules["torchvision.ops"].boxes = None
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    sys.modules["matplotlib"].use = lambda *a, **k: None
    sys.modules["pycocotools.coco"].COCO = object
    sys.modules["pycocotools.cocoeval"].COCOeval = object


def synth(seed, shape, mode):
    """Smooth structure plus noise plus saturated patches, so that the negative lobes clip at both ends."""
    rng = np.random.default_rng(seed)
    ih, iw = shape
    c = 1 if mode == "L" else 3
    y, x = np.mgrid[0:ih, 0:iw]
    img = np.stack([127 + 100 * np.sin(x / (3.0 + k) + k) * np.cos(y / (5.0 - k)) for k in range(c)], -1)
    img += rng.normal(0, 30, img.shape)
    img[: ih // 4, : iw // 3] = 255
    img[ih // 2: ih // 2 + 3, :] = 0
    img = np.clip(img, 0, 255).astype(np.uint8)
    return img[..., 0] if mode == "L" else img


class _Captured(Exception):
    pass


class _Capture:
    """Stands in for the model of evaluate_one_img: keeps the tensor it is called with and leaves."""

    def __init__(self):
        self.x = None

    def eval(self):
        return self

    def __call__(self, images):
        self.x = images.detach().cpu().numpy().copy()
        raise _Captured()


def main():
    from PIL import Image
    _stub()
    sys.path.insert(0, REF)
    import importlib
    ev = importlib.import_module("tools.eval")
    ds = importlib.import_module("utils.datasets")
    out = {}
    with tempfile.TemporaryDirectory() as root:
        os.makedirs(os.path.join(root, "JPEGImages"))
        os.makedirs(os.path.join(root, "boxes"))
        for i, (name, shape, mode, input_shape) in enumerate(CASES):
            img = synth(i, shape, mode)
            Image.fromarray(img, mode).save(os.path.join(root, "JPEGImages", name + ".png"))
            cap = _Capture()
            try:
                ev.evaluate_one_img(cap, [name + ".png 1,2,3,4,0"], list(input_shape), 6, "cpu", root, os.path.join(root, "boxes"), ["c"] * 6)
            except _Captured:
                pass
            assert cap.x is not None and cap.x.shape == (1, 3) + tuple(input_shape), (name, None if cap.x is None else cap.x.shape)
            out["eval/%s/src" % name] = img
            out["eval/%s/input_shape" % name] = np.array(input_shape, np.int32)
            out["eval/%s/out" % name] = cap.x[0].astype(np.float32)
        for i, (name, shape, mode, input_shape) in enumerate(LETTERBOX):
            img = synth(100 + i, shape, mode)
            Image.fromarray(img, mode).save(os.path.join(root, "JPEGImages", name + ".png"))
            d = ds.Datasets([name + ".png 5,6,30,40,1"], list(input_shape), 6, False, root)
            image, label, info = d[0]
            assert image.shape == (3,) + tuple(input_shape) and info[1:] == (shape[1], shape[0])
            out["letterbox/%s/src" % name] = img
            out["letterbox/%s/input_shape" % name] = np.array(input_shape, np.int32)
            out["letterbox/%s/out" % name] = np.asarray(image, np.float32)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(CASES) + len(LETTERBOX), "cases")


if __name__ == "__main__":
    main()
