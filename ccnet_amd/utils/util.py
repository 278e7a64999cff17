"""Weight loading with the reference's contract (utils/util.py:90-108): a .pth holding
{"epoch", "model": state_dict, "ap50", "ap5095"} (tools/train.py:196-201); tensors whose shape matches are loaded,
mismatching shapes are skipped, a key the model lacks raises KeyError -- exactly as the reference behaves."""
from collections import OrderedDict

import numpy as np
import torch


def load_model_weights(model, model_path, device):
    """Returns (model, best_ap5095).  The checkpoint is read with weights_only=True (tensors and plain containers only)."""
    model_dict = model.state_dict()
    weights = torch.load(model_path, map_location=device, weights_only=True)
    weights_dict = weights["model"]
    best_ap5095 = weights["ap5095"]
    weights_dict = {k: v for k, v in weights_dict.items() if np.shape(model_dict[k]) == np.shape(v)}
    model_dict.update(weights_dict)
    model.load_state_dict(OrderedDict(model_dict))
    return model, best_ap5095


def get_classes(classes_path):
    """One class name per line (utils/util.py get_classes)."""
    with open(classes_path, encoding="utf-8") as f:
        names = [c.strip() for c in f.readlines()]
    return names, len(names)


def _read_lines(path):
    with open(path, encoding="utf-8") as f:
        return [line.strip() for line in f.readlines()]


def get_train_lines(train_path):
    """Annotation lines of the train split, stripped (utils/util.py:42-55)."""
    return _read_lines(train_path)


def get_val_lines(val_path):
    """Annotation lines of the val split, stripped (utils/util.py:58-71; tools/eval.py:28)."""
    return _read_lines(val_path)


def get_test_lines(test_path):
    """Annotation lines of the test split, stripped (utils/util.py:74-87)."""
    return _read_lines(test_path)


def get_root_path(project_name="CCNet"):
    """Project root as the reference's callers expect it (utils/util.py:244-255): the part of this file's directory up to
    and including `<project_name>/`.  The reference returns nonsense when the checkout is not inside a directory of that
    name (str.find gives -1); here CCNET_ROOT, else the repository root, is the fallback."""
    import os
    here = os.path.abspath(os.path.dirname(__file__)).replace("\\", "/")
    marker = project_name + "/"
    at = here.find(marker)
    if at >= 0:
        return here[:at + len(marker)]
    return os.environ.get("CCNET_ROOT", os.path.dirname(os.path.dirname(here)) + "/")


def draw_rectangle(results, image, iw, ih, input_shape, class_names):
    """Boxes and `class:conf%` labels drawn on `image` in place (utils/util.py:175-228; called by tools/demo.py:80).
    Drawing is the caller's OpenCV, not part of the compute path; colours come from a small hash, not the reference's table."""
    import cv2
    rows = results[0]
    for row in rows:
        top, left, bottom, right = [int(np.floor(v)) for v in row[:4]]
        top, left, bottom, right = max(0, top), max(0, left), min(int(ih), bottom), min(int(iw), right)
        cls = int(row[6])
        colour = ((cls * 67 + 29) % 256, (cls * 131 + 83) % 256, (cls * 197 + 151) % 256)
        text = "%s:%.1f%%" % (class_names[cls], float(row[4] * row[5]) * 100.0)
        (tw, th), _ = cv2.getTextSize(text, cv2.FONT_HERSHEY_SIMPLEX, 0.4, 1)
        cv2.rectangle(image, (left, top), (right, bottom), colour, thickness=2)
        cv2.rectangle(image, (left, top + 1), (left + tw + 1, top + int(1.5 * th)), tuple(int(0.7 * c) for c in colour), -1)
        cv2.putText(image, text, (left, top + th), cv2.FONT_HERSHEY_SIMPLEX, 0.4, (255, 255, 255) if sum(colour) < 384 else (0, 0, 0), thickness=1)
    return image
