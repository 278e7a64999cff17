"""tools/socket2springboot.py:inference of the reference (:33-123), source compatible, on libccnet_hip.

Differences, all outside the tensor path: the model is built and its weights are loaded once per
(model_path, num_classes, device) and kept (the reference rebuilds the model for every request, :58-61); `class_names`
may be passed instead of being read from datasets/<dataset>/classes.txt; `images` may carry already decoded BGR arrays
so that callers without OpenCV file access (and the tests) can use the entry point.  The socket server itself
(:126-215) is control plane and is not rebuilt.
"""
import os
import threading

import numpy as np
import torch

from ..network.net import Net
from ..utils.util import load_model_weights
from .demo import detections_to_annotations, predict, preprocess

_models = {}
_lock = threading.Lock()


def get_classes(path):
    """utils/util.py:get_classes: one class name per line -> (names, count)."""
    with open(path, encoding="utf-8") as f:
        names = [line.strip() for line in f.readlines() if line.strip()]
    return names, len(names)


def _model_for(model_path, num_classes, device):
    key = (os.path.abspath(model_path) if model_path else None, num_classes, str(device))
    with _lock:
        model = _models.get(key)
        if model is None:
            model = Net(depth=0.33, width=0.5, num_classes=num_classes, act="silu")  # :58
            if model_path:
                model, _ = load_model_weights(model, model_path, device)
            model.to(device)
            model.eval()
            _models[key] = model
        return model


def inference(model_name, model_path, dataset, input_shape, conf_thres, nms_thres, image_paths, class_names=None, images=None,
              device="cuda", datasets_root=None):
    """Returns {"image": [{"image_path", "image_id"}...], "annotations": [{"image_id", "box": [l,t,r,b],
    "predicted_class", "conf"}...]} for model_name == "Net", {} otherwise (:48, :121-122)."""
    if model_name != "Net":
        return {}
    if class_names is None:
        root = datasets_root or os.getcwd()
        class_names, _ = get_classes(os.path.join(root, "datasets/%s/classes.txt" % dataset))
    num_classes = len(class_names)
    model = _model_for(model_path, num_classes, device)
    msg = {"image": [], "annotations": []}
    for image_id, image_path in enumerate(image_paths):
        if images is not None:
            image = images[image_id]
        else:
            import cv2
            image = cv2.imread(image_path)
        msg["image"].append({"image_path": image_path, "image_id": image_id})
        if image is None or getattr(image, "ndim", 0) != 3 or image.shape[2] != 3:
            continue  # :69-73, "not a 3 channel picture"
        data, image_info = preprocess(np.asarray(image), input_shape)
        results = predict(model, torch.from_numpy(data).to(device), num_classes, input_shape, image_info, conf_thres, nms_thres)
        msg["annotations"].extend(detections_to_annotations(results[0], image_id, class_names))
    return msg
