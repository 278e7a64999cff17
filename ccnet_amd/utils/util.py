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
