"""device_initializer / model_initializer with the reference's semantics (utils/initialization.py:16-60)."""
import logging

import torch

logger = logging.getLogger(__name__)

# (depth, width) presets, utils/initialization.py:53-54
_PRESETS = {"tiny": (0.33, 0.375), "s": (0.33, 0.50), "m": (0.67, 0.75), "l": (1.00, 1.00), "x": (1.33, 1.25)}


def device_initializer():
    """torch.device("cuda") when a GPU is visible (ROCm reports as cuda), else cpu -- on which ccnet_amd refuses to run."""
    if torch.cuda.is_available():
        logger.info("GPU: %s x%d", torch.cuda.get_device_name(0), torch.cuda.device_count())
        return torch.device("cuda")
    logger.warning("no GPU visible: ccnet_amd has no CPU path")
    return torch.device("cpu")


def model_initializer(model_type):
    """'tiny' | 's' | 'm' | 'l' | 'x' -> (depth, width)."""
    if model_type not in _PRESETS:
        raise KeyError("unknown model type %r (expected one of %s)" % (model_type, ", ".join(_PRESETS)))
    return _PRESETS[model_type]
