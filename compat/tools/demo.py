"""Import-path shim: `tools.demo.inference(model, image, input_shape, conf_thres, nms_thres, device)` (reference tools/demo.py:28)."""
from ccnet_amd.tools.demo import detections_to_annotations, inference, predict, preprocess  # noqa: F401
