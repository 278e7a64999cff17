"""The predict entry points of the reference's callers, kept source compatible:
tools/demo.py:inference(model, image, input_shape, conf_thres, nms_thres, device) and the per-image body of
tools/socket2springboot.py:inference (:95-119).  The tensor side (model -> decode_outputs -> post_process) runs on
libccnet_hip; image decode / resize / drawing stay with the caller's OpenCV, as in the reference."""
import numpy as np
import torch

from ..utils.bbox import decode_outputs, post_process

_MEAN = np.array([0.485, 0.456, 0.406])
_STD = np.array([0.229, 0.224, 0.225])


def preprocess(image, input_shape):
    """tools/demo.py:49-60: bicubic resize to (h, w) (no letterbox), /255, mean/std, HWC->CHW, add batch.  Needs cv2."""
    import cv2  # not a dependency of the compute path; only this helper needs it
    ih, iw = image.shape[0], image.shape[1]
    h, w = input_shape
    data = cv2.resize(image, (w, h), interpolation=cv2.INTER_CUBIC)
    data = np.array(data, dtype="float32")
    data /= 255.0
    data -= _MEAN
    data /= _STD
    return np.expand_dims(np.transpose(data, (2, 0, 1)), 0).astype(np.float32), [[0, iw, ih]]


def predict(model, images, num_classes, input_shape, image_info, conf_thres, nms_thres):
    """The three hot-path statements shared by every caller (tools/demo.py:68-73, socket2springboot.py:95-100,
    eval.py:75-80): images is a float32 (B,3,H,W) CUDA tensor."""
    with torch.no_grad():
        outputs = model(images)
        outputs = decode_outputs(outputs, input_shape)
        return post_process(outputs, num_classes, input_shape, image_info, conf_thres=conf_thres, nms_thres=nms_thres)


# The reference reads these two as module globals inside inference() (tools/demo.py:72,80; its __main__ sets them, :92).  A
# caller that imports the module may set them here the same way; otherwise they are taken from the model.
num_classes = None
class_names = None


def detect(model, image, input_shape, conf_thres, nms_thres, device, num_classes=None):
    """inference() without the drawing: returns (results, image_info) -- results is post_process' list with one
    (n, 7) float32 array [top, left, bottom, right, obj_conf, class_conf, class_id] for the one image."""
    data, image_info = preprocess(image, input_shape)
    images = torch.from_numpy(data).to(device)
    if num_classes is None:
        num_classes = globals()["num_classes"] or model._cfg["nc"]
    return predict(model, images, num_classes, input_shape, image_info, conf_thres, nms_thres), image_info


def inference(model, image, input_shape, conf_thres, nms_thres, device, num_classes=None, class_names=None, draw=None):
    """tools/demo.py:28-81, same six positional arguments, same return value: the IMAGE, with the detections drawn on it
    in place by utils.util.draw_rectangle (:80) -- or untouched when post_process returns None for it (:77-79).  The
    reference takes num_classes / class_names from module globals; here keyword arguments, then this module's globals of the
    same names, then the model's own class count and the class ids as names.  `draw` replaces draw_rectangle (same
    arguments).  detect() is the variant that returns the detections."""
    results, image_info = detect(model, image, input_shape, conf_thres, nms_thres, device, num_classes)
    if results[0] is None:
        return image
    if class_names is None:
        class_names = globals()["class_names"] or [str(i) for i in range(model._cfg["nc"])]
    if draw is None:
        from ..utils.util import draw_rectangle as draw
    draw(results, image, image_info[0][1], image_info[0][2], input_shape, class_names)
    return image


def detections_to_annotations(results_i, image_id, class_names):
    """Row -> dict conversion of tools/socket2springboot.py:107-119."""
    out = []
    if results_i is None:
        return out
    for row in results_i:
        top, left, bottom, right = [int(v) for v in row[:4]]
        out.append({"image_id": image_id, "box": [left, top, right, bottom], "predicted_class": class_names[int(row[6])],
                    "conf": float(row[4] * row[5])})
    return out
