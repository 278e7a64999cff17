"""tools/socket2springboot.py:inference of the reference (:33-123), source compatible, on libccnet_hip.

Differences, all outside the tensor path: the model is built and its weights are loaded once per
(model_path, num_classes, device) and kept (the reference rebuilds the model for every request, :58-61); `class_names`
may be passed instead of being read from datasets/<dataset>/classes.txt; `images` may carry already decoded BGR arrays
so that callers without OpenCV file access (and the tests) can use the entry point.  The wire protocol of the server
(:174-215: a JSON request terminated by "over", the reply `str(dict)`, "500" on any exception) is `handle_request`;
`ServerThreading` / `main` (:126-172) are the same thin socket loop around it.
"""
import collections
import json
import os
import socket
import threading

import numpy as np
import torch

from ..network.net import Net
from ..utils.util import load_model_weights
from .demo import detections_to_annotations, predict, preprocess

_models = collections.OrderedDict()  # LRU, newest last
_lock = threading.Lock()
MAX_MODELS = max(int(os.environ.get("CCNET_MAX_MODELS", "4")), 1)
# input_shape comes from the client: every distinct (H, W) compiles a plan with its own activation buffers, so sizes are
# validated (multiples of 32, bounded side) before anything is allocated; Net itself keeps at most CCNET_MAX_PLANS plans
MAX_INPUT_SIDE = int(os.environ.get("CCNET_MAX_INPUT_SIDE", "2048"))


def check_input_shape(input_shape):
    if (not isinstance(input_shape, (list, tuple)) or len(input_shape) != 2
            or not all(isinstance(v, int) and not isinstance(v, bool) for v in input_shape)):
        raise ValueError("input_shape must be [height, width] integers")
    h, w = input_shape
    if h < 32 or w < 32 or h % 32 or w % 32 or h > MAX_INPUT_SIDE or w > MAX_INPUT_SIDE:
        raise ValueError("input_shape %r: sides must be multiples of 32 in [32, %d]" % (list(input_shape), MAX_INPUT_SIDE))
    return [h, w]


def get_classes(path):
    """utils/util.py:get_classes: one class name per line -> (names, count)."""
    with open(path, encoding="utf-8") as f:
        names = [line.strip() for line in f.readlines() if line.strip()]
    return names, len(names)


def _model_for(model_path, num_classes, device):
    """One loaded model per (checkpoint file, its mtime and size, num_classes, device): the reference reloads the file on
    every request (:58-61), so a checkpoint overwritten in place must be picked up -- the stat is part of the key."""
    path = os.path.abspath(model_path) if model_path else None
    stamp = None
    if path:
        st = os.stat(path)
        stamp = (st.st_mtime_ns, st.st_size)
    key = (path, stamp, num_classes, str(device))
    with _lock:
        model = _models.get(key)
        if model is None:
            for k in [k for k in _models if k[0] == path and k[2:] == key[2:]]:
                del _models[k]  # the same file with an older stamp
            while len(_models) >= MAX_MODELS:
                _models.popitem(last=False)
            model = Net(depth=0.33, width=0.5, num_classes=num_classes, act="silu")  # :58
            if model_path:
                model, _ = load_model_weights(model, model_path, device)
            model.to(device)
            model.eval()
            _models[key] = model
        else:
            _models.move_to_end(key)
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
    input_shape = check_input_shape(input_shape)
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


def _reply_text(res):
    """str(dict) as the reference sends it (:210).  Its "conf" values are numpy float32 scalars, which numpy 1.x (the
    reference's era) prints as their shortest decimal form; keep that text whatever numpy is installed."""
    if not res:
        return str(res)
    anns = [dict(a, conf=float(str(np.float32(a["conf"])))) for a in res.get("annotations", [])]
    return str({"image": res.get("image", []), "annotations": anns})


def handle_request(msg, **kwargs):
    """One request of the protocol (:181-213): `msg` is the received text ending in "over"; returns the reply text --
    `str(result dict)`, or "500" when anything raises.  kwargs go to `inference` (class_names, images, device, ...)."""
    try:
        text = msg
        if text.strip().endswith("over"):
            text = text[:-4]  # :190 cuts exactly four characters, as received
        req = json.loads(text)
        image_paths = [p for p in req["image_path"]]
        res = inference(req["model_name"], req["model_path"], req["dataset"], req["input_shape"], req["conf_thres"], req["nms_thres"], image_paths,
                        **kwargs)
        return _reply_text(res)
    except Exception:
        return "500"


class ServerThreading(threading.Thread):
    """:158-216: one thread per connection; reads until the text ends in "over", answers once, closes."""

    def __init__(self, client_socket, address, recvsize=1024 * 1024, encoding="utf-8", **kwargs):
        threading.Thread.__init__(self)
        self._socket = client_socket
        self._address = address
        self._recvsize = recvsize
        self._encoding = encoding
        self._kwargs = kwargs

    def run(self):
        try:
            msg = ""
            while True:
                rec = self._socket.recv(self._recvsize)
                msg += rec.decode(self._encoding)
                if msg.strip().endswith("over") or not rec:
                    break
            self._socket.send(handle_request(msg, **self._kwargs).encode(self._encoding))
        except Exception:
            try:
                self._socket.send("500".encode(self._encoding))
            except OSError:
                pass
        finally:
            self._socket.close()


def main(host=None, port=12345, **kwargs):
    """:126-155.  Unlike the reference, models stay loaded between requests (`_model_for`)."""
    server_socket = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    server_socket.bind((host if host is not None else socket.gethostname(), port))
    server_socket.listen(5)
    try:
        while True:
            client_socket, address = server_socket.accept()
            ServerThreading(client_socket, address, **kwargs).start()
    finally:
        server_socket.close()
