"""tools/eval.py of the reference, tensor side on libccnet_hip: `evaluate` (eval.py:48-104) writes one detection-result
txt per image in the format utils/map.py reads (`class conf[:6] left top right bottom`, integer-truncated pixels),
`evaluate_one_img` (eval.py:107-170) does the same from image files with the resize and normalisation on the GPU, and
`map_from_dirs` scores such a directory against a ground-truth directory with the restated arithmetic of
ccnet_amd/utils/map.py (the reference's get_map, utils/map.py:307-841, minus its plotting)."""
import glob
import os

import numpy as np
import torch

from ..utils import map as cmap
from ..utils.bbox import decode_outputs, post_process


def evaluate(model, val_dataloader, input_shape, num_classes, device, save_box_path, class_names, conf_thres=0.001, nms_thres=0.6):
    """val_dataloader yields (image (B,3,H,W) array-like, _, image_info) with image_info[i] = [file_name, iw, ih]
    (utils/datasets.py).  Same file naming and line format as eval.py:81-103."""
    os.makedirs(save_box_path, exist_ok=True)
    model.eval()
    for batch in val_dataloader:
        image, _, image_info = batch
        with torch.no_grad():
            image = torch.as_tensor(np.asarray(image), dtype=torch.float32).to(device)
            outputs = decode_outputs(model(image), input_shape)
            outputs = post_process(outputs, num_classes, input_shape, image_info, conf_thres=conf_thres, nms_thres=nms_thres)
        for i, output in enumerate(outputs):
            file_name, _ = os.path.splitext(str(image_info[i][0]))
            with open(os.path.join(save_box_path, file_name + ".txt"), "w") as f:
                if output is None:
                    return
                _write_rows(f, output, class_names)


def _write_rows(f, output, class_names):
    label = np.array(output[:, 6], dtype="int32")
    conf = output[:, 4] * output[:, 5]
    for j, c in enumerate(label):
        top, left, bottom, right = output[j, :4]
        f.write("%s %s %s %s %s %s\n" % (class_names[int(c)], str(conf[j])[:6], str(int(left)), str(int(top)), str(int(right)), str(int(bottom))))


def evaluate_one_img(model, val_lines, input_shape, num_classes, device, data_path, save_box_path, class_names, conf_thres=0.001, nms_thres=0.6,
                     batch_size=1):
    """tools/eval.py:107-178: each line of val_lines names a file under <data_path>/JPEGImages; the file is decoded on the
    host (PIL, `convert("RGB")`, :129-134) and everything after that -- bicubic resize, normalisation, CHW (:135-140),
    forward, decode, post_process -- runs on the GPU (ccnet_amd/utils/preprocess.py).  batch_size > 1 groups images per
    forward; keep 1 for `network.net.Net` when results must equal the reference's image-at-a-time run (its Transformer
    attends over the batch axis)."""
    from PIL import Image

    from ..utils.preprocess import preprocess
    os.makedirs(save_box_path, exist_ok=True)
    model.eval()
    names = [line.split()[0] for line in val_lines]
    for s in range(0, len(names), max(1, int(batch_size))):
        chunk = names[s:s + max(1, int(batch_size))]
        images = []
        for name in chunk:
            with Image.open("%s/JPEGImages/%s" % (data_path, name)) as im:
                images.append(np.asarray(im.convert("RGB")))
        with torch.no_grad():
            x, image_info = preprocess(images, input_shape, device)
            outputs = decode_outputs(model(x), input_shape)
            results = post_process(outputs, num_classes, input_shape, image_info, conf_thres=conf_thres, nms_thres=nms_thres)
        for name, output in zip(chunk, results):
            with open(os.path.join(save_box_path, os.path.splitext(name)[0] + ".txt"), "w") as f:
                if output is None:
                    return
                _write_rows(f, output, class_names)


def _read_dir(path, class_names, with_conf):
    recs = {}
    for fn in sorted(glob.glob(os.path.join(path, "*.txt"))):
        rows = []
        with open(fn) as f:
            for line in f:
                parts = line.split()
                if not parts or "difficult" in parts:
                    continue  # utils/map.py does not count "difficult" boxes
                n = 5 if with_conf else 4
                name = " ".join(parts[:-n])
                if name not in class_names:
                    continue
                nums = [float(v) for v in parts[-n:]]
                rows.append((class_names.index(name),) + tuple(nums))
        recs[os.path.splitext(os.path.basename(fn))[0]] = rows
    return recs


def map_from_dirs(gt_path, dr_path, class_names, min_overlaps=(0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95)):
    """{threshold: mAP} over the images that have a ground-truth file (get_map errors out on a missing partner file;
    here a missing detection file counts as no detections)."""
    gt = _read_dir(gt_path, list(class_names), False)
    dr = _read_dir(dr_path, list(class_names), True)
    ids = sorted(gt)
    g = [gt[i] for i in ids]
    d = [dr.get(i, []) for i in ids]
    return {t: cmap.average_precision(g, d, len(class_names), t)[0] for t in min_overlaps}
