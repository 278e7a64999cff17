"""In-memory restatement of the reference's mAP arithmetic (utils/map.py), used to gate the bf16 path at detection level.

The reference goes through text files: tools/eval.py:89-103 writes one line per detection
(`class conf[:6] int(left) int(top) int(right) int(bottom)`, conf = obj_conf * class_conf), utils/map.py:450 sorts the
detections of a class by confidence, :496-531 matches each greedily to the ground-truth box of its image and class with
the largest overlap (pixel-inclusive `+1` widths, :503-508; a ground truth can be used once, a second match is a false
positive), :591-609 turns the cumulative tp/fp into precision/recall and :112-153 (`voc_ap`) integrates them.  The same
steps run here on lists; nothing is read from or written to disk.
"""
import numpy as np


def voc_ap(rec, prec):
    """Area under the monotone precision envelope (utils/map.py:112-153).  rec, prec: sequences; returns ap."""
    mrec = [0.0] + [float(r) for r in rec] + [1.0]
    mpre = [0.0] + [float(p) for p in prec] + [0.0]
    for i in range(len(mpre) - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    ap = 0.0
    for i in range(1, len(mrec)):
        if mrec[i] != mrec[i - 1]:
            ap += (mrec[i] - mrec[i - 1]) * mpre[i]
    return ap


def detections_to_records(outputs, conf_digits=True):
    """post_process output (list over images of (n,7) [top,left,bottom,right,obj,cls_conf,cls]) -> per-image lists of
    (class_id, conf, left, top, right, bottom) exactly as tools/eval.py:89-103 prints them (integer-truncated pixels,
    confidence cut to the first 6 characters of its str())."""
    recs = []
    for out in outputs:
        rows = []
        if out is not None:
            for r in np.asarray(out):
                conf = float(r[4]) * float(r[5]) if not conf_digits else float(str(np.float32(r[4]) * np.float32(r[5]))[:6])
                top, left, bottom, right = r[:4]
                rows.append((int(r[6]), conf, int(left), int(top), int(right), int(bottom)))
        recs.append(rows)
    return recs


def average_precision(gt, dets, num_classes, min_overlap=0.5):
    """gt: per-image list of (class_id, left, top, right, bottom); dets: per-image list of
    (class_id, conf, left, top, right, bottom).  Returns (mAP over the classes present in gt, {class: ap})."""
    aps = {}
    for c in range(num_classes):
        gts = [[g[1:] for g in img if g[0] == c] for img in gt]
        n_gt = sum(len(g) for g in gts)
        if n_gt == 0:
            continue  # utils/map.py only iterates over ground-truth classes
        used = [[False] * len(g) for g in gts]
        cand = [(d[1], i, d[2:]) for i, img in enumerate(dets) for d in img if d[0] == c]
        cand.sort(key=lambda t: -t[0])  # stable, like list.sort(reverse=True) on the confidence (:450)
        tp, fp = [], []
        for _, i, bb in cand:
            ovmax, match = -1.0, -1
            for k, bbgt in enumerate(gts[i]):
                iw = min(bb[2], bbgt[2]) - max(bb[0], bbgt[0]) + 1
                ih = min(bb[3], bbgt[3]) - max(bb[1], bbgt[1]) + 1
                if iw > 0 and ih > 0:
                    ua = (bb[2] - bb[0] + 1) * (bb[3] - bb[1] + 1) + (bbgt[2] - bbgt[0] + 1) * (bbgt[3] - bbgt[1] + 1) - iw * ih
                    ov = iw * ih / ua
                    if ov > ovmax:
                        ovmax, match = ov, k
            if ovmax >= min_overlap and not used[i][match]:
                used[i][match] = True
                tp.append(1); fp.append(0)
            else:
                tp.append(0); fp.append(1)
        ctp, cfp = np.cumsum(tp), np.cumsum(fp)
        rec = [float(t) / max(n_gt, 1) for t in ctp]
        prec = [float(t) / max(int(f) + int(t), 1) for t, f in zip(ctp, cfp)]
        aps[c] = voc_ap(rec, prec)
    return (sum(aps.values()) / len(aps) if aps else 0.0), aps


def map_50_95(gt, dets, num_classes):
    """COCO-style mean over IoU 0.50:0.05:0.95 of the VOC mAP (tools/eval.py:186 sweeps the same thresholds)."""
    vals = [average_precision(gt, dets, num_classes, t)[0] for t in (0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95)]
    return float(np.mean(vals)), vals


def match_boxes(ref, got, iou_thr=0.9):
    """Box-level agreement of two post_process outputs (lists over images of (n,7) rows [top,left,bottom,right,obj,cls_conf,cls]):
    each reference row is matched greedily (best IoU first come, a `got` row is used once) to a row of the same image and
    class.  Returns dict(n_ref, n_got, matched = reference rows with a same-class partner at IoU >= iou_thr, mean_iou of the
    matched pairs, max_score_diff over the matched pairs).  The bf16 detection-level gate (tests/test_bf16_gate_gpu.py)."""
    n_ref = n_got = matched = 0
    ious, sdiff = [], 0.0
    for a, b in zip(ref, got):
        a = np.zeros((0, 7), np.float32) if a is None else np.asarray(a, dtype=np.float64)
        b = np.zeros((0, 7), np.float32) if b is None else np.asarray(b, dtype=np.float64)
        n_ref += a.shape[0]
        n_got += b.shape[0]
        if a.shape[0] == 0 or b.shape[0] == 0:
            continue
        t = np.maximum(a[:, None, 0], b[None, :, 0]); l = np.maximum(a[:, None, 1], b[None, :, 1])
        bo = np.minimum(a[:, None, 2], b[None, :, 2]); r = np.minimum(a[:, None, 3], b[None, :, 3])
        inter = np.clip(bo - t, 0, None) * np.clip(r - l, 0, None)
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        with np.errstate(invalid="ignore", divide="ignore"):
            iou = inter / (area_a[:, None] + area_b[None, :] - inter)
        iou = np.where(a[:, None, 6] == b[None, :, 6], np.nan_to_num(iou), -1.0)
        used = np.zeros(b.shape[0], bool)
        for i in range(a.shape[0]):
            row = np.where(used, -1.0, iou[i])
            j = int(row.argmax())
            if row[j] >= iou_thr:
                used[j] = True
                matched += 1
                ious.append(float(row[j]))
                sdiff = max(sdiff, abs(a[i, 4] * a[i, 5] - b[j, 4] * b[j, 5]))
    return {"n_ref": n_ref, "n_got": n_got, "matched": matched, "mean_iou": float(np.mean(ious)) if ious else 0.0,
            "max_score_diff": float(sdiff)}
