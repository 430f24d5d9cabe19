"""Pedestrian AP@0.5 / log-average miss rate from NMS keep-sets -- same API and results as the reference's
other_utils/metrics.py (voc_ap :7, log_average_miss_rate :31, box_iou :62, compute_ap_lamr :81), so the second half
of the headline metric ("eval AP within +-0.1") can be computed on the GPU box from this framework's detections
without pycocotools.  Host-side numpy by design: the inputs are <= 100 detections per image after NMS.

Differences from the reference, both deliberate:
  * `labels` is not modified (the reference rewrites the caller's arrays in place, so it can be called once only);
  * the per-prediction IoU row and the curve post-processing are vectorised; the greedy matching stays sequential
    because a ground truth claimed by a higher-scoring detection turns later matches into false positives.
"""
import math

import numpy as np

IOU_THRESHOLD = 0.5


def voc_ap(recall, precision):
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]             # precision envelope
    step = np.nonzero(mrec[1:] != mrec[:-1])[0] + 1
    return np.sum((mrec[step] - mrec[step - 1]) * mpre[step])


def log_average_miss_rate(recall, fp_cumsum, num_imgs):
    fppi = fp_cumsum / float(num_imgs)
    mr = 1 - recall
    x = np.concatenate(([-1.0], fppi))
    y = np.concatenate(([1.0], mr))
    # fppi is non-decreasing: the last index with x <= ref is a right-sided binary search
    pos = np.searchsorted(x, np.logspace(-2.0, 0.0, num=9), side="right") - 1
    lamr = math.exp(np.mean(np.log(np.maximum(1e-10, y[pos]))))
    return lamr, fppi, mr


def box_iou(box1, box2):
    """box1 [1,4] (or [m,4]) xyxy, box2 [n,4] xyxy -> [m,n]; extents are pixel-inclusive (+1)."""
    area1 = (box1[:, 2] - box1[:, 0] + 1) * (box1[:, 3] - box1[:, 1] + 1)
    area2 = (box2[:, 2] - box2[:, 0] + 1) * (box2[:, 3] - box2[:, 1] + 1)
    wh = np.minimum(box1[:, None, 2:], box2[:, 2:]) - np.maximum(box1[:, None, :2], box2[:, :2]) + 1
    inter = np.prod(np.clip(wh, 0, 1e5), axis=2)
    return inter / (area1[:, None] + area2 - inter)


def _absolute_xyxy(labels, shapes):
    assert len(labels) == len(shapes), "label's len != shape's len"
    out = []
    for lab, (w, h) in zip(labels, shapes):
        lab = np.array(lab, dtype=np.float32, copy=True)
        lab[:, [1, 3]] *= w
        lab[:, [2, 4]] *= h
        lab[:, 1] -= lab[:, 3] / 2
        lab[:, 2] -= lab[:, 4] / 2
        lab[:, 3] += lab[:, 1]
        lab[:, 4] += lab[:, 2]
        out.append(lab)
    return out


def compute_ap_lamr(preds, labels, shapes):
    """preds: list of dicts {img_id, conf, bbox (float32 xyxy)} sorted by descending confidence;
    labels: per image [k,5] (class/flag, xc, yc, w, h) relative; shapes: per image (w, h).
    Returns {'recall','precision','fppi','mr','ap','lamr'}."""
    gts = [lab[:, 1:].astype(np.int32) for lab in _absolute_xyxy(labels, shapes)]
    claimed = [np.zeros(g.shape[0], dtype=bool) for g in gts]
    nt = sum(g.shape[0] for g in gts)
    tp = np.zeros(len(preds), dtype=np.int32)
    for n, pr in enumerate(preds):
        i = pr["img_id"]
        iou = box_iou(pr["bbox"].reshape(-1, 4), gts[i])[0]
        k = int(np.argmax(iou))
        if iou[k] >= IOU_THRESHOLD and not claimed[i][k]:
            claimed[i][k] = True
            tp[n] = 1
    tp_cum = np.cumsum(tp)
    fp_cum = np.cumsum(1 - tp)
    recall = tp_cum / nt
    precision = tp_cum / (tp_cum + fp_cum)
    ap = voc_ap(recall, precision)
    lamr, fppi, mr = log_average_miss_rate(recall, fp_cum, len(labels))
    return {"recall": recall, "precision": precision, "fppi": fppi, "mr": mr, "ap": ap, "lamr": lamr}
