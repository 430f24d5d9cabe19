"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's pedestrian AP / log-average-miss-rate
evaluator (other_utils/metrics.py:7-165).  Pinned by tests/golden/ap.npz, which was produced by importing the
reference's compute_ap_lamr in the build container (tests/golden/make_golden.py ap).  Only tests/ may import this.

Deliberately scalar: one prediction, one ground-truth box, one curve point at a time.
"""
import math

import numpy as np

IOU_THRESHOLD = 0.5        # metrics.py:4


def voc_ap(recall, precision):
    """metrics.py:7-28: area under the precision envelope, sentinels (0,0) and (1,0)."""
    r = [0.0] + [float(v) for v in recall] + [1.0]
    p = [0.0] + [float(v) for v in precision] + [0.0]
    for i in range(len(p) - 2, -1, -1):          # running maximum from the right
        if p[i + 1] > p[i]:
            p[i] = p[i + 1]
    area = 0.0
    terms = []
    for i in range(1, len(r)):
        if r[i] != r[i - 1]:
            terms.append((r[i] - r[i - 1]) * p[i])
    return float(np.sum(np.array(terms, dtype=np.float64))) if terms else area


def log_average_miss_rate(recall, fp_cumsum, num_imgs):
    """metrics.py:31-59: miss rate sampled at 9 log-spaced FPPI points in [1e-2, 1], geometric mean."""
    fppi = np.asarray(fp_cumsum) / float(num_imgs)
    mr = 1 - np.asarray(recall)
    xs = [-1.0] + [float(v) for v in fppi]
    ys = [1.0] + [float(v) for v in mr]
    picked = []
    for ref in np.logspace(-2.0, 0.0, num=9):
        last = 0
        for j, v in enumerate(xs):
            if v <= ref:
                last = j
        picked.append(ys[last])
    lamr = math.exp(np.mean(np.log(np.maximum(1e-10, np.array(picked)))))
    return lamr, fppi, mr


def iou_plus_one(pred_xyxy, gt_xyxy_int):
    """metrics.py:62-78: pixel-inclusive IoU (+1 on every extent) of one float32 box against one int32 box,
    with the reference's numpy type promotion (float32 area for the prediction, float64 for the rest)."""
    a1 = (pred_xyxy[2] - pred_xyxy[0] + 1) * (pred_xyxy[3] - pred_xyxy[1] + 1)                # float32
    a2 = (gt_xyxy_int[2] - gt_xyxy_int[0] + 1) * (gt_xyxy_int[3] - gt_xyxy_int[1] + 1)        # int32
    w = np.float64(min(np.float64(pred_xyxy[2]), np.float64(gt_xyxy_int[2]))) - max(np.float64(pred_xyxy[0]), np.float64(gt_xyxy_int[0])) + 1
    h = np.float64(min(np.float64(pred_xyxy[3]), np.float64(gt_xyxy_int[3]))) - max(np.float64(pred_xyxy[1]), np.float64(gt_xyxy_int[1])) + 1
    inter = min(max(w, 0.0), 1e5) * min(max(h, 0.0), 1e5)
    return inter / (np.float64(a1) + np.float64(a2) - inter)


def compute_ap_lamr(preds, labels, shapes):
    """metrics.py:81-165.  preds: list of {img_id, conf, bbox float32 xyxy} in descending confidence;
    labels: per image float32 [k,5] (flag, xc, yc, w, h) relative; shapes: per image (w, h).
    Unlike the reference this does not modify `labels` (the reference converts them in place, :103-110)."""
    gts, used = [], []
    for lab, (w, h) in zip(labels, shapes):
        lab = np.array(lab, dtype=np.float32, copy=True)
        lab[:, [1, 3]] *= w                                    # :105-110, all in float32
        lab[:, [2, 4]] *= h
        lab[:, 1] -= lab[:, 3] / 2
        lab[:, 2] -= lab[:, 4] / 2
        lab[:, 3] = lab[:, 1] + lab[:, 3]
        lab[:, 4] = lab[:, 2] + lab[:, 4]
        gts.append(lab[:, 1:].astype(np.int32))                # :136 truncation toward zero
        used.append([False] * lab.shape[0])
    nt = sum(g.shape[0] for g in gts)
    tp, fp = [], []
    for pr in preds:
        g = gts[pr["img_id"]]
        best, best_iou = 0, -1.0
        for k in range(g.shape[0]):
            v = iou_plus_one(pr["bbox"], g[k])
            if v > best_iou:                                   # np.argmax: first maximum
                best, best_iou = k, v
        hit = best_iou >= IOU_THRESHOLD and not used[pr["img_id"]][best]      # :140-152 duplicates are FP
        if hit:
            used[pr["img_id"]][best] = True
        tp.append(1 if hit else 0)
        fp.append(0 if hit else 1)
    tpc, fpc = np.cumsum(np.array(tp, dtype=np.int32)), np.cumsum(np.array(fp, dtype=np.int32))
    recall = tpc / nt
    precision = tpc / (tpc + fpc)
    ap = voc_ap(recall, precision)
    lamr, fppi, mr = log_average_miss_rate(recall, fpc, len(labels))
    return {"recall": recall, "precision": precision, "fppi": fppi, "mr": mr, "ap": ap, "lamr": lamr}
