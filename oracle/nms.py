"""Oracle (test infrastructure): box decode helpers and non-maximum suppression on the CPU.

Restates reference build_utils/utils.py:40-57 (xyxy2xywh / xywh2xyxy), :60-92 (scale_coords /
clip_coords), :387-464 (non_max_suppression) and the third-party `torchvision.ops.nms` called at
utils.py:448.

PARITY UNPINNED for `torchvision.ops.nms`: the reference neither vendors nor version-pins
torchvision (requirements.txt:1-7 does not list it) and it is absent from this image.  The
restatement below follows torchvision's documented / published CPU algorithm:
  * process boxes in order of decreasing score (stable: equal scores keep index order),
  * keep a box unless an already kept box has IoU > threshold with it (strictly greater),
  * IoU = inter / (area_a + area_b - inter), inter = max(0, xx2-xx1) * max(0, yy2-yy1), no +1,
  * all arithmetic in float32,
and is anchored by the hand-computed known-answer cases in tests/test_oracle_nms.py.
"""
import numpy as np
import torch


def nms_numpy(boxes, scores, iou_thres):
    """Greedy NMS with torchvision semantics.  boxes [n,4] xyxy, scores [n]; returns kept indices
    (int64) in decreasing-score order."""
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)                       # float32
    order = np.argsort(-scores, kind="stable")
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_thres)
    for pos in range(n):
        i = order[pos]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[pos + 1:]
        if rest.size == 0:
            break
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def xywh2xyxy(x):
    """utils.py:50-57: centre/size -> corners (x - w/2, y - h/2, x + w/2, y + h/2)."""
    y = torch.zeros_like(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def xyxy2xywh(x):
    """utils.py:40-47."""
    y = torch.zeros_like(x)
    y[:, 0] = (x[:, 0] + x[:, 2]) / 2
    y[:, 1] = (x[:, 1] + x[:, 3]) / 2
    y[:, 2] = x[:, 2] - x[:, 0]
    y[:, 3] = x[:, 3] - x[:, 1]
    return y


def clip_coords(boxes, img_shape):
    """utils.py:87-92 (in place)."""
    boxes[:, 0].clamp_(0, img_shape[1])
    boxes[:, 1].clamp_(0, img_shape[0])
    boxes[:, 2].clamp_(0, img_shape[1])
    boxes[:, 3].clamp_(0, img_shape[0])


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """utils.py:60-84: undo the letterbox (in place on coords)."""
    if ratio_pad is None:
        gain = max(img1_shape) / max(img0_shape)
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    clip_coords(coords, img0_shape)
    return coords


def non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6, multi_label=True, classes=None,
                        agnostic=False, max_num=100, return_indices=False):
    """utils.py:387-464 without the 10 s wall-clock bail-out (:461-462, nondeterministic).

    prediction [B, N, 5+nc] (cx, cy, w, h, obj, cls...) -> list of [n,6] (x1,y1,x2,y2,conf,cls) or
    None per image.  With return_indices also returns, per image, the row index into the N
    candidates of every output detection (the keep-set the GPU path must reproduce exactly)."""
    min_wh, max_wh = 2, 4096                                      # :399
    nc = prediction[0].shape[1] - 5                               # :403
    multi_label = multi_label and nc > 1                          # :404
    out = [None] * prediction.shape[0]
    out_idx = [None] * prediction.shape[0]
    for xi, x in enumerate(prediction):
        rows = torch.arange(x.shape[0])
        m = x[:, 4] > conf_thres                                  # :408
        x, rows = x[m], rows[m]
        m = ((x[:, 2:4] > min_wh) & (x[:, 2:4] < max_wh)).all(1)  # :409
        x, rows = x[m], rows[m]
        if not x.shape[0]:
            continue
        x = x.clone()
        x[..., 5:] *= x[..., 4:5]                                 # :416 conf = obj * cls
        box = xywh2xyxy(x[:, :4])                                 # :419
        if multi_label:                                           # :422-424
            i, j = (x[:, 5:] > conf_thres).nonzero(as_tuple=False).t()
            x = torch.cat((box[i], x[i, j + 5].unsqueeze(1), j.float().unsqueeze(1)), 1)
            rows = rows[i]
        else:                                                     # :425-427
            conf, j = x[:, 5:].max(1)
            keepm = conf > conf_thres
            x = torch.cat((box, conf.unsqueeze(1), j.float().unsqueeze(1)), 1)[keepm]
            rows = rows[keepm]
            j = j[keepm]
        if classes:                                               # :430-431
            cm = (x[:, 5:6] == torch.tensor(classes, dtype=x.dtype)).any(1)
            x, rows = x[cm], rows[cm]
        n = x.shape[0]
        if not n:
            continue
        c = x[:, 5] * 0 if agnostic else x[:, 5]                  # :446
        boxes, scores = x[:, :4].clone() + c.view(-1, 1) * max_wh, x[:, 4]   # :447
        keep = torch.as_tensor(nms_numpy(boxes.numpy(), scores.numpy(), iou_thres), dtype=torch.long)
        keep = keep[:max_num]                                     # :449
        out[xi] = x[keep]
        out_idx[xi] = rows[keep]
    return (out, out_idx) if return_indices else out
