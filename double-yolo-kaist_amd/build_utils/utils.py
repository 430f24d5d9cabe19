"""Loss, target assignment, IoU, NMS and box utilities of the reference's build_utils/utils.py on the
MI355X HIP path (same names and signatures; reference build_utils/utils.py:24-469).  Importing this
module needs neither cv2 nor torchvision."""
import glob
import math
import os
import random

import numpy as np
import torch
import torch.nn as nn


def init_seeds(seed=0):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def check_file(file):
    """return `file` if it exists, else the first recursive glob match (reference utils.py:30-37)"""
    if os.path.isfile(file):
        return file
    files = glob.glob("./**/" + file, recursive=True)
    assert len(files), "File Not Found: %s" % file
    return files[0]


def get_yolo_layers(model):
    return [i for i, d in enumerate(model.module_defs) if d["type"] == "yolo"]


# ------------------------------------------------------------------------------------------------
# box helpers (reference utils.py:40-171): tiny elementwise glue on whatever device the tensors are
def xyxy2xywh(x):
    y = torch.zeros_like(x) if isinstance(x, torch.Tensor) else np.zeros_like(x)
    y[:, 0] = (x[:, 0] + x[:, 2]) / 2
    y[:, 1] = (x[:, 1] + x[:, 3]) / 2
    y[:, 2] = x[:, 2] - x[:, 0]
    y[:, 3] = x[:, 3] - x[:, 1]
    return y


def xywh2xyxy(x):
    y = torch.zeros_like(x) if isinstance(x, torch.Tensor) else np.zeros_like(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def clip_coords(boxes, img_shape):
    """clamp xyxy boxes to (height, width) in place"""
    boxes[:, 0].clamp_(0, img_shape[1])
    boxes[:, 1].clamp_(0, img_shape[0])
    boxes[:, 2].clamp_(0, img_shape[1])
    boxes[:, 3].clamp_(0, img_shape[0])


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """map xyxy boxes from the letterboxed network input back to the original image (in place)"""
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


def bbox_iou(box1, box2, x1y1x2y2=True, GIoU=False, DIoU=False, CIoU=False):
    """IoU / GIoU / DIoU / CIoU of box1 [4,n] against box2 [n,4] (reference utils.py:95-138)"""
    box2 = box2.t()
    if x1y1x2y2:
        b1_x1, b1_y1, b1_x2, b1_y2 = box1[0], box1[1], box1[2], box1[3]
        b2_x1, b2_y1, b2_x2, b2_y2 = box2[0], box2[1], box2[2], box2[3]
    else:
        b1_x1, b1_x2 = box1[0] - box1[2] / 2, box1[0] + box1[2] / 2
        b1_y1, b1_y2 = box1[1] - box1[3] / 2, box1[1] + box1[3] / 2
        b2_x1, b2_x2 = box2[0] - box2[2] / 2, box2[0] + box2[2] / 2
        b2_y1, b2_y2 = box2[1] - box2[3] / 2, box2[1] + box2[3] / 2
    inter = (torch.min(b1_x2, b2_x2) - torch.max(b1_x1, b2_x1)).clamp(0) * \
            (torch.min(b1_y2, b2_y2) - torch.max(b1_y1, b2_y1)).clamp(0)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1
    union = (w1 * h1 + 1e-16) + w2 * h2 - inter
    iou = inter / union
    if GIoU or DIoU or CIoU:
        cw = torch.max(b1_x2, b2_x2) - torch.min(b1_x1, b2_x1)
        ch = torch.max(b1_y2, b2_y2) - torch.min(b1_y1, b2_y1)
        if GIoU:
            c_area = cw * ch + 1e-16
            return iou - (c_area - union) / c_area
        c2 = cw ** 2 + ch ** 2 + 1e-16
        rho2 = ((b2_x1 + b2_x2) - (b1_x1 + b1_x2)) ** 2 / 4 + ((b2_y1 + b2_y2) - (b1_y1 + b1_y2)) ** 2 / 4
        if DIoU:
            return iou - rho2 / c2
        v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
        with torch.no_grad():
            alpha = v / (1 - iou + v)
        return iou - (rho2 / c2 + v * alpha)
    return iou


def box_iou(box1, box2):
    """pairwise IoU of xyxy boxes [N,4] x [M,4] -> [N,M]"""
    area1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    area2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    inter = (torch.min(box1[:, None, 2:], box2[:, 2:]) - torch.max(box1[:, None, :2], box2[:, :2])).clamp(0).prod(2)
    return inter / (area1[:, None] + area2 - inter)


def wh_iou(wh1, wh2):
    """IoU of boxes sharing a corner: wh1 [n,2], wh2 [m,2] -> [n,m]"""
    wh1 = wh1[:, None]
    wh2 = wh2[None]
    inter = torch.min(wh1, wh2).prod(2)
    return inter / (wh1.prod(2) + wh2.prod(2) - inter)


class FocalLoss(nn.Module):
    """focal-loss wrapper around a BCEWithLogitsLoss (reference utils.py:174-201; fl_gamma is 0 in both
    reference hyp files, so the HIP loss never routes through it)"""

    def __init__(self, loss_fcn, gamma=1.5, alpha=0.25):
        super(FocalLoss, self).__init__()
        self.loss_fcn = loss_fcn
        self.gamma = gamma
        self.alpha = alpha
        self.reduction = loss_fcn.reduction
        self.loss_fcn.reduction = "none"

    def forward(self, pred, true):
        loss = self.loss_fcn(pred, true)
        pred_prob = torch.sigmoid(pred)
        p_t = true * pred_prob + (1 - true) * (1 - pred_prob)
        loss = loss * (true * self.alpha + (1 - true) * (1 - self.alpha)) * (1.0 - p_t) ** self.gamma
        if self.reduction == "mean":
            return loss.mean()
        if self.reduction == "sum":
            return loss.sum()
        return loss


def smooth_BCE(eps=0.1):
    return 1.0 - 0.5 * eps, 0.5 * eps


# ------------------------------------------------------------------------------------------------
# hot path: loss, target assignment and NMS run as HIP kernels (dyk/detect.py)
def compute_loss(p, targets, model):
    from dyk import detect
    return detect.compute_loss(p, targets, model)


def build_targets(p, targets, model):
    from dyk import detect
    return detect.build_targets(p, targets, model)


def non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6, multi_label=True, classes=None, agnostic=False,
                        max_num=100):
    from dyk import detect
    return detect.non_max_suppression(prediction, conf_thres, iou_thres, multi_label, classes, agnostic, max_num)
