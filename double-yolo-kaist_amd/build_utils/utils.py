"""Loss, target assignment, IoU, NMS and box utilities: the names and call signatures of the reference's
build_utils/utils.py (reference build_utils/utils.py:24-469) on the MI355X HIP path.

Hot-path entry points (`compute_loss`, `build_targets`, `non_max_suppression`, and `xywh2xyxy` / `xyxy2xywh` /
`scale_coords` / `clip_coords` on device tensors) run as HIP kernels through the C ABI (dyk/detect.py,
csrc/boxes.hip).  The same box helpers accept host tensors and numpy arrays, which is how the reference's dataset
and plotting code (out of scope here) calls them; that is host glue, not a fallback of the device path.
Importing this module needs neither cv2 nor torchvision.
"""
import ctypes
import glob
import math
import os
import random

import numpy as np
import torch
import torch.nn as nn


def init_seeds(seed=0):
    """seed the three generators the reference seeds (utils.py:24-27)"""
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)


def check_file(file):
    """`file` itself when it exists, otherwise the first match of a recursive search below the working
    directory; AssertionError('File Not Found: ...') when there is none (utils.py:30-37)"""
    if os.path.isfile(file):
        return file
    hits = glob.glob(os.path.join(".", "**", file), recursive=True)
    assert hits, "File Not Found: %s" % file
    return hits[0]


def get_yolo_layers(model):
    return [i for i, d in enumerate(model.module_defs) if d["type"] == "yolo"]


# ------------------------------------------------------------------------------------------------------------
# box helpers
def _on_device(t):
    return isinstance(t, torch.Tensor) and t.is_cuda


def _dyk_rows(t, what):
    """(data pointer, row count, row stride in floats) of a 2-D fp32 device tensor whose rows are contiguous"""
    from dyk import lib as L
    if t.dim() != 2 or t.shape[1] < 4 or t.dtype != torch.float32 or (t.shape[0] > 0 and t.stride(1) != 1):
        raise L.DykError("%s expects an fp32 [n, >=4] device tensor with unit column stride, got %s %s strides %s"
                         % (what, t.dtype, tuple(t.shape), t.stride()))
    return t.data_ptr(), int(t.shape[0]), int(t.stride(0)) if t.shape[0] > 1 else int(t.shape[1])


def _kernel_view(t):
    """the tensor the box kernels can work on: `t` itself when it is fp32 with unit column stride, else an fp32 copy
    with contiguous rows (the reference's tensor expressions accept any dtype / view: half or double boxes, column
    views -- the kernels compute in fp32, the result is cast back by the caller)"""
    if t.dtype == torch.float32 and (t.shape[0] == 0 or t.stride(1) == 1):
        return t
    return t.to(torch.float32).contiguous()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _convert(x, to_xyxy):
    if _on_device(x):
        from dyk.lib import check, load
        xk = _kernel_view(x)
        out = torch.zeros(x.shape, dtype=torch.float32, device=x.device)   # columns past the box stay 0 (reference: zeros_like)
        src, n, ld = _dyk_rows(xk, "xywh2xyxy" if to_xyxy else "xyxy2xywh")
        dst, _, ldo = _dyk_rows(out, "box conversion output")
        if n:
            check(load().dyk_box_convert(src, dst, n, ld, ldo, 1 if to_xyxy else 0, _stream()), "dyk_box_convert")
        return out if x.dtype == torch.float32 else out.to(x.dtype)
    out = torch.zeros_like(x) if isinstance(x, torch.Tensor) else np.zeros_like(x)
    first, second = x[:, 0:2], x[:, 2:4]
    if to_xyxy:                                    # centre / size -> corners
        half = second / 2
        out[:, 0:2] = first - half
        out[:, 2:4] = first + half
    else:                                          # corners -> centre / size
        out[:, 0:2] = (first + second) / 2
        out[:, 2:4] = second - first
    return out


def xyxy2xywh(x):
    """[n,4] (x1,y1,x2,y2) -> (xc,yc,w,h)   (utils.py:40-47)"""
    return _convert(x, False)


def xywh2xyxy(x):
    """[n,4] (xc,yc,w,h) -> (x1,y1,x2,y2)   (utils.py:50-57)"""
    return _convert(x, True)


def _clip_scale(boxes, img0_shape, pad=(0.0, 0.0), gain=1.0, scale=False):
    h0, w0 = float(img0_shape[0]), float(img0_shape[1])
    if _on_device(boxes):
        from dyk.lib import check, load
        bk = _kernel_view(boxes)
        ptr, n, ld = _dyk_rows(bk, "scale_coords")
        if n:
            check(load().dyk_scale_coords(ptr, n, ld, float(pad[0]), float(pad[1]), float(gain), w0, h0, 1 if scale else 0,
                                          _stream()), "dyk_scale_coords")
        if bk is not boxes:
            boxes.copy_(bk)                        # in place, like the reference (dtype / view of the caller kept)
        return
    for cols, off, hi in (((0, 2), pad[0], w0), ((1, 3), pad[1], h0)):
        for c in cols:
            col = boxes[:, c]
            if scale:
                col -= off
                col /= gain
            col.clamp_(0, hi)


def clip_coords(boxes, img_shape):
    """clamp (x1,y1,x2,y2) rows to an image of shape (height, width), in place (utils.py:84-92)"""
    _clip_scale(boxes, img_shape)


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """undo the letterbox: boxes found on the network input `img1_shape` (h, w) are moved back to the frame
    `img0_shape` (h, w) -- subtract the padding, divide by the resize gain, clamp; in place, returns `coords`.
    ratio_pad = ((ratio_h, ratio_w), (pad_x, pad_y)) as recorded by the loader; when absent the gain is
    max(img1)/max(img0) and the padding is what centres the resized frame (utils.py:60-81)."""
    if ratio_pad is not None:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    else:
        gain = max(img1_shape) / max(img0_shape)
        pad = tuple((img1_shape[ax] - img0_shape[ax] * gain) / 2 for ax in (1, 0))
    _clip_scale(coords, img0_shape, pad, gain, scale=True)
    return coords


def _corners(b, xyxy):
    """four coordinate rows (x1, y1, x2, y2) of boxes stored as rows of a [4,n] tensor"""
    if xyxy:
        return b[0], b[1], b[2], b[3]
    hw, hh = b[2] / 2, b[3] / 2
    return b[0] - hw, b[1] - hh, b[0] + hw, b[1] + hh


def bbox_iou(box1, box2, x1y1x2y2=True, GIoU=False, DIoU=False, CIoU=False):
    """IoU (or G/D/C-IoU) of box1 [4,n] with box2 [n,4], element by element (utils.py:95-138).  The training
    loss evaluates the same expressions inside dyk_yolo_loss; this tensor form serves callers outside it."""
    ax1, ay1, ax2, ay2 = _corners(box1, x1y1x2y2)
    bx1, by1, bx2, by2 = _corners(box2.t(), x1y1x2y2)
    overlap_w = (torch.min(ax2, bx2) - torch.max(ax1, bx1)).clamp(0)
    overlap_h = (torch.min(ay2, by2) - torch.max(ay1, by1)).clamp(0)
    inter = overlap_w * overlap_h
    aw, ah, bw, bh = ax2 - ax1, ay2 - ay1, bx2 - bx1, by2 - by1
    union = (aw * ah + 1e-16) + bw * bh - inter
    iou = inter / union
    if not (GIoU or DIoU or CIoU):
        return iou
    hull_w = torch.max(ax2, bx2) - torch.min(ax1, bx1)          # smallest enclosing box
    hull_h = torch.max(ay2, by2) - torch.min(ay1, by1)
    if GIoU:
        hull = hull_w * hull_h + 1e-16
        return iou - (hull - union) / hull
    diag2 = hull_w ** 2 + hull_h ** 2 + 1e-16
    dist2 = ((bx1 + bx2) - (ax1 + ax2)) ** 2 / 4 + ((by1 + by2) - (ay1 + ay2)) ** 2 / 4
    if DIoU:
        return iou - dist2 / diag2
    aspect = (4 / math.pi ** 2) * torch.pow(torch.atan(bw / bh) - torch.atan(aw / ah), 2)
    with torch.no_grad():
        trade = aspect / (1 - iou + aspect)
    return iou - (dist2 / diag2 + aspect * trade)


def box_iou(box1, box2):
    """all-pairs IoU of corner boxes [N,4] x [M,4] -> [N,M] (utils.py:141-163)"""
    def area(b):
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lo = torch.max(box1[:, None, :2], box2[:, :2])
    hi = torch.min(box1[:, None, 2:], box2[:, 2:])
    inter = (hi - lo).clamp(0).prod(2)
    return inter / (area(box1)[:, None] + area(box2) - inter)


def wh_iou(wh1, wh2):
    """IoU of sizes anchored at a common corner, wh1 [n,2] x wh2 [m,2] -> [n,m] (utils.py:166-171)"""
    a, b = wh1[:, None], wh2[None]
    inter = torch.min(a, b).prod(2)
    return inter / (a.prod(2) + b.prod(2) - inter)


class FocalLoss(nn.Module):
    """modulates an element-wise BCE-with-logits criterion by alpha_t * (1 - p_t)^gamma (utils.py:174-201).
    fl_gamma is 0 in both reference hyp files, so the HIP loss never routes through it."""

    def __init__(self, loss_fcn, gamma=1.5, alpha=0.25):
        super().__init__()
        self.loss_fcn, self.gamma, self.alpha = loss_fcn, gamma, alpha
        self.reduction = loss_fcn.reduction
        self.loss_fcn.reduction = "none"           # the wrapper reduces after weighting

    def forward(self, pred, true):
        prob = torch.sigmoid(pred)
        p_t = true * prob + (1 - true) * (1 - prob)
        alpha_t = true * self.alpha + (1 - true) * (1 - self.alpha)
        loss = self.loss_fcn(pred, true) * (alpha_t * (1.0 - p_t) ** self.gamma)
        return {"mean": loss.mean, "sum": loss.sum}.get(self.reduction, lambda: loss)()


def smooth_BCE(eps=0.1):
    """label-smoothing targets (positive, negative) (utils.py:204-206)"""
    return 1.0 - 0.5 * eps, 0.5 * eps


# ------------------------------------------------------------------------------------------------------------
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
