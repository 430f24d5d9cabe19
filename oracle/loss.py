"""Oracle (test infrastructure): target assignment and YOLO loss on the CPU, fp32 / int64.

Restates reference build_utils/utils.py:95-138 (bbox_iou), :166-171 (wh_iou), :209-293
(compute_loss) and :296-384 (build_targets).  Pinned by tests/golden/targets_*.npz and
loss_*.npz generated from the reference (tests/golden/make_golden.py).
"""
import math

import torch
import torch.nn.functional as F


def wh_iou(wh1, wh2):
    """utils.py:166-171: IoU of boxes sharing a corner; wh1 [n,2], wh2 [m,2] -> [n,m]."""
    a = wh1[:, None]
    b = wh2[None]
    inter = torch.min(a, b).prod(2)
    return inter / (a.prod(2) + b.prod(2) - inter)


def bbox_iou_xywh(box1, box2, kind="ciou"):
    """utils.py:95-138 with x1y1x2y2=False.  box1 [4,n] (cx,cy,w,h rows), box2 [n,4]."""
    box2 = box2.t()
    b1_x1, b1_x2 = box1[0] - box1[2] / 2, box1[0] + box1[2] / 2
    b1_y1, b1_y2 = box1[1] - box1[3] / 2, box1[1] + box1[3] / 2
    b2_x1, b2_x2 = box2[0] - box2[2] / 2, box2[0] + box2[2] / 2
    b2_y1, b2_y2 = box2[1] - box2[3] / 2, box2[1] + box2[3] / 2
    inter = (torch.min(b1_x2, b2_x2) - torch.max(b1_x1, b2_x1)).clamp(0) * \
            (torch.min(b1_y2, b2_y2) - torch.max(b1_y1, b2_y1)).clamp(0)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1
    union = (w1 * h1 + 1e-16) + w2 * h2 - inter                 # :116
    iou = inter / union
    if kind == "iou":
        return iou
    cw = torch.max(b1_x2, b2_x2) - torch.min(b1_x1, b2_x1)
    ch = torch.max(b1_y2, b2_y2) - torch.min(b1_y1, b2_y1)
    if kind == "giou":                                          # :122-124
        c_area = cw * ch + 1e-16
        return iou - (c_area - union) / c_area
    c2 = cw ** 2 + ch ** 2 + 1e-16                              # :127
    rho2 = ((b2_x1 + b2_x2) - (b1_x1 + b1_x2)) ** 2 / 4 + ((b2_y1 + b2_y2) - (b1_y1 + b1_y2)) ** 2 / 4
    if kind == "diou":
        return iou - rho2 / c2
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)   # :133
    with torch.no_grad():
        alpha = v / (1 - iou + v)                               # :135
    return iou - (rho2 / c2 + v * alpha)


def build_targets(pred_shapes, targets, anchor_vecs, iou_t):
    """utils.py:296-384.  pred_shapes: list of (B,na,ny,nx,no); targets [nt,6] =
    (image, class, xc, yc, w, h) normalised; anchor_vecs: list of [na,2] (anchors / stride).
    Returns per head: tcls, tbox, (b, a, gj, gi), anchors  -- indices are int64."""
    nt = targets.shape[0]
    tcls, tbox, indices, anch = [], [], [], []
    gain = torch.ones(6)
    for i, shape in enumerate(pred_shapes):
        anchors = anchor_vecs[i]
        gain[2:] = torch.tensor(shape)[[3, 2, 3, 2]].float()    # :328  (nx, ny, nx, ny)
        na = anchors.shape[0]
        at = torch.arange(na).view(na, 1).repeat(1, nt)         # :336
        a, t = [], targets * gain
        if nt:
            j = wh_iou(anchors, t[:, 4:6]) > iou_t              # :352  [na, nt]
            a, t = at[j], t.repeat(na, 1, 1)[j]                 # :361  anchor-major, then target order
        b, c = t[:, :2].long().t()                              # :367
        gxy = t[:, 2:4]
        gwh = t[:, 4:6]
        gij = gxy.long()                                        # :370 truncation, no clamp
        gi, gj = gij.t()
        indices.append((b, a, gj, gi))
        tbox.append(torch.cat((gxy - gij, gwh), 1))
        anch.append(anchors[a])
        tcls.append(c)
    return tcls, tbox, indices, anch


def compute_loss(p, targets, anchor_vecs, hyp, nc, gr, v4):
    """utils.py:209-293.  p: list of [B,na,ny,nx,5+nc] raw logits (fp32).  hyp needs box/obj/cls,
    cls_pw, obj_pw, iou_t, fl_gamma (> 0: both BCE terms become focal, :236-238) and selects CIoU by the presence of
    key 'ciou' (:264).  Returns dict of three [1] tensors."""
    lcls, lbox, lobj = torch.zeros(1), torch.zeros(1), torch.zeros(1)
    tcls, tbox, indices, anchors = build_targets([tuple(pi.shape) for pi in p], targets, anchor_vecs, hyp["iou_t"])
    cls_pw = torch.tensor([hyp["cls_pw"]])
    obj_pw = torch.tensor([hyp["obj_pw"]])
    gamma = float(hyp.get("fl_gamma", 0.0))

    def bce(pred, true, pw):
        """BCEWithLogitsLoss(pos_weight, 'mean') (:229-230); with fl_gamma > 0 wrapped as FocalLoss(.., gamma) does
        (:184-197, alpha at its default 0.25 of :176): element loss * alpha_factor * (1 - p_t) ** gamma, then the mean"""
        if gamma <= 0:
            return F.binary_cross_entropy_with_logits(pred, true, pos_weight=pw)
        el = F.binary_cross_entropy_with_logits(pred, true, pos_weight=pw, reduction="none")
        prob = torch.sigmoid(pred)                                            # :190
        p_t = true * prob + (1 - true) * (1 - prob)                           # :191
        el = el * (true * 0.25 + (1 - true) * 0.75) * (1.0 - p_t) ** gamma    # :192-194
        return el.mean()                                                      # :196-197
    for i, pi in enumerate(p):
        b, a, gj, gi = indices[i]
        tobj = torch.zeros_like(pi[..., 0])
        nb = b.shape[0]
        if nb:
            ps = pi[b, a, gj, gi]                               # :248
            if v4:                                              # :258-259
                pxy = ps[:, :2].sigmoid() * 2. - 0.5
                pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anchors[i]
            else:                                               # :261-262
                pxy = ps[:, :2].sigmoid()
                pwh = ps[:, 2:4].exp().clamp(max=1E3) * anchors[i]
            pbox = torch.cat((pxy, pwh), 1)
            iou = bbox_iou_xywh(pbox.t(), tbox[i], "ciou" if "ciou" in hyp else "giou")
            lbox = lbox + (1.0 - iou).mean()                    # :268
            tobj[b, a, gj, gi] = (1.0 - gr) + gr * iou.detach().clamp(0).type(tobj.dtype)   # :271
            if nc > 1:                                          # :274-277
                t = torch.full_like(ps[:, 5:], 0.0)
                t[range(nb), tcls[i]] = 1.0
                lcls = lcls + bce(ps[:, 5:], t, cls_pw)           # :277
        lobj = lobj + bce(pi[..., 4], tobj, obj_pw)             # :283
    return {"box_loss": lbox * hyp["box"], "obj_loss": lobj * hyp["obj"], "class_loss": lcls * hyp["cls"]}
