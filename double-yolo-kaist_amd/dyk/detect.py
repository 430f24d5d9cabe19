"""Host side of the loss / target-assignment / NMS kernels (dyk_yolo_loss, dyk_build_targets, dyk_nms)."""
import ctypes

import torch

from . import lib as L
from .lib import check, load


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _model_of(model):
    # build_targets tolerates a DataParallel / DDP wrapper (reference utils.py:315,321)
    return model.module if hasattr(model, "module") and hasattr(model.module, "module_list") else model


def _targets_desc(shapes, targets, model):
    """shapes: list of (B, na, ny, nx, no).  Returns (desc, keepalive dict)."""
    m = _model_of(model)
    dev = targets.device
    nt = int(targets.shape[0])
    nheads = len(shapes)
    na = shapes[0][1]
    cap = max(na * nt, 1)
    t = L.DykTargetsDesc()
    tg = targets.detach().to(torch.float32).contiguous()
    keep = {"targets": tg}
    t.targets = tg.data_ptr() if nt else None
    t.nt, t.nheads, t.na = nt, nheads, na
    for h, (j, shp) in enumerate(zip(m.yolo_layers, shapes)):
        t.ny[h], t.nx[h] = shp[2], shp[3]
        # anchor_vec is a device tensor after model.to(device): reading it back every step would be a host sync in the
        # middle of the train step (the CPU then enqueues the backward behind an idle GPU) -- read once per model
        cache = m.__dict__.setdefault("_dyk_anchor_cache", {})
        av = cache.get(j)
        if av is None:
            av = cache[j] = m.module_list[j].anchor_vec.detach().float().cpu().reshape(-1).tolist()
        for q, v in enumerate(av):
            t.anchor_vec[h][q] = v
    t.iou_t = float(m.hyp["iou_t"])
    # (torch.empty: no fill launches inside the train step -- build_targets_kernel writes counts[h] unconditionally and the
    # match lists are only ever read up to counts[h])
    keep["counts"] = torch.empty(nheads, dtype=torch.int32, device=dev)
    keep["indices"] = torch.empty((nheads, 4, cap), dtype=torch.int64, device=dev)
    keep["tbox"] = torch.empty((nheads, cap, 4), dtype=torch.float32, device=dev)
    keep["anch"] = torch.empty((nheads, cap, 2), dtype=torch.float32, device=dev)
    keep["tcls"] = torch.empty((nheads, cap), dtype=torch.int64, device=dev)
    t.counts, t.indices, t.tbox = keep["counts"].data_ptr(), keep["indices"].data_ptr(), keep["tbox"].data_ptr()
    t.anch, t.tcls = keep["anch"].data_ptr(), keep["tcls"].data_ptr()
    return t, keep


def _require_cuda(t, what):
    if not t.is_cuda:
        raise L.DykError("%s runs on the MI355X HIP path only (got a %s tensor); there is no CPU fallback" % (what, t.device.type))


def build_targets(p, targets, model):
    """reference build_utils/utils.py:296-384 -> (tcls, tbox, indices, anch), lists over heads."""
    _require_cuda(p[0], "build_targets")
    t, keep = _targets_desc([tuple(pi.shape) for pi in p], targets.to(p[0].device), model)
    check(load().dyk_build_targets(ctypes.byref(t), _stream()), "dyk_build_targets")
    counts = keep["counts"].cpu().tolist()
    tcls, tbox, indices, anch = [], [], [], []
    nc = _model_of(model).nc
    for h, n in enumerate(counts):
        ib = keep["indices"][h, :, :n]
        indices.append((ib[0], ib[1], ib[2], ib[3]))
        tbox.append(keep["tbox"][h, :n])
        anch.append(keep["anch"][h, :n])
        c = keep["tcls"][h, :n]
        tcls.append(c)
        if n:
            assert int(c.max()) < nc, ("Model accepts %g classes labeled from 0-%g, however you labelled a class %g. "
                                       % (nc, nc - 1, int(c.max())))
    return tcls, tbox, indices, anch


class _LossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, targets, *p):
        m = _model_of(model)
        h = m.hyp
        dev = p[0].device
        ps = [pi.detach().float().contiguous() for pi in p]
        t, keep = _targets_desc([tuple(pi.shape) for pi in ps], targets.to(dev), model)
        d = L.DykLossDesc()
        # ONE buffer for everything dyk_yolo_loss wants zeroed: acc (12 doubles) | flag | 4 spare bytes | dp of every head |
        # tobj of every head, back to back -- one fill launch instead of 2 * nheads + 2 (and the backward scales all heads'
        # gradients in one launch).  acc leads: the allocation is 8-byte aligned whatever the grids are (441 * B * ny * nx
        # floats of dp + tobj are an odd count when B, ny and nx are all odd -- batch 1 at 416 x 416 -- ADVICE r4)
        n_dp = [pi.numel() for pi in ps]
        n_to = [pi.numel() // pi.shape[4] for pi in ps]
        flat = torch.empty(24 + 2 + sum(n_dp) + sum(n_to), dtype=torch.float32, device=dev)
        acc = flat[0:24].view(torch.float64)
        flag = flat[24:25].view(torch.int32)
        dps, tobjs, off = [], [], 26
        for pi, n in zip(ps, n_dp):
            dps.append(flat[off:off + n].view(pi.shape))
            off += n
        for pi, n in zip(ps, n_to):
            tobjs.append(flat[off:off + n].view(pi.shape[:4]))
            off += n
        for i, pi in enumerate(ps):
            d.p[i], d.dp[i], d.tobj[i] = pi.data_ptr(), dps[i].data_ptr(), tobjs[i].data_ptr()
        d.nheads, d.B, d.no = len(ps), ps[0].shape[0], ps[0].shape[4]
        d.nc = d.no - 5
        if m.nc != d.nc:
            raise ValueError("model.nc = %d but the heads predict %d classes" % (m.nc, d.nc))
        d.v4 = 1 if "yolov4" in m.cfg else 0
        d.ciou = 1 if "ciou" in h else 0
        d.hyp_box, d.hyp_obj, d.hyp_cls = float(h["box"]), float(h["obj"]), float(h["cls"])
        d.cls_pw, d.obj_pw, d.gr = float(h["cls_pw"]), float(h["obj_pw"]), float(m.gr)
        # focal loss around both BCE terms when hyp['fl_gamma'] > 0 (utils.py:236-238: FocalLoss(BCE, g), alpha at its default)
        d.fl_gamma, d.fl_alpha = max(0.0, float(h.get("fl_gamma", 0.0))), 0.25
        out = torch.empty(3, dtype=torch.float32, device=dev)
        d.acc, d.out, d.flag = acc.data_ptr(), out.data_ptr(), flag.data_ptr()
        lib = load()
        # (the one fill of `flat` covers acc and the flag word)
        rc = lib.dyk_yolo_loss(ctypes.byref(d), ctypes.byref(t), _stream())
        check(rc, "dyk_yolo_loss")
        ctx.dps = dps
        ctx.dp_flat = flat[26:26 + sum(n_dp)]
        ctx.no = d.no
        ctx.set_materialize_grads(False)        # an unused loss term arrives as None (no zeros tensor: a fill launch)
        # bit 0: a target fell outside the grid (the reference raises IndexError there).  Copied to pinned host memory
        # behind the loss kernels; examined without a host sync by raise_if_target_outside_grid (optimizer.step())
        ring = m.__dict__.get("_dyk_flag_ring")
        if ring is None:
            ring = m.__dict__["_dyk_flag_ring"] = [torch.zeros(32, dtype=torch.int32).pin_memory(), 0]
        pend = m.__dict__.setdefault("_dyk_loss_flags", [])
        if len(pend) >= 32:                      # nobody examined them (no fused optimizer in use): recycle the oldest slot
            pend.pop(0)[1].synchronize()
        host = ring[0][ring[1] % 32:ring[1] % 32 + 1]
        ring[1] += 1
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pend.append((host, ev, flag))
        m._dyk_loss_flag = flag
        m._dyk_loss_keep = (keep, tobjs, acc, ps)
        # three views of `out` (no clone launches).  They are outputs of a custom Function: in-place arithmetic ON them is refused
        # by autograd; the reference's harness only sums / stacks them (kaist_train_eval_utils.py:70-85)
        return out[0:1], out[1:2], out[2:3]

    @staticmethod
    def backward(ctx, gbox, gobj, gcls):
        gs = []
        for x in (gbox, gobj, gcls):
            if x is not None and (x.dtype != torch.float32 or not x.is_contiguous()):
                x = x.float().contiguous()
            gs.append(x)
        ptr = [x.data_ptr() if x is not None else None for x in gs]
        check(load().dyk_loss_scale_grads3(ctx.dp_flat.data_ptr(), ctx.dp_flat.numel(), ctx.no, ptr[0], ptr[1], ptr[2], _stream()),
              "dyk_loss_scale_grads3")
        return (None, None) + tuple(ctx.dps)


def raise_if_target_outside_grid(model, wait=True):
    """IndexError if a compute_loss call since the last check saw a target whose grid cell lies outside the
    prediction grid (x or y == 1.0 exactly) -- the reference fails there with 'index out of range'
    (build_utils/utils.py:248 after the unclamped :370).  wait=False only examines flags whose asynchronous copy
    has completed; the fused optimizers call it that way on every step."""
    m = _model_of(model)
    pending = m.__dict__.get("_dyk_loss_flags", [])
    keep, bad = [], False
    for host, ev, flag in pending:
        if wait:
            ev.synchronize()
        if ev.query():
            bad = bad or bool(int(host[0]) & 1)
        else:
            keep.append((host, ev, flag))
    m.__dict__["_dyk_loss_flags"] = keep
    if bad:
        raise IndexError("a target box centre lies outside the prediction grid (x or y == 1.0): the reference fails "
                         "with 'index out of range' in compute_loss (build_utils/utils.py:248)")


def compute_loss(p, targets, model):
    """reference build_utils/utils.py:209-293 -> {'box_loss','obj_loss','class_loss'}, each shape [1]."""
    _require_cuda(p[0], "compute_loss")
    lbox, lobj, lcls = _LossFunction.apply(model, targets, *p)
    return {"box_loss": lbox, "obj_loss": lobj, "class_loss": lcls}


def non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6, multi_label=True, classes=None, agnostic=False,
                        max_num=100, return_rows=False):
    """reference build_utils/utils.py:387-464 -> list (per image) of [n,6] tensors or None."""
    _require_cuda(prediction, "non_max_suppression")
    pred = prediction.detach().float().contiguous()
    B, N, no = pred.shape
    nc = no - 5
    multi = bool(multi_label) and nc > 1
    lib = load()
    d = L.DykNmsDesc()
    per = int(lib.dyk_nms_workspace_bytes(N, no, 1 if multi else 0))
    ws = torch.empty(B * per, dtype=torch.uint8, device=pred.device)
    out = torch.zeros((B, max_num, 6), dtype=torch.float32, device=pred.device)
    rows = torch.zeros((B, max_num), dtype=torch.int32, device=pred.device)
    counts = torch.zeros(B, dtype=torch.int32, device=pred.device)
    d.pred, d.out, d.out_rows, d.counts, d.ws, d.ws_per_image = pred.data_ptr(), out.data_ptr(), rows.data_ptr(), counts.data_ptr(), ws.data_ptr(), per
    d.B, d.N, d.no = B, N, no
    d.conf_thres, d.iou_thres = float(conf_thres), float(iou_thres)
    d.multi_label, d.agnostic, d.max_num = 1 if multi else 0, 1 if agnostic else 0, int(max_num)
    cl = list(classes) if classes else []
    d.n_classes = len(cl)
    for i, c in enumerate(cl[:16]):
        d.classes[i] = int(c)
    check(lib.dyk_nms(ctypes.byref(d), _stream()), "dyk_nms")
    cnt = counts.cpu().tolist()                    # the one host sync of the eval step
    res = [out[b, :n] if n else None for b, n in enumerate(cnt)]
    if return_rows:
        return res, [rows[b, :n].long() if n else None for b, n in enumerate(cnt)]
    return res
