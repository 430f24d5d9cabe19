"""Stand-alone forward of the operator classes in build_utils/layers.py and models.YOLOLayer on torch
NCHW float32 CUDA tensors (the reference's `module(x)` / `module(x, outputs)` surface).  Each call converts to the
channels-last compute layout, runs the same HIP kernels the compiled plan uses, and converts back.  This is the
module-level compatibility surface (no autograd); `models.YOLO.forward` never goes through here.
"""
import ctypes

import torch

from . import lib as L
from . import ops
from .lib import check, load


def _dtype():
    return torch.bfloat16 if torch.is_autocast_enabled() else torch.float32


def _ru(n, a):
    return (n + a - 1) // a * a


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _nhwc(x, cpad=None):
    ops._require_cuda(x)
    return ops.to_nhwc(x.float().contiguous(), _dtype(), cpad=cpad)


def concat(tensors):
    """FeatureConcat.forward for several inputs (reference layers.py:44)"""
    dt = _dtype()
    B, _, H, W = tensors[0].shape
    cs = [t.shape[1] for t in tensors]
    buf = torch.zeros((B, H, W, sum(cs)), dtype=dt, device=tensors[0].device)
    c0 = 0
    for t, c in zip(tensors, cs):
        ops._require_cuda(t)
        ops.to_nhwc(t.float().contiguous(), dt, out=buf[..., c0:c0 + c])
        c0 += c
    return ops.to_nchw(buf)


def weighted_fusion(x, others, w, n):
    """WeightedFeatureFusion.forward (reference layers.py:63-85); with mismatched channel counts the sum covers the first
    min(nx, na) channels and the output keeps x's channel count (:78-83)"""
    xd = _nhwc(x)
    weff = None
    if w is not None:
        weff = torch.zeros(n, device=x.device)
        check(load().dyk_wfuse_weights(w.detach().float().contiguous().data_ptr(), weff.data_ptr(), n, _stream()))
    out = xd
    for i, a in enumerate(others):
        ad = _nhwc(a)
        nx, na = xd.shape[3], ad.shape[3]
        C = min(nx, na)
        vec = 16 // xd.element_size()
        if nx != na and (C % vec or abs(nx - na) % vec):
            raise NotImplementedError("WeightedFeatureFusion with channel counts %d / %d that are not whole 16-byte vectors" % (nx, na))
        nxt = torch.empty_like(xd)
        w0 = weff[0:1] if (weff is not None and i == 0) else None
        d = ops.ew_desc(a=out[..., :C], b=ad[..., :C], out=nxt[..., :C], p0=w0, p1=weff[i + 1:i + 2] if weff is not None else None)
        ops.call("dyk_axpby", d)
        if nx > na:                                            # the rest of x passes through (scaled by w[0] on the first term)
            ops.call("dyk_axpby", ops.ew_desc(a=out[..., C:], out=nxt[..., C:], p0=w0))
        out = nxt
    return ops.to_nchw(out)


def squeeze_excitation(x, w1, b1, w2, b2):
    """SqueezeExcitation.forward (reference layers.py:184-190)"""
    xd = _nhwc(x)
    B, H, W, C = xd.shape
    Cs = w1.shape[0]
    pooled = torch.zeros(B, C, device=x.device)
    scale = torch.zeros(B, C, device=x.device)
    ops.call("dyk_se_pool", ops.ew_desc(a=xd, B=B, H=H, W=W, alpha=1.0 / (H * W)), pooled)
    prm = [t.detach().float().contiguous() for t in (w1, b1, w2, b2)]
    fd = L.DykSeFcDesc()
    fd.pooled, fd.scale = pooled.data_ptr(), scale.data_ptr()
    fd.w1, fd.b1, fd.w2, fd.b2 = (t.data_ptr() for t in prm)
    fd.B, fd.C, fd.Cs = B, C, Cs
    fcws = torch.empty(B * (C + 2 * Cs), device=x.device)       # h | dt1 | t2 (include/dyk_hip.h)
    fd.ws = fcws.data_ptr()
    ops.call("dyk_se_fc_fwd", fd)
    z = torch.empty_like(xd)
    ops.call("dyk_se_scale", ops.ew_desc(a=xd, out=z, p0=scale, B=B, H=H, W=W))
    return ops.to_nchw(z)


def conv_bn_act(x, conv, bn, act, training):
    """nn.Sequential(Conv2d[, BatchNorm2d][, activation]) forward (reference models.py:28-64); dense or depthwise."""
    dt = _dtype()
    k, s, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    cin, cout = conv.in_channels, conv.out_channels
    depthwise = conv.groups > 1
    if depthwise and not (conv.groups == cin == cout):
        raise NotImplementedError("grouped convolution that is not depthwise")
    cpad = _ru(cin, 32)
    xd = _nhwc(x, cpad=cpad)
    dev = x.device
    if depthwise:
        if bn is None:
            raise NotImplementedError("depthwise convolution without BatchNorm2d")
        wt = conv.weight.detach().float().reshape(cout, k * k).t().contiguous()

        def run_conv(stats=None, **epi):
            return ops.dwconv_fwd(xd, wt, k, s, pad, stats=stats, stats_slots=1, C=cout)
    else:
        wp = ops.pack_weight(conv.weight.detach().float().contiguous(), dt, cin_pad=cpad)

        def run_conv(stats=None, **epi):
            return ops.conv2d_fwd(xd, wp, k, s, pad, cout, stats=stats, **epi)
    if bn is None:
        bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
        return ops.to_nchw(run_conv(act=act, shift=bias))
    gamma, beta = bn.weight.detach().float().contiguous(), bn.bias.detach().float().contiguous()
    if not training:
        scale, shift = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
        check(load().dyk_bn_fold(gamma.data_ptr(), beta.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                 float(bn.eps), scale.data_ptr(), shift.data_ptr(), cout, _stream()))
        if not depthwise:
            return ops.to_nchw(run_conv(act=act, scale=scale, shift=shift))
        y = run_conv()
    else:
        stats = torch.zeros(2 * cout, dtype=torch.float64, device=dev)
        y = run_conv(stats=stats)
        n = y.shape[0] * y.shape[1] * y.shape[2]
        scale, shift, _, _ = ops.bn_finalize(stats, n, gamma, beta, bn.running_mean, bn.running_var,
                                             momentum=bn.momentum if bn.momentum is not None else 0.1, eps=bn.eps)
        bn.num_batches_tracked += 1
    z = torch.empty_like(y)
    ops.call("dyk_bn_act_fwd", ops.ew_desc(a=y, out=z, act=act, p0=scale, p1=shift, C=cout))
    return ops.to_nchw(z, C=cout)


def maxpool(x, k, stride=1):
    xd = _nhwc(x)
    B, H, W, C = xd.shape
    pad = (k - 1) // 2
    z = torch.empty((B, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1, C), dtype=xd.dtype, device=xd.device)
    d = ops.ew_desc(a=xd, out=z, B=B, H=H, W=W, k=k)
    d.slots = stride
    ops.call("dyk_maxpool_fwd", d, None)
    return ops.to_nchw(z)


def upsample2x(x):
    xd = _nhwc(x)
    B, H, W, C = xd.shape
    z = torch.empty((B, 2 * H, 2 * W, C), dtype=xd.dtype, device=xd.device)
    ops.call("dyk_upsample2x_fwd", ops.ew_desc(a=xd, out=z, B=B, H=H, W=W))
    return ops.to_nchw(z)


def yolo_layer(p, layer):
    """YOLOLayer.forward (reference models.py:218-258): p [B, na*no, ny, nx] -> training: [B,na,ny,nx,no];
    eval: (decoded [B, na*ny*nx, no], raw)"""
    ops._require_cuda(p)
    B, _, ny, nx = p.shape
    na, no = layer.na, layer.no
    y = torch.zeros((B, ny, nx, 32 if na * no <= 32 else _ru(na * no, 4)), dtype=torch.float32, device=p.device)
    ld = y.shape[3]
    ops.to_nhwc(p.float().contiguous(), torch.float32, out=y[..., :na * no])
    out = torch.empty((B, na, ny, nx, no), dtype=torch.float32, device=p.device)
    check(load().dyk_head_permute_fwd(y.data_ptr(), out.data_ptr(), B, ny, nx, na, no, ld, _stream()))
    if layer.training:
        return out
    io = torch.empty((B, na * ny * nx, no), dtype=torch.float32, device=p.device)
    d = L.DykDecodeDesc()
    d.p, d.io = out.data_ptr(), io.data_ptr()
    d.B, d.na, d.ny, d.nx, d.no = B, na, ny, nx, no
    d.rows_total, d.row_offset, d.v4, d.stride = na * ny * nx, 0, 1 if layer.bf_type == "yolov4" else 0, float(layer.stride)
    for i, v in enumerate(layer.anchor_vec.detach().float().cpu().reshape(-1).tolist()):
        d.anchor_vec[i] = v
    ops.call("dyk_yolo_decode", d)
    return io, out


def prepare_images(imgs, size=None, div=255.0):
    """Harness input path (reference train_utils/kaist_train_eval_utils.py:54-55, 59-71): the loader's uint8 batch
    [B,3,H,W] -> float32 in 0..1, optionally resized like F.interpolate(size=size, mode='bilinear',
    align_corners=False), in one HIP pass.  float32 input is taken as already divided (div forced to 1)."""
    ops._require_cuda(imgs)
    if imgs.dim() != 4:
        raise ValueError("prepare_images expects [B,C,H,W], got %s" % (tuple(imgs.shape),))
    if imgs.dtype == torch.uint8:
        sdt = L.DYK_U8
    elif imgs.dtype == torch.float32:
        sdt, div = L.DYK_F32, 1.0
    else:
        raise TypeError("prepare_images: uint8 or float32 images, got %s" % imgs.dtype)
    imgs = imgs.contiguous()
    B, C, H, W = imgs.shape
    Ho, Wo = (H, W) if size is None else (int(size[0]), int(size[1]))
    out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=imgs.device)
    check(load().dyk_image_prep(imgs.data_ptr(), out.data_ptr(), B * C, H, W, Ho, Wo, sdt, float(div), _stream()),
          "dyk_image_prep")
    return out


def multi_scale_pair(v_imgs, l_imgs, img_size, gs=32):
    """The multi-scale step of the reference training loop (kaist_train_eval_utils.py:59-71): scale factor
    sf = img_size / max(H, W), new size ceil(x*sf/gs)*gs per axis, both streams resized alike."""
    import math
    assert v_imgs.shape[:2] == l_imgs.shape[:2]
    sf = img_size / max(v_imgs.shape[2:])
    ns = None
    if sf != 1:
        ns = [math.ceil(x * sf / gs) * gs for x in v_imgs.shape[2:]]
    return prepare_images(v_imgs, ns), prepare_images(l_imgs, ns)
