"""Operator-level wrappers over the C ABI (one Python function per libdyk_hip.so entry point).

Tensors are torch CUDA tensors used purely as device-memory handles: channels-last
``[B, H, W, C]`` (bf16 or f32), possibly a channel slice of a wider buffer.  Nothing here
computes with torch; a missing library or a CPU tensor raises.
"""
import ctypes

import torch

from . import lib as _l
from .lib import (ACT_CODES, DYK_BF16, DYK_F32, EPI_ACCUM, EPI_AFFINE, EPI_OUT_F32, EPI_RESIDUAL, EPI_STATS,
                  DykConvDesc, DykWgradDesc, check, load)


def dtype_code(dt):
    if dt == torch.bfloat16:
        return DYK_BF16
    if dt == torch.float32:
        return DYK_F32
    raise TypeError("dyk supports bf16 and f32 activations, got %s" % dt)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _l.DykError("dyk operators run on the GPU only (got a %s tensor); there is no CPU fallback"
                              % t.device.type)


def nhwc_ld(t):
    """pixel stride (in elements) of a channels-last [B,H,W,C] tensor / channel-slice view."""
    B, H, W, C = t.shape
    ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else (t.stride(0) if B > 1 else C))
    assert t.stride(3) == 1 or C == 1, "channel dim must be innermost"
    if W > 1 and H > 1:
        assert t.stride(1) == W * ld, "tensor is not pixel-contiguous"
    if B > 1 and (H > 1 or W > 1):
        assert t.stride(0) == H * W * ld, "tensor is not pixel-contiguous"
    return ld


# --------------------------------------------------------------------------- layout / pack
def to_nhwc(x_nchw, dtype, cpad=None, mul=1.0, out=None):
    """float32 NCHW -> channels-last `dtype` [B,H,W,cpad] (zero padded channels)."""
    _require_cuda(x_nchw)
    assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    B, C, H, W = x_nchw.shape
    cpad = cpad or C
    if out is None:
        out = torch.empty((B, H, W, cpad), dtype=dtype, device=x_nchw.device)
    check(load().dyk_nchw_to_nhwc(_ptr(x_nchw), _ptr(out), B, C, H, W, cpad, nhwc_ld(out), float(mul),
                                  dtype_code(dtype), _stream()), "dyk_nchw_to_nhwc")
    return out


def to_nchw(x_nhwc, C=None):
    _require_cuda(x_nhwc)
    B, H, W, Cp = x_nhwc.shape
    C = C or Cp
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x_nhwc.device)
    check(load().dyk_nhwc_to_nchw(_ptr(x_nhwc), _ptr(out), B, C, H, W, nhwc_ld(x_nhwc),
                                  dtype_code(x_nhwc.dtype), _stream()), "dyk_nhwc_to_nchw")
    return out


def pack_weight(w_oihw, dtype, transposed=False, cout_pad=None, cin_pad=None, out=None):
    """torch OIHW float32 -> [taps][rows][cols] `dtype` (rows=Cout, cols=Cin; swapped if transposed)."""
    _require_cuda(w_oihw)
    assert w_oihw.dtype == torch.float32 and w_oihw.is_contiguous()
    Cout, Cin, kh, kw = w_oihw.shape
    cout_pad = cout_pad or Cout
    cin_pad = cin_pad or Cin
    R, C = (cin_pad, cout_pad) if transposed else (cout_pad, cin_pad)
    if out is None:
        out = torch.empty((kh * kw, R, C), dtype=dtype, device=w_oihw.device)
    check(load().dyk_pack_conv_weight(_ptr(w_oihw), _ptr(out), Cout, Cin, kh, kw, cout_pad, cin_pad,
                                      1 if transposed else 0, dtype_code(dtype), _stream()), "dyk_pack_conv_weight")
    return out


# --------------------------------------------------------------------------- conv tap tables
def conv_out_size(n, k, stride, pad):
    return (n + 2 * pad - k) // stride + 1


def fwd_taps(k, pad):
    """(tdy, tdx, twt) of a k x k forward convolution with zero padding `pad`."""
    taps = []
    for kh in range(k):
        for kw in range(k):
            taps.append((kh - pad, kw - pad, kh * k + kw))
    return taps


def dgrad_classes(k, pad, stride, Hi, Wi):
    """Data gradient of a stride-s conv as s*s dense launches, one per parity class of the
    input pixel (yi, xi) = (s*yh + py, s*xh + px).  Yields (py, px, Hg, Wg, taps) where taps are
    offsets into the *output-gradient* grid: source (yh + tdy, xh + tdx), weight tap twt."""
    out = []
    for py in range(stride):
        for px in range(stride):
            Hg = (Hi - py + stride - 1) // stride
            Wg = (Wi - px + stride - 1) // stride
            if Hg <= 0 or Wg <= 0:
                continue
            taps = []
            for kh in range(k):
                if (py + pad - kh) % stride:
                    continue
                for kw in range(k):
                    if (px + pad - kw) % stride:
                        continue
                    taps.append(((py + pad - kh) // stride, (px + pad - kw) // stride, kh * k + kw))
            out.append((py, px, Hg, Wg, taps))
    return out


def make_conv_desc(x, w, y, *, Hi, Wi, Cin, Cout, Hg, Wg, Ho, Wo, taps, isy=1, isx=1, osy=1, osx=1, ooy=0, oox=0,
                   act="linear", scale=None, shift=None, res=None, stats=None, accumulate=False, out_f32=False,
                   ldx=None, ldy=None, ldr=None, B=None, dtype=None):
    d = DykConvDesc()
    d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    d.scale = scale.data_ptr() if scale is not None else None
    d.shift = shift.data_ptr() if shift is not None else None
    d.res = res.data_ptr() if res is not None else None
    d.stats = stats.data_ptr() if stats is not None else None
    d.dtype = dtype_code(x.dtype) if dtype is None else dtype
    d.ldx = ldx if ldx is not None else nhwc_ld(x)
    d.ldy = ldy if ldy is not None else nhwc_ld(y)
    d.ldr = ldr if ldr is not None else (nhwc_ld(res) if res is not None else 0)
    d.B = B if B is not None else x.shape[0]
    d.Hi, d.Wi, d.Cin, d.Cout = Hi, Wi, Cin, Cout
    d.Hg, d.Wg, d.Ho, d.Wo = Hg, Wg, Ho, Wo
    d.isy, d.isx, d.osy, d.osx, d.ooy, d.oox = isy, isx, osy, osx, ooy, oox
    d.ntaps = len(taps)
    for i, (dy, dx, wt) in enumerate(taps):
        d.tdy[i], d.tdx[i], d.twt[i] = dy, dx, wt
    d.act = ACT_CODES[act] if isinstance(act, str) else int(act)
    flags = 0
    if scale is not None or shift is not None:
        flags |= EPI_AFFINE
    if res is not None:
        flags |= EPI_RESIDUAL
    if stats is not None:
        flags |= EPI_STATS
    if accumulate:
        flags |= EPI_ACCUM
    if out_f32:
        flags |= EPI_OUT_F32
    d.flags = flags
    return d


def attach_splitk(d, S, device):
    """split-K across workgroups for one launch of `d` (DykConvDesc.splitk, include/dyk_hip.h): slab scratch + zeroed tile
    counters sized by dyk_conv_splitk_ws_bytes for d's tune word.  Returns the two tensors (keep them alive over the launch)."""
    d.splitk = int(S)
    nt = ctypes.c_int32(0)
    need = int(load().dyk_conv_splitk_ws_bytes(ctypes.byref(d), ctypes.byref(nt)))
    if need <= 0:
        d.splitk = 0
        return None
    ws = torch.empty(need, dtype=torch.uint8, device=device)
    cnt = torch.zeros(max(nt.value, 1), dtype=torch.int32, device=device)
    d.sk_ws, d.sk_ws_bytes, d.sk_cnt, d.sk_cnt_n = ws.data_ptr(), need, cnt.data_ptr(), nt.value
    return ws, cnt


def conv2d_fwd(x, wp, k, stride, pad, Cout, *, act="linear", scale=None, shift=None, res=None, stats=None,
               out=None, out_f32=False, tune=0, stats_slots=0, splitk=0, keep=None):
    """y = epilogue(conv(x, w)); x [B,Hi,Wi,Cin] channels-last, wp = pack_weight(w).  splitk > 1: S workgroups per output
    tile (`keep`: a list that receives the scratch tensors, for callers that look at the tile counters afterwards)."""
    _require_cuda(x, wp)
    B, Hi, Wi, Cin = x.shape
    Ho, Wo = conv_out_size(Hi, k, stride, pad), conv_out_size(Wi, k, stride, pad)
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    d = make_conv_desc(x, wp, out, Hi=Hi, Wi=Wi, Cin=Cin, Cout=Cout, Hg=Ho, Wg=Wo, Ho=Ho, Wo=Wo,
                       taps=fwd_taps(k, pad), isy=stride, isx=stride, act=act, scale=scale, shift=shift,
                       res=res, stats=stats, out_f32=out_f32)
    d.tune, d.stats_slots = tune, stats_slots
    scratch = attach_splitk(d, splitk, x.device) if splitk > 1 else None
    if keep is not None:
        keep.append(scratch)
    check(load().dyk_conv_igemm(ctypes.byref(d), _stream()), "dyk_conv_igemm")
    return out


def conv2d_dgrad(dy, wpt, k, stride, pad, Hi, Wi, Cin, *, out=None, accumulate=False, merge=False):
    """dx = conv_transpose(dy, w); dy [B,Ho,Wo,Cout], wpt = pack_weight(w, transposed=True).
    merge: the parity classes of a strided conv in ONE launch (DykConvDesc.ncls) when they share the launch grid."""
    _require_cuda(dy, wpt)
    B, Ho, Wo, Cout = dy.shape
    if out is None:
        out = torch.empty((B, Hi, Wi, Cin), dtype=dy.dtype, device=dy.device)
    classes = dgrad_classes(k, pad, stride, Hi, Wi)
    if merge and len(classes) > 1:
        if not (all(c[4] for c in classes) and len({(c[2], c[3]) for c in classes}) == 1 and len(classes) <= 4):
            raise ValueError("parity classes of this data gradient cannot share a launch")
        d = make_conv_desc(dy, wpt, out, Hi=Ho, Wi=Wo, Cin=Cout, Cout=Cin, Hg=classes[0][2], Wg=classes[0][3], Ho=Hi, Wo=Wi,
                           taps=[t for c in classes for t in c[4]], osy=stride, osx=stride, accumulate=accumulate)
        d.ncls, q0 = len(classes), 0
        for c, (py, px, _, _, taps) in enumerate(classes):
            d.cls_first[c], d.cls_ntaps[c], d.cls_ooy[c], d.cls_oox[c] = q0, len(taps), py, px
            q0 += len(taps)
        check(load().dyk_conv_igemm(ctypes.byref(d), _stream()), "dyk_conv_igemm(dgrad, merged classes)")
        return out
    for (py, px, Hg, Wg, taps) in classes:
        d = make_conv_desc(dy, wpt, out, Hi=Ho, Wi=Wo, Cin=Cout, Cout=Cin, Hg=Hg, Wg=Wg, Ho=Hi, Wo=Wi,
                           taps=taps, osy=stride, osx=stride, ooy=py, oox=px, accumulate=accumulate)
        check(load().dyk_conv_igemm(ctypes.byref(d), _stream()), "dyk_conv_igemm(dgrad)")
    return out


def conv2d_wgrad(x, dy, k, stride, pad, *, dw=None, splits=0, Cin=None, Cout=None, tune=0):
    """dw[t][co][ci] += sum_n dy[n][co] x[src(n,t)][ci]  (fp32, packed tap-major order)."""
    _require_cuda(x, dy)
    B, Hi, Wi, Cx = x.shape
    _, Ho, Wo, Cy = dy.shape
    Cin = Cin or Cx
    Cout = Cout or Cy
    if dw is None:
        dw = torch.zeros((k * k, Cout, Cin), dtype=torch.float32, device=x.device)
    d = DykWgradDesc()
    d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
    d.dtype = dtype_code(x.dtype)
    d.ldx, d.lddy = nhwc_ld(x), nhwc_ld(dy)
    d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, Hi, Wi, Cin, Ho, Wo, Cout
    d.isy = d.isx = stride
    taps = fwd_taps(k, pad)
    d.ntaps = len(taps)
    for i, (ty, tx, wt) in enumerate(taps):
        d.tdy[i], d.tdx[i], d.twt[i] = ty, tx, wt
    d.splits = splits
    d.lddw = 0
    d.tune = tune
    check(load().dyk_conv_wgrad(ctypes.byref(d), _stream()), "dyk_conv_wgrad")
    return dw


# --------------------------------------------------------------------------- depthwise convolution
def _dw_desc(x, y, w_taps, k, stride, pad, C):
    from .lib import DykDwDesc
    d = DykDwDesc()
    d.x, d.y = x.data_ptr(), y.data_ptr()
    d.w = w_taps.data_ptr() if w_taps is not None else None
    d.dtype = dtype_code(x.dtype)
    d.ldx, d.ldy = nhwc_ld(x), nhwc_ld(y)
    d.B, d.Hi, d.Wi = x.shape[0], x.shape[1], x.shape[2]
    d.Ho, d.Wo, d.C = y.shape[1], y.shape[2], C
    d.k, d.stride, d.pad = k, stride, pad
    return d


def dwconv_fwd(x, w_taps, k, stride, pad, *, stats=None, stats_slots=1, C=None):
    """depthwise conv; x channels-last [B,H,W,ld>=C], w_taps fp32 [k*k][C]"""
    _require_cuda(x, w_taps)
    B, Hi, Wi, Cx = x.shape
    C = C or Cx
    Ho, Wo = conv_out_size(Hi, k, stride, pad), conv_out_size(Wi, k, stride, pad)
    y = torch.zeros((B, Ho, Wo, Cx), dtype=x.dtype, device=x.device)
    d = _dw_desc(x, y, w_taps, k, stride, pad, C)
    if stats is not None:
        d.stats, d.stats_slots = stats.data_ptr(), stats_slots
    check(load().dyk_dwconv_fwd(ctypes.byref(d), _stream()), "dyk_dwconv_fwd")
    return y


def dwconv_dgrad(dy, w_taps, k, stride, pad, Hi, Wi, *, out=None, accumulate=False, C=None):
    _require_cuda(dy, w_taps)
    B, Ho, Wo, Cy = dy.shape
    C = C or Cy
    if out is None:
        out = torch.zeros((B, Hi, Wi, Cy), dtype=dy.dtype, device=dy.device)
    d = _dw_desc(out, dy, w_taps, k, stride, pad, C)
    d.flags = 1 if accumulate else 0
    check(load().dyk_dwconv_dgrad(ctypes.byref(d), _stream()), "dyk_dwconv_dgrad")
    return out


def dwconv_wgrad(x, dy, k, stride, pad, *, dw=None, C=None, planes=False):
    """depthwise weight gradient [k*k][C] fp32 (accumulated into `dw`).  planes=True: the way the plan runs it -- every workgroup
    row stores its own partial plane (dyk_dwconv_wgrad_rows of them, DykDwDesc.part); returns the planes [rows][k*k][C]"""
    _require_cuda(x, dy)
    C = C or x.shape[3]
    if dw is None:
        dw = torch.zeros((k * k, C), dtype=torch.float32, device=x.device)
    d = _dw_desc(x, dy, None, k, stride, pad, C)
    d.dw = dw.data_ptr()
    part = None
    if planes:
        rows = load().dyk_dwconv_wgrad_rows(ctypes.byref(d))
        if rows < 1:
            raise RuntimeError("dyk_dwconv_wgrad_rows: %d" % rows)
        part = torch.full((rows, k * k, C), float("nan"), dtype=torch.float32, device=x.device)
        d.part = part.data_ptr()
    check(load().dyk_dwconv_wgrad(ctypes.byref(d), _stream()), "dyk_dwconv_wgrad")
    return part if planes else dw


# --------------------------------------------------------------------------- elementwise family
def ew_desc(a=None, b=None, out=None, *, C=None, npix=None, act="linear", flags=0, alpha=1.0, beta=1.0,
            p0=None, p1=None, p2=None, p3=None, red=None, aux=None, B=0, H=0, W=0, k=0):
    """Build a DykEwDesc from channels-last tensors ([B,H,W,C] views)."""
    d = _l.DykEwDesc()
    ref = a if a is not None else out
    d.dtype = dtype_code(ref.dtype)
    d.C = C if C is not None else ref.shape[3]
    d.npix = npix if npix is not None else ref.shape[0] * ref.shape[1] * ref.shape[2]
    d.a, d.b, d.out = _ptr(a), _ptr(b), _ptr(out)
    d.lda = nhwc_ld(a) if a is not None else 0
    d.ldb = nhwc_ld(b) if b is not None else 0
    d.ldo = nhwc_ld(out) if out is not None else 0
    d.p0, d.p1, d.p2, d.p3, d.red, d.aux = _ptr(p0), _ptr(p1), _ptr(p2), _ptr(p3), _ptr(red), _ptr(aux)
    d.act = ACT_CODES[act] if isinstance(act, str) else int(act)
    d.flags, d.alpha, d.beta = flags, alpha, beta
    d.B, d.H, d.W, d.k = B, H, W, k
    return d


def call(name, desc, *extra):
    fn = getattr(load(), name)
    check(fn(ctypes.byref(desc), *[_ptr(e) if isinstance(e, torch.Tensor) else e for e in extra], _stream()), name)


def bn_finalize(stats, count, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5):
    C = gamma.numel()
    dev = stats.device
    scale, shift, mean, rstd = (torch.empty(C, device=dev) for _ in range(4))
    d = _l.DykBnFinalizeDesc()
    d.stats, d.gamma, d.beta = stats.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    d.running_mean = running_mean.data_ptr() if running_mean is not None else None
    d.running_var = running_var.data_ptr() if running_var is not None else None
    d.scale, d.shift, d.save_mean, d.save_rstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    d.C, d.count, d.momentum, d.eps = C, count, momentum, eps
    call("dyk_bn_finalize", d)
    return scale, shift, mean, rstd
