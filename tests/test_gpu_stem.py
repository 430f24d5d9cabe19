"""Direct stem convolution (csrc/stem.hip: dyk_stem_conv_fwd / dyk_stem_conv_wgrad) through the C ABI against torch's
CPU fp32 convolution: float and uint8 images (the uint8 path must equal `img.float() / 255.0` bit for bit in its
input conversion), 16 / 32 filters, stride 1 / 2, ragged sizes, BatchNorm statistics, eval affine + activation,
weight gradient incl. accumulation and run-to-run bit identity."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _desc(img, w, stride, dtype):
    from dyk import lib as L
    B, _, H, W = img.shape
    cout = w.shape[0]
    d = L.DykStemDesc()
    d.img, d.in_u8 = img.data_ptr(), 1 if img.dtype == torch.uint8 else 0
    d.dtype = L.DYK_BF16 if dtype == torch.bfloat16 else L.DYK_F32
    d.B, d.H, d.W, d.Cout, d.k, d.stride, d.pad = B, H, W, cout, 3, stride, 1
    d.Ho, d.Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    return d


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


CASES = [(32, 1, torch.bfloat16, False, (2, 40, 56), False), (32, 1, torch.float32, True, (3, 33, 47), False),
         (16, 2, torch.bfloat16, True, (2, 64, 96), False), (32, 2, torch.float32, False, (1, 31, 45), False),
         (32, 1, torch.bfloat16, True, (16, 128, 160), False),
         # tight rows (ld == cout) + uint8 + bf16: the word-load / prefetching weight-gradient kernel
         (16, 2, torch.bfloat16, True, (2, 64, 96), True), (16, 2, torch.bfloat16, True, (3, 70, 600), True),
         (32, 1, torch.bfloat16, True, (2, 37, 300), True), (16, 1, torch.bfloat16, True, (1, 9, 12), True)]


@pytest.mark.parametrize("cout,stride,dtype,u8,shape,tight", CASES)
def test_stem_forward_statistics_and_weight_gradient(cout, stride, dtype, u8, shape, tight):
    from dyk import lib as L
    lib = L.load()
    B, H, W = shape
    g = torch.Generator().manual_seed(cout + stride + H)
    img8 = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g)
    imgf = img8.float() / 255.0
    w = torch.randn(cout, 3, 3, 3, generator=g) * 0.3                     # OIHW, as nn.Conv2d stores it
    ref = F.conv2d(imgf, w, stride=stride, padding=1)                     # [B,cout,Ho,Wo]
    img = (img8 if u8 else imgf).cuda().contiguous()
    w_store = w.permute(0, 2, 3, 1).contiguous().cuda()                   # parameter-store layout [co][ky][kx][c]
    wt = w_store.view(cout, 27).t().contiguous()
    d = _desc(img, w, stride, dtype)
    Ho, Wo = d.Ho, d.Wo
    ld = cout if tight else 32
    y = torch.zeros((B, Ho, Wo, ld), dtype=dtype, device="cuda")
    slots = 4
    stats = torch.zeros((slots, 2 * cout), dtype=torch.float64, device="cuda")
    d.wt, d.y, d.ldy, d.stats, d.stats_slots = wt.data_ptr(), y.data_ptr(), ld, stats.data_ptr(), slots
    L.check(lib.dyk_stem_conv_fwd(ctypes.byref(d), _stream()), "dyk_stem_conv_fwd")
    got = y[..., :cout].float().cpu().permute(0, 3, 1, 2)
    tol = 2.0 ** -8 if dtype == torch.bfloat16 else 2e-6
    assert float((got - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6
    if cout < ld:
        assert float(y[..., cout:].abs().max()) == 0.0                   # padding columns are never written
    st = stats.sum(0).cpu()
    rd = ref.double()
    assert torch.allclose(st[:cout], rd.sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(st[cout:], (rd * rd).sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
    # eval form: folded BatchNorm affine + activation, no statistics
    scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    y2 = torch.zeros_like(y)
    d.y, d.stats, d.scale, d.shift, d.act = y2.data_ptr(), None, scale.data_ptr(), shift.data_ptr(), L.ACT_CODES["leaky"]
    L.check(lib.dyk_stem_conv_fwd(ctypes.byref(d), _stream()), "dyk_stem_conv_fwd")
    ref2 = F.leaky_relu(ref * scale.cpu().view(1, -1, 1, 1) + shift.cpu().view(1, -1, 1, 1), 0.1)
    got2 = y2[..., :cout].float().cpu().permute(0, 3, 1, 2)
    assert float((got2 - ref2).abs().max()) <= tol * float(ref2.abs().max()) + 1e-6
    # weight gradient: dW = sum_p dy (x) patch, accumulated into dw
    dy = (torch.randn(B, Ho, Wo, ld, generator=g) * 0.1).to(dtype)
    dy[..., cout:] = 0
    wr = w.clone().requires_grad_(True)
    (F.conv2d(imgf, wr, stride=stride, padding=1) * dy[..., :cout].float().permute(0, 3, 1, 2)).sum().backward()
    want = wr.grad.permute(0, 2, 3, 1).reshape(cout, 27)
    planes = lib.dyk_stem_wgrad_planes(ctypes.byref(d))
    assert planes >= 1
    part = torch.empty(planes * cout * 27, dtype=torch.float32, device="cuda")
    dw = torch.full((cout, 27), 0.5, dtype=torch.float32, device="cuda")
    dyc = dy.cuda()
    d.dy, d.lddy, d.dw, d.part = dyc.data_ptr(), ld, dw.data_ptr(), part.data_ptr()
    L.check(lib.dyk_stem_conv_wgrad(ctypes.byref(d), _stream()), "dyk_stem_conv_wgrad")
    first = dw.clone()
    assert float((first.cpu() - 0.5 - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-5
    dw.fill_(0.5)
    L.check(lib.dyk_stem_conv_wgrad(ctypes.byref(d), _stream()), "dyk_stem_conv_wgrad")
    assert torch.equal(dw, first), "weight gradient must be bit-reproducible"


@pytest.mark.parametrize("cout,stride,shape", [(32, 1, (2, 37, 300)), (16, 2, (3, 70, 600)), (32, 1, (4, 64, 128))])
def test_stem_weight_gradient_with_the_batchnorm_backward_apply_inside(cout, stride, shape):
    """DykStemDesc.bn_fused (uint8 images, bf16): the weight gradient reads da and the raw conv output, folds the reduction
    replicas and applies  dz = scale * (da - S1/N - xhat * S2/N)  on the fly; workgroup 0 adds S2 / S1 to dgamma / dbeta.
    Against the two separate passes (dyk_bn_act_bwd_apply, then dyk_stem_conv_wgrad on its output): the same dz, rounded to
    bf16 the same way -- weight gradient, dgamma and dbeta agree to fp32 summation level; DYK_EW_SKIP makes the separate pass
    a no-op; a float image batch is not fusable."""
    from dyk import lib as L
    lib = L.load()
    B, H, W = shape
    g = torch.Generator().manual_seed(7 * cout + stride)
    img = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g).cuda().contiguous()
    d = _desc(img, torch.zeros(cout, 3, 3, 3), stride, torch.bfloat16)
    Ho, Wo = d.Ho, d.Wo
    n = B * Ho * Wo
    da = (torch.randn(B, Ho, Wo, cout, generator=g) * 0.2).bfloat16().cuda()
    yraw = (torch.randn(B, Ho, Wo, cout, generator=g) * 1.3 + 0.2).bfloat16().cuda()
    mean = yraw.float().mean((0, 1, 2))
    rstd = (yraw.float().var((0, 1, 2), unbiased=False) + 1e-5).rsqrt()
    scale = (torch.rand(cout, generator=g) + 0.5).cuda() * rstd
    vecs = torch.cat([scale, torch.zeros(cout, device="cuda"), mean, rstd]).contiguous()
    xhat = (yraw.float() - mean) * rstd
    slots = 16
    red = torch.zeros(slots, 2, cout, dtype=torch.float64, device="cuda")
    # the sums as a fused data-gradient epilogue leaves them: spread over the replicas
    s1, s2 = da.double().sum((0, 1, 2)), (da.double() * xhat.double()).sum((0, 1, 2))
    wts = torch.rand(slots, generator=g).double().cuda()
    wts /= wts.sum()
    red[:, 0] = wts.view(-1, 1) * s1
    red[:, 1] = wts.view(-1, 1) * s2
    planes = lib.dyk_stem_wgrad_planes(ctypes.byref(d))
    part = torch.empty(planes * cout * 27, dtype=torch.float32, device="cuda")
    # ---- two passes
    dz = torch.empty_like(da)
    dg0, db0 = torch.full((cout,), 0.25, device="cuda"), torch.full((cout,), -0.5, device="cuda")
    e = L.DykEwDesc()
    e.a, e.b, e.out = da.data_ptr(), yraw.data_ptr(), dz.data_ptr()
    e.p0, e.p1, e.p2, e.p3 = vecs.data_ptr(), vecs.data_ptr() + 4 * cout, vecs.data_ptr() + 8 * cout, vecs.data_ptr() + 12 * cout
    e.red, e.slots, e.aux, e.aux2 = red.data_ptr(), slots, dg0.data_ptr(), db0.data_ptr()
    e.dtype, e.npix, e.C, e.lda, e.ldb, e.ldo, e.act = L.DYK_BF16, n, cout, cout, cout, cout, 0
    L.check(lib.dyk_bn_act_bwd_apply(ctypes.byref(e), _stream()), "dyk_bn_act_bwd_apply")
    dw0 = torch.zeros((cout, 27), dtype=torch.float32, device="cuda")
    d.dy, d.lddy, d.dw, d.part = dz.data_ptr(), cout, dw0.data_ptr(), part.data_ptr()
    L.check(lib.dyk_stem_conv_wgrad(ctypes.byref(d), _stream()), "dyk_stem_conv_wgrad")
    # ---- one pass
    dg1, db1 = torch.full((cout,), 0.25, device="cuda"), torch.full((cout,), -0.5, device="cuda")
    dw1 = torch.zeros((cout, 27), dtype=torch.float32, device="cuda")
    d.dy, d.dw = None, dw1.data_ptr()
    d.bn_da, d.bn_yraw, d.bn_vecs, d.bn_red, d.bn_slots = da.data_ptr(), yraw.data_ptr(), vecs.data_ptr(), red.data_ptr(), slots
    d.bn_dgamma, d.bn_dbeta = dg1.data_ptr(), db1.data_ptr()
    assert lib.dyk_stem_wgrad_bn_fusable(ctypes.byref(d)) == 1
    d.bn_fused = 1
    L.check(lib.dyk_stem_conv_wgrad(ctypes.byref(d), _stream()), "dyk_stem_conv_wgrad(bn_fused)")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dw1).all()) and float(dw0.abs().max()) > 0
    assert float((dw1 - dw0).abs().max()) <= 2e-3 * float(dw0.abs().max()), float((dw1 - dw0).abs().max()) / float(dw0.abs().max())
    assert torch.allclose(dg1, dg0, rtol=1e-6, atol=1e-5) and torch.allclose(db1, db0, rtol=1e-6, atol=1e-5)
    assert float((dg1 - 0.25 - s2.float()).abs().max()) <= 1e-4 * max(1.0, float(s2.abs().max()))
    # DYK_EW_SKIP: the separate pass does nothing
    dz.fill_(7.0)
    e.flags = L.EW_SKIP
    L.check(lib.dyk_bn_act_bwd_apply(ctypes.byref(e), _stream()), "dyk_bn_act_bwd_apply(skip)")
    assert float((dz.float() - 7.0).abs().max()) == 0.0 and torch.equal(dg0, dg1)
    # float images: not fusable, and asking for it anyway is refused
    imgf = (img.float() / 255.0).contiguous()
    d.img, d.in_u8 = imgf.data_ptr(), 0
    assert lib.dyk_stem_wgrad_bn_fusable(ctypes.byref(d)) == 0
    assert lib.dyk_stem_conv_wgrad(ctypes.byref(d), _stream()) != 0


def test_stem_rejects_unsupported_shapes():
    from dyk import lib as L
    lib = L.load()
    img = torch.zeros(1, 3, 16, 16, device="cuda")
    d = _desc(img, torch.zeros(32, 3, 3, 3), 1, torch.bfloat16)
    d.Cout = 24
    assert lib.dyk_stem_conv_fwd(ctypes.byref(d), _stream()) != 0
    d.Cout, d.k = 32, 5
    assert lib.dyk_stem_conv_fwd(ctypes.byref(d), _stream()) != 0


def test_model_accepts_uint8_and_float_batches_identically():
    """models.YOLO on the loader's uint8 batch == on `batch.float() / 255.0` (kaist_train_eval_utils.py:54-55), bit for bit"""
    from build_utils.parse_config import materialize_cfg
    from helpers import C3, oracle_net
    from models import YOLO
    torch.manual_seed(0)
    m = YOLO(materialize_cfg(C3))
    m.load_state_dict(oracle_net(C3).synth_state(0))
    m.dyk_dtype = "bf16"
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(9)
    v8 = torch.randint(0, 256, (2, 3, 64, 96), dtype=torch.uint8, generator=g).cuda()
    l8 = torch.randint(0, 256, (2, 3, 64, 96), dtype=torch.uint8, generator=g).cuda()
    with torch.no_grad():
        a, _ = m(v8, l8)
        b, _ = m(v8.float() / 255.0, l8.float() / 255.0)
    assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
