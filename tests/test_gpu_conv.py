"""GPU parity of the implicit-GEMM convolution (dyk_conv_igemm) against torch CPU fp32 conv2d.

The checker here is plain torch.nn.functional on the CPU (same arithmetic the oracle uses for
models.py:34-62); the thing under test is reached only through the C ABI.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # (B, Cin, Cout, H, W, k, stride)
    (2, 32, 64, 16, 20, 3, 1),
    (2, 64, 32, 17, 23, 3, 1),      # ragged spatial extent
    (1, 128, 128, 32, 40, 3, 1),
    (2, 64, 128, 16, 20, 3, 2),
    (2, 64, 64, 15, 21, 3, 2),      # odd size stride 2
    (2, 128, 64, 16, 20, 1, 1),
    (1, 256, 18, 16, 20, 1, 1),     # head: Cout not a multiple of 4*...
    (1, 64, 160, 8, 12, 1, 1),      # Cout tile tail (160 = 128 + 32)
    (1, 32, 32, 12, 12, 1, 2),      # 1x1 stride 2
    (1, 64, 64, 12, 12, 5, 1),      # 25 taps
]


def _ref_nhwc(t):  # NCHW cpu -> channels-last cuda handled by the product converters
    return t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_and_dgrad(case, dtype):
    from dyk import ops
    B, Cin, Cout, H, W, k, s = case
    pad = k // 2
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    if dtype == torch.bfloat16:  # the checker sees the same rounded operands
        x = x.bfloat16().float()
        w = w.bfloat16().float()
    y_ref = F.conv2d(x, w, stride=s, padding=pad)
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2

    xd = ops.to_nhwc(x.cuda(), dtype)
    wp = ops.pack_weight(w.cuda(), dtype)
    y = ops.conv2d_fwd(xd, wp, k, s, pad, Cout)
    y_nchw = ops.to_nchw(y).cpu()
    err = (y_nchw - y_ref).abs().max().item()
    assert err <= tol * max(1.0, y_ref.abs().max().item()), "fwd max err %g" % err

    # data gradient: dx = conv_transpose(dy, w)
    dy = torch.randn(y_ref.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w, dy, stride=s, padding=pad)
    cpad = (Cout + 31) // 32 * 32   # the gradient GEMM's K must be a multiple of 32 channels
    dyd = ops.to_nhwc(dy.cuda(), dtype, cpad=cpad)
    wpt = ops.pack_weight(w.cuda(), dtype, transposed=True, cout_pad=cpad)
    dx = ops.conv2d_dgrad(dyd, wpt, k, s, pad, H, W, Cin)
    dx_nchw = ops.to_nchw(dx).cpu()
    err = (dx_nchw - dx_ref).abs().max().item()
    assert err <= tol * max(1.0, dx_ref.abs().max().item()), "dgrad max err %g" % err
    # stride > 1, even maps: the parity classes in one launch (DykConvDesc.ncls) -- same arithmetic per class, same bits
    if s > 1 and H % s == 0 and W % s == 0 and k >= s:
        dx1 = ops.conv2d_dgrad(dyd, wpt, k, s, pad, H, W, Cin, merge=True)
        assert torch.equal(dx1, dx), "merged parity classes differ from the per-class launches"
    # accumulate flag: second pass adds onto the first
    ops.conv2d_dgrad(dyd, wpt, k, s, pad, H, W, Cin, out=dx, accumulate=True)
    err = (ops.to_nchw(dx).cpu() - 2 * dx_ref).abs().max().item()
    assert err <= 2.5 * tol * max(1.0, dx_ref.abs().max().item()), "dgrad accumulate max err %g" % err


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_epilogue(dtype):
    """affine + activation + residual + per-channel statistics + fp32 output + channel-slice output."""
    from dyk import ops
    B, Cin, Cout, H, W, k = 2, 64, 96, 16, 20, 3
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * 9) ** 0.5
    res = torch.randn(B, Cout, H, W, generator=g)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g)
    if dtype == torch.bfloat16:
        x, w, res = x.bfloat16().float(), w.bfloat16().float(), res.bfloat16().float()
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    conv = F.conv2d(x, w, padding=1)
    xd = ops.to_nhwc(x.cuda(), dtype)
    wp = ops.pack_weight(w.cuda(), dtype)
    resd = ops.to_nhwc(res.cuda(), dtype)
    for act, fn in [("leaky", lambda t: F.leaky_relu(t, 0.1)), ("mish", F.mish), ("relu6", F.relu6),
                    ("hard-swish", F.hardswish), ("hard-sigmoid", F.hardsigmoid), ("relu", F.relu)]:
        y_ref = fn(conv * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) + res
        # write into a channel slice [32:128] of a 160-channel buffer
        buf = torch.zeros(B, H, W, 160, dtype=dtype, device="cuda")
        out = buf[..., 32:32 + Cout]
        ops.conv2d_fwd(xd, wp, k, 1, 1, Cout, act=act, scale=scale.cuda(), shift=shift.cuda(), res=resd, out=out)
        y = ops.to_nchw(out).cpu()
        err = (y - y_ref).abs().max().item()
        assert err <= tol * max(1.0, y_ref.abs().max().item()), "%s max err %g" % (act, err)
        assert buf[..., :32].abs().max().item() == 0 and buf[..., 128:].abs().max().item() == 0
    # statistics + fp32 output
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device="cuda")
    y32 = ops.conv2d_fwd(xd, wp, k, 1, 1, Cout, stats=stats, out_f32=True)
    assert y32.dtype == torch.float32
    n = B * H * W
    s = stats.cpu()
    mean_ref = conv.mean(dim=(0, 2, 3)).double()
    sq_ref = (conv.double() ** 2).mean(dim=(0, 2, 3))
    assert (s[:Cout] / n - mean_ref).abs().max().item() < 1e-4
    assert ((s[Cout:] / n - sq_ref).abs() / sq_ref).max().item() < 1e-3
    err = (y32.permute(0, 3, 1, 2).cpu() - conv).abs().max().item()
    assert err <= 1e-3 * max(1.0, conv.abs().max().item())


WG_CASES = [
    (2, 32, 64, 16, 20, 3, 1), (2, 64, 32, 17, 23, 3, 1), (1, 128, 128, 32, 40, 3, 1), (2, 64, 128, 16, 20, 3, 2),
    (2, 64, 64, 15, 21, 3, 2), (2, 128, 64, 16, 20, 1, 1), (1, 64, 160, 8, 12, 1, 1), (3, 256, 256, 8, 10, 3, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", WG_CASES)
def test_conv_wgrad(case, dtype):
    from dyk import ops
    B, Cin, Cout, H, W, k, s = case
    pad = k // 2
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, Cin, H, W, generator=g)
    Ho, Wo = ops.conv_out_size(H, k, s, pad), ops.conv_out_size(W, k, s, pad)
    dy = torch.randn(B, Cout, Ho, Wo, generator=g)
    if dtype == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    dw_ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dy, stride=s, padding=pad)
    xd, dyd = ops.to_nhwc(x.cuda(), dtype), ops.to_nhwc(dy.cuda(), dtype)
    for splits in (0, 1, 3):
        dw = ops.conv2d_wgrad(xd, dyd, k, s, pad, splits=splits)          # [t][co][ci]
        got = dw.view(k, k, Cout, Cin).permute(2, 3, 0, 1).cpu()
        err = (got - dw_ref).abs().max().item()
        assert err <= 1e-4 * max(1.0, dw_ref.abs().max().item()), "wgrad splits=%d max err %g" % (splits, err)
    # accumulation into an existing gradient
    ops.conv2d_wgrad(xd, dyd, k, s, pad, dw=dw)
    err = (dw.view(k, k, Cout, Cin).permute(2, 3, 0, 1).cpu() - 2 * dw_ref).abs().max().item()
    assert err <= 2e-4 * max(1.0, dw_ref.abs().max().item())


def test_conv_wgrad_head_padded_ld():
    """Cout = 18 head gradient living in a ld = 32 buffer (tail channels hold garbage)."""
    from dyk import ops
    B, Cin, Cout, H, W = 2, 64, 18, 8, 10
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    dy = torch.randn(B, Cout, H, W, generator=g).bfloat16().float()
    dw_ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 1, 1), dy)
    xd = ops.to_nhwc(x.cuda(), torch.bfloat16)
    buf = torch.full((B, H, W, 32), 1e30, dtype=torch.bfloat16, device="cuda")
    ops.to_nhwc(dy.cuda(), torch.bfloat16, out=buf[..., :Cout])
    for tune in (0, (3 << 28) | 4):            # per-tap kernel, pixel-streaming kernel (round 6)
        dw = ops.conv2d_wgrad(xd, buf[..., :Cout], 1, 1, 0, tune=tune)
        err = (dw.view(Cout, Cin).cpu() - dw_ref.view(Cout, Cin)).abs().max().item()
        assert err <= 1e-4 * max(1.0, dw_ref.abs().max().item()), hex(tune)


DW_CASES = [
    # (B, C, H, W, k, stride, pad)
    (2, 64, 16, 20, 3, 1, 1),
    (2, 72, 17, 23, 3, 1, 1),       # C not a multiple of 32 (rows padded to 96), ragged extent
    (1, 120, 12, 12, 5, 1, 2),      # 25 taps
    (2, 16, 32, 40, 3, 2, 1),       # DepthwiseSeparableConv2d stride 2
    (2, 200, 9, 11, 3, 2, 1),       # odd size stride 2
    (1, 960, 4, 5, 5, 1, 2),
    (1, 40, 10, 10, 5, 1, 1),       # DepthwiseSeparableConv2d keeps padding 1 for any kernel size (layers.py:223)
    (2, 16, 40, 36, 3, 1, 1),       # LDS-tiled kernel: 2 channel vectors, 32-row tiles, ragged tile grid
    (1, 24, 50, 37, 5, 1, 2),       # 3 channel vectors per workgroup (odd group width), several tiles per image
    (1, 184, 20, 19, 5, 1, 2),      # 23 channel vectors: three groups of 8, the last one partly filled
    (2, 8, 70, 18, 3, 1, 1),        # a single channel vector (64-row tiles)
    (16, 120, 33, 50, 5, 1, 2),     # persistent tiled weight gradient: two tiles per workgroup, ragged tiles both ways
    (4, 16, 256, 320, 3, 1, 1),     # ... three tiles per workgroup at the MobileNetV3 cfg's first depthwise layer
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", DW_CASES)
def test_depthwise_conv_fwd_dgrad_wgrad(case, dtype):
    """dyk_dwconv_{fwd,dgrad,wgrad} against torch CPU conv2d(groups=C) and its autograd"""
    from dyk import ops
    B, C, H, W, k, s, pad = case
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) / k
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    x.requires_grad_(True)
    w.requires_grad_(True)
    y_ref = F.conv2d(x, w, stride=s, padding=pad, groups=C)
    dy = torch.randn(y_ref.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    y_ref.backward(dy)
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    cpad = (C + 31) // 32 * 32
    xd = ops.to_nhwc(x.detach().cuda(), dtype, cpad=cpad)
    wt = w.detach().reshape(C, k * k).t().contiguous().cuda()          # tap-major [k*k][C] fp32
    slots = 4
    stats = torch.zeros(slots, 2, C, dtype=torch.float64, device="cuda")
    y = ops.dwconv_fwd(xd, wt, k, s, pad, stats=stats, stats_slots=slots, C=C)
    y_nchw = ops.to_nchw(y, C=C).cpu()
    err = (y_nchw - y_ref.detach()).abs().max().item()
    assert err <= tol * max(1.0, y_ref.abs().max().item()), "fwd max err %g" % err
    assert float(y[..., C:].abs().max()) == 0.0 if cpad > C else True
    # statistics are taken from the fp32 accumulators (before rounding to the storage dtype)
    st = stats.sum(0).cpu()
    yr = y_ref.detach().double()
    n = B * y_ref.shape[2] * y_ref.shape[3]
    assert torch.allclose(st[0], yr.sum((0, 2, 3)), rtol=1e-4, atol=1e-3 * n ** 0.5)
    assert torch.allclose(st[1], (yr * yr).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)

    dyd = ops.to_nhwc(dy.cuda(), dtype, cpad=cpad)
    dx = ops.dwconv_dgrad(dyd, wt, k, s, pad, H, W, C=C)
    err = (ops.to_nchw(dx, C=C).cpu() - x.grad).abs().max().item()
    assert err <= tol * max(1.0, x.grad.abs().max().item()), "dgrad max err %g" % err
    ops.dwconv_dgrad(dyd, wt, k, s, pad, H, W, C=C, out=dx, accumulate=True)
    err = (ops.to_nchw(dx, C=C).cpu() - 2 * x.grad).abs().max().item()
    assert err <= 2.5 * tol * max(1.0, x.grad.abs().max().item()), "dgrad accumulate max err %g" % err

    dw = ops.dwconv_wgrad(xd, dyd, k, s, pad, C=C)
    dw_ref = w.grad.reshape(C, k * k).t()
    err = (dw.cpu() - dw_ref).abs().max().item()
    assert err <= 1e-4 * max(1.0, dw_ref.abs().max().item()), "wgrad max err %g" % err
    # the way the plan runs it: one partial plane per workgroup row, every plane fully written, same planes from run to run
    parts = ops.dwconv_wgrad(xd, dyd, k, s, pad, C=C, planes=True)
    assert not torch.isnan(parts).any(), "a partial plane was left unwritten"
    err = (parts.double().sum(0).float().cpu() - dw_ref).abs().max().item()
    assert err <= 1e-4 * max(1.0, dw_ref.abs().max().item()), "wgrad planes max err %g" % err
    assert torch.equal(parts, ops.dwconv_wgrad(xd, dyd, k, s, pad, C=C, planes=True))


@pytest.mark.parametrize("case", [(2, 64, 128, 16, 20, 3, 1), (1, 128, 96, 17, 23, 3, 1), (3, 64, 64, 9, 13, 1, 1),
                                  (2, 32, 32, 20, 24, 3, 2), (1, 192, 128, 12, 20, 3, 1), (2, 256, 64, 8, 10, 1, 1)])
def test_conv_every_tile_configuration(case):
    """every instantiation the autotuner may pick (K step x ring depth x pixel tile 80/128/160 x channel tile) gives
    the same result, with the statistics + affine/activation epilogues"""
    from dyk import ops
    from dyk.plan import _conv_candidates
    B, Cin, Cout, H, W, k, s = case
    pad = k // 2
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B, Cin, H, W, generator=g)).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g)
    y_ref = F.conv2d(x, w, stride=s, padding=pad)
    z_ref = F.leaky_relu(y_ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.1)
    xd = ops.to_nhwc(x.cuda(), torch.bfloat16)
    wp = ops.pack_weight(w.cuda(), torch.bfloat16)
    probe = ops.make_conv_desc(xd, wp, xd, Hi=H, Wi=W, Cin=Cin, Cout=Cout, Hg=1, Wg=1, Ho=1, Wo=1, taps=ops.fwd_taps(k, pad))
    cands = _conv_candidates(probe)
    assert len(cands) >= 6
    n = B * y_ref.shape[2] * y_ref.shape[3]
    for tune in cands:
        stats = torch.zeros(4, 2, Cout, dtype=torch.float64, device="cuda")
        y = ops.conv2d_fwd(xd, wp, k, s, pad, Cout, stats=stats, stats_slots=4, tune=tune)
        err = (ops.to_nchw(y).cpu() - y_ref).abs().max().item()
        assert err <= 1.2e-2 * max(1.0, y_ref.abs().max().item()), (hex(tune), err)
        st = stats.sum(0).cpu()
        assert torch.allclose(st[0], y_ref.double().sum((0, 2, 3)), rtol=1e-3, atol=2e-2 * n ** 0.5), hex(tune)
        assert torch.allclose(st[1], (y_ref.double() ** 2).sum((0, 2, 3)), rtol=2e-3, atol=1e-2), hex(tune)
        z = ops.conv2d_fwd(xd, wp, k, s, pad, Cout, act="leaky", scale=scale.cuda(), shift=shift.cuda(), tune=tune)
        err = (ops.to_nchw(z).cpu() - z_ref).abs().max().item()
        assert err <= 1.5e-2 * max(1.0, z_ref.abs().max().item()), (hex(tune), err)


@pytest.mark.parametrize("act", ["hard-swish", "relu", "linear"])
@pytest.mark.parametrize("case", [(2, 72, 17, 23, 3), (1, 120, 12, 20, 5), (2, 16, 40, 36, 3)])
def test_depthwise_dgrad_with_fused_batchnorm_backward_reduce(case, act):
    """DykDwDesc.res: the depthwise data gradient stores da = dx * act'(u*scale + shift) (u = raw output of the conv that
    produced the depthwise input) and accumulates sum(da), sum(da * xhat) -- the MobileNet expansion conv's BN backward
    reduce, folded into the launch that produces its dz (bf16, LDS-tiled kernel)."""
    import ctypes
    from dyk import lib as L
    from dyk import ops
    acts = {"hard-swish": F.hardswish, "relu": F.relu, "linear": lambda t: t}
    B, C, H, W, k = case
    pad = k // 2
    g = torch.Generator().manual_seed(21)
    w = torch.randn(C, 1, k, k, generator=g) / k
    dy = torch.randn(B, C, H, W, generator=g).bfloat16().float()
    u = (torch.randn(B, C, H, W, generator=g) * 2).bfloat16().float()
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    mean, rstd = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    dz = torch.nn.grad.conv2d_input((B, C, H, W), w, dy, padding=pad, groups=C).bfloat16().float()
    t = (u * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).requires_grad_(True)
    acts[act](t).backward(dz)
    da_ref = t.grad
    xhat = (u - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    s1_ref, s2_ref = da_ref.double().sum((0, 2, 3)), (da_ref.double() * xhat.double()).sum((0, 2, 3))
    cpad = (C + 31) // 32 * 32
    dyd = ops.to_nhwc(dy.cuda(), torch.bfloat16, cpad=cpad)
    ud = ops.to_nhwc(u.cuda(), torch.bfloat16, cpad=cpad)
    wt = w.reshape(C, k * k).t().contiguous().cuda()
    out = torch.zeros((B, H, W, cpad), dtype=torch.bfloat16, device="cuda")
    slots = 4
    red = torch.zeros(slots, 2, C, dtype=torch.float64, device="cuda")
    bn = torch.cat([scale, shift, mean, rstd]).cuda().contiguous()
    d = ops._dw_desc(out, dyd, wt, k, 1, pad, C)
    d.res, d.ldr, d.bn, d.act = ud.data_ptr(), cpad, bn.data_ptr(), ops.ACT_CODES[act]
    d.stats, d.stats_slots = red.data_ptr(), slots
    L.check(L.load().dyk_dwconv_dgrad(ctypes.byref(d), None), "dyk_dwconv_dgrad(fused BN reduce)")
    got = ops.to_nchw(out, C=C).cpu()
    err = (got - da_ref).abs().max().item()
    assert err <= 1.5e-2 * max(1.0, da_ref.abs().max().item()), err
    st = red.sum(0).cpu()
    n = B * H * W
    assert torch.allclose(st[0], s1_ref, rtol=2e-3, atol=2e-2 * n ** 0.5 * 0.1), (st[0] - s1_ref).abs().max()
    assert torch.allclose(st[1], s2_ref, rtol=2e-3, atol=2e-2 * n ** 0.5 * 0.1), (st[1] - s2_ref).abs().max()
    # the sums are those of the STORED (rounded) gradient times act', in fp32: compare against exactly that
    da_st = ops.to_nchw(out, C=C).cpu().double()
    assert torch.allclose(st[0], da_st.sum((0, 2, 3)), rtol=1e-2, atol=0.5)
    d.flags = 1
    assert L.load().dyk_dwconv_dgrad(ctypes.byref(d), None) == -3      # DYK_ERR_UNSUPPORTED: no accumulate form


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["mish", "leaky", "linear", "relu6"])
def test_dgrad_with_fused_batchnorm_backward_reduce(dtype, act):
    """DYK_EPI_BNBWD: the data gradient that produces dz of a BatchNorm+activation output stores
    da = dz * act'(y*scale + shift) and accumulates sum(da), sum(da * xhat) (the BN backward's reduce pass)."""
    import ctypes
    from dyk import lib as L
    from dyk import ops
    acts = {"mish": F.mish, "leaky": lambda t: F.leaky_relu(t, 0.1), "linear": lambda t: t, "relu6": F.relu6}
    B, Cin, Cout, H, W, k = 2, 64, 96, 12, 20, 3          # the conv whose dgrad is launched: Cin -> Cout
    g = torch.Generator().manual_seed(11)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * 9) ** 0.5)
    dy = torch.randn(B, Cout, H, W, generator=g)          # gradient wrt the conv's raw output
    y_prev = torch.randn(B, Cin, H, W, generator=g)       # raw output of the producer (the BN'd conv before)
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    if dtype == torch.bfloat16:
        w, dy, y_prev = w.bfloat16().float(), dy.bfloat16().float(), y_prev.bfloat16().float()
    mean = y_prev.mean((0, 2, 3))
    var = y_prev.var((0, 2, 3), unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    scale, shift = gamma * rstd, beta - mean * gamma * rstd
    dz = torch.nn.grad.conv2d_input((B, Cin, H, W), w, dy, padding=1)
    if dtype == torch.bfloat16:
        dz = dz.bfloat16().float()                         # the kernel rounds the gradient tile to the storage dtype first
    u = (y_prev * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).requires_grad_(True)
    acts[act](u).backward(dz)
    da_ref = u.grad
    xhat = (y_prev - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    s1_ref, s2_ref = da_ref.double().sum((0, 2, 3)), (da_ref.double() * xhat.double()).sum((0, 2, 3))

    cpad = (Cout + 31) // 32 * 32
    dyd = ops.to_nhwc(dy.cuda(), dtype, cpad=cpad)
    wpt = ops.pack_weight(w.cuda(), dtype, transposed=True, cout_pad=cpad)
    yd = ops.to_nhwc(y_prev.cuda(), dtype)
    out = torch.empty((B, H, W, Cin), dtype=dtype, device="cuda")
    slots = 4
    red = torch.zeros(slots, 2, Cin, dtype=torch.float64, device="cuda")
    vec = [t.cuda().contiguous() for t in (scale, shift, mean, rstd)]
    (py, px, Hg, Wg, taps), = ops.dgrad_classes(k, 1, 1, H, W)
    d = ops.make_conv_desc(dyd, wpt, out, Hi=H, Wi=W, Cin=cpad, Cout=Cin, Hg=Hg, Wg=Wg, Ho=H, Wo=W, taps=taps, act=act)
    d.flags = L.EPI_BNBWD
    d.res, d.ldr = yd.data_ptr(), Cin
    d.scale, d.shift, d.aux0, d.aux1 = (t.data_ptr() for t in vec)
    d.stats, d.stats_slots = red.data_ptr(), slots
    for tune in (0, 64 | (2 << 8) | (2 << 12), 64 | (2 << 8) | (4 << 12)):        # generic 128, 160-pixel tile, halo tile
        red.zero_()
        d.tune = tune
        L.check(L.load().dyk_conv_igemm(ctypes.byref(d), None), "dyk_conv_igemm(BNBWD)")
        tol = 3e-5 if dtype == torch.float32 else 1.5e-2
        got = ops.to_nchw(out).cpu()
        err = (got - da_ref).abs().max().item()
        assert err <= tol * max(1.0, da_ref.abs().max().item()), (hex(tune), err)
        st = red.sum(0).cpu()
        n = B * H * W
        assert torch.allclose(st[0], s1_ref, rtol=2e-3, atol=(5e-4 if dtype == torch.float32 else 5e-2) * n ** 0.5), hex(tune)
        assert torch.allclose(st[1], s2_ref, rtol=2e-3, atol=(5e-4 if dtype == torch.float32 else 5e-2) * n ** 0.5), hex(tune)
    # residual chain (DYK_EPI_ADDEND): dz = dgrad + gradient over the plain [shortcut]; dz itself is stored (rounded to
    # the storage dtype), the sums are those of da = dz * act'(u) taken from the rounded value
    gadd = torch.randn(B, Cin, H, W, generator=g)
    if dtype == torch.bfloat16:
        gadd = gadd.bfloat16().float()
    dzt = dz + gadd
    if dtype == torch.bfloat16:
        dzt = dzt.bfloat16().float()
    u2 = u.detach().clone().requires_grad_(True)
    acts[act](u2).backward(dzt)
    s1c, s2c = u2.grad.double().sum((0, 2, 3)), (u2.grad.double() * xhat.double()).sum((0, 2, 3))
    addd = ops.to_nhwc(gadd.cuda(), dtype)
    d.flags = L.EPI_BNBWD | L.EPI_ADDEND
    d.add = addd.data_ptr()
    for tune in (0, 64 | (2 << 8) | (1 << 12), 64 | (2 << 8) | (4 << 12)):
        red.zero_()
        out.zero_()
        d.tune = tune
        L.check(L.load().dyk_conv_igemm(ctypes.byref(d), None), "dyk_conv_igemm(BNBWD|ADDEND)")
        got = ops.to_nchw(out).cpu()
        err = (got - dzt).abs().max().item()
        assert err <= tol * max(1.0, dzt.abs().max().item()), (hex(tune), err)
        st = red.sum(0).cpu()
        assert torch.allclose(st[0], s1c, rtol=2e-3, atol=(5e-4 if dtype == torch.float32 else 5e-2) * n ** 0.5), hex(tune)
        assert torch.allclose(st[1], s2c, rtol=2e-3, atol=(5e-4 if dtype == torch.float32 else 5e-2) * n ** 0.5), hex(tune)
    # the keep-dz form WITHOUT an addend (add == NULL: the last [shortcut] of a residual chain, whose gradient the skip branch
    # still needs): the same bits as an all-zero addend tensor, on the generic tiles and on the persistent pointwise kernel
    zer = torch.zeros_like(addd)
    for tune in (0, 64 | (2 << 8) | (1 << 12), 64 | (2 << 8) | (4 << 12), (7 << 12) | (2 << 8) | (1 << 24), (7 << 12) | (3 << 8) | (1 << 25)):
        res = []
        for ptr in (zer.data_ptr(), None):
            red.zero_()
            out.zero_()
            d.tune, d.add = tune, ptr
            L.check(L.load().dyk_conv_igemm(ctypes.byref(d), None), "dyk_conv_igemm(BNBWD|ADDEND, add == NULL)")
            res.append((out.clone(), red.clone()))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), hex(tune)
    d.add = addd.data_ptr()
    # illegal flag combinations are rejected
    d.flags = L.EPI_BNBWD | L.EPI_ACCUM
    assert L.load().dyk_conv_igemm(ctypes.byref(d), None) != 0
    d.flags = L.EPI_ADDEND
    assert L.load().dyk_conv_igemm(ctypes.byref(d), None) != 0


@pytest.mark.parametrize("k,act,residual", [(1, "leaky", False), (3, "leaky", True), (3, "mish", False), (1, "linear", True)])
def test_conv_batchnorm_forward_in_one_launch(k, act, residual):
    """DYK_EPI_BNFWD: conv + train-mode BatchNorm + activation (+ the plain [shortcut] add) in ONE launch -- statistics, arrival
    counter, fold, normalise from the accumulators -- against the two-launch path it replaces (conv with statistics, then
    dyk_bn_finalize_act_fwd): raw output, normalised output, scale / shift / saved mean / rstd and running statistics are all
    BIT-IDENTICAL, for every tile shape the front end accepts; launches above dyk_conv_bnfwd_max_grid() are refused."""
    import ctypes
    from dyk import lib as L
    from dyk import ops
    lib = L.load()
    dtype = torch.bfloat16
    B, Cin, Cout, H, W = 4, 64, 160, 16, 20
    g = torch.Generator().manual_seed(31 + k)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    xd = ops.to_nhwc(x.cuda(), dtype)
    wp = ops.pack_weight(w.cuda(), dtype)
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.3).cuda()
    resd = ops.to_nhwc(torch.randn(B, Cout, H, W, generator=g).cuda(), dtype) if residual else None
    slots, n = 4, B * H * W
    zbig = torch.zeros(B, H, W, 256, dtype=dtype, device="cuda")            # y2 is a channel slice of a wider buffer

    def two_launch(tune):
        stats = torch.zeros(slots * 2 * Cout, dtype=torch.float64, device="cuda")
        y = ops.conv2d_fwd(xd, wp, k, 1, k // 2, Cout, stats=stats, stats_slots=slots, tune=tune)
        vec = [torch.zeros(Cout, device="cuda") for _ in range(4)]
        rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
        f = L.DykBnFinalizeDesc()
        f.stats, f.gamma, f.beta, f.running_mean, f.running_var = stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr()
        f.scale, f.shift, f.save_mean, f.save_rstd = (t.data_ptr() for t in vec)
        f.C, f.count, f.momentum, f.eps, f.slots = Cout, n, 0.03, 1e-4, slots
        z = torch.zeros_like(zbig)
        e = ops.ew_desc(a=y, b=resd, out=z[..., 64:64 + Cout], act=act, p0=vec[0], p1=vec[1])
        L.check(lib.dyk_bn_finalize_act_fwd(ctypes.byref(f), ctypes.byref(e), None), "dyk_bn_finalize_act_fwd")
        return y, z, vec, rm, rv

    def one_launch(tune):
        stats = torch.zeros(slots * 2 * Cout, dtype=torch.float64, device="cuda")
        y = torch.empty((B, H, W, Cout), dtype=dtype, device="cuda")
        z = torch.zeros_like(zbig)
        vec = [torch.zeros(Cout, device="cuda") for _ in range(4)]
        rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
        cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
        d = ops.make_conv_desc(xd, wp, y, Hi=H, Wi=W, Cin=Cin, Cout=Cout, Hg=H, Wg=W, Ho=H, Wo=W, taps=ops.fwd_taps(k, k // 2),
                               act=act, res=resd, stats=stats)
        d.flags |= L.EPI_BNFWD
        d.tune, d.stats_slots = tune, slots
        d.scale, d.shift, d.bn_save_mean, d.bn_save_rstd = (t.data_ptr() for t in vec)
        d.bn_gamma, d.bn_beta, d.bn_running_mean, d.bn_running_var = gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr()
        d.y2, d.ldy2, d.bn_counter = z[..., 64:64 + Cout].data_ptr(), 256, cnt.data_ptr()
        d.bn_count, d.bn_momentum, d.bn_eps = n, 0.03, 1e-4
        rc = lib.dyk_conv_igemm(ctypes.byref(d), None)
        return rc, d, (y, z, vec, rm, rv), cnt

    ran = 0
    for tune in (0, 64 | (2 << 8) | (1 << 12), 64 | (2 << 8) | (2 << 12), 128 | (2 << 8) | (2 << 12), 64 | (3 << 8) | (2 << 24),
                 64 | (2 << 8) | (2 << 12) | (2 << 24)):
        rc, d, got, cnt = one_launch(tune)
        grid = lib.dyk_conv_grid(ctypes.byref(d))
        if grid > lib.dyk_conv_bnfwd_max_grid():
            assert rc == -3, hex(tune)
            continue
        L.check(rc, "dyk_conv_igemm(BNFWD)")
        ran += 1
        ref = two_launch(tune)
        assert cnt.tolist() == [0, 0, 0, 0], (hex(tune), cnt.tolist(), grid)          # re-armed by the last workgroup to leave
        assert torch.equal(got[0], ref[0]), hex(tune)                               # raw conv output
        assert torch.equal(got[1], ref[1]), (hex(tune), (got[1].float() - ref[1].float()).abs().max().item())
        for a, b in zip(got[2], ref[2]):
            assert torch.equal(a, b), hex(tune)
        assert torch.equal(got[3], ref[3]) and torch.equal(got[4], ref[4]), hex(tune)
    assert ran >= 3
    # a launch with more workgroups than may wait on one another
    big = ops.to_nhwc(torch.randn(16, Cin, 64, 80, generator=g).cuda(), dtype)
    yb = torch.empty((16, 64, 80, Cout), dtype=dtype, device="cuda")
    d = ops.make_conv_desc(big, wp, yb, Hi=64, Wi=80, Cin=Cin, Cout=Cout, Hg=64, Wg=80, Ho=64, Wo=80, taps=ops.fwd_taps(k, k // 2),
                           act=act, stats=torch.zeros(slots * 2 * Cout, dtype=torch.float64, device="cuda"))
    d.flags |= L.EPI_BNFWD
    d.stats_slots, d.bn_count, d.ldy2 = slots, 16 * 64 * 80, Cout
    keep = [torch.zeros(Cout, device="cuda") for _ in range(2)] + [torch.empty_like(yb), torch.zeros(4, dtype=torch.int32, device="cuda")]
    d.scale, d.shift, d.y2, d.bn_counter = (t.data_ptr() for t in keep)
    assert lib.dyk_conv_igemm(ctypes.byref(d), None) == -3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_wgrad_plane_mode_and_grad_reduce(dtype):
    """DykWgradDesc.part: every K split stores its tiles into its own plane, dyk_grad_reduce folds the planes into the
    gradient buffer (accumulating).  Same values as the atomic mode, and bit-identical from run to run."""
    import ctypes
    from dyk import lib as L
    from dyk import ops
    B, Cin, Cout, H, W, k = 4, 96, 160, 24, 40, 3
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    if dtype == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    xd, dyd = ops.to_nhwc(x.cuda(), dtype), ops.to_nhwc(dy.cuda(), dtype)
    dw_atomic = ops.conv2d_wgrad(xd, dyd, k, 1, 1)                       # [k*k][Cout][Cin], fp32 atomics
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dy, padding=1).permute(2, 3, 0, 1).reshape(k * k, Cout, Cin)
    assert (dw_atomic.cpu() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()

    d = L.DykWgradDesc()
    d.x, d.dy = xd.data_ptr(), dyd.data_ptr()
    d.dtype = ops.dtype_code(dtype)
    d.ldx, d.lddy = Cin, Cout
    d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, Cin, H, W, Cout
    d.isy = d.isx = 1
    taps = ops.fwd_taps(k, 1)
    d.ntaps = len(taps)
    for i, (ty, tx, wt) in enumerate(taps):
        d.tdy[i], d.tdx[i], d.twt[i] = ty, tx, wt
    lib = L.load()
    plane = k * k * Cout * Cin
    outs = []
    for tune in (2, 3, 2 | (2 << 8), 2 | (1 << 24), 3 | (1 << 24), 2 | (2 << 8) | (1 << 24)):   # ring depth 2 / 3, K-grouped workgroups, 64x64 tile cap
        d.tune, d.splits, d.part = tune, 0, None
        splits = lib.dyk_conv_wgrad_splits(ctypes.byref(d))
        assert splits >= 2
        part = torch.full((splits * plane,), float("nan"), device="cuda")      # every plane must be fully written
        G = torch.ones(plane + 64, device="cuda")                         # reduce accumulates onto existing values
        d.dw, d.part, d.part_stride, d.splits = G.data_ptr() + 64 * 4, part.data_ptr(), plane, splits
        e = (L.DykGradReduceEntry * 1)()
        e[0].g_off, e[0].part_off, e[0].plane, e[0].n, e[0].splits, e[0].chunk_begin = 64, 0, plane, plane, splits, 0
        tab = torch.frombuffer(bytearray(bytes(e)), dtype=torch.uint8).cuda()
        runs = []
        for _ in range(2):
            G.fill_(1.0)
            part.fill_(float("nan"))
            L.check(lib.dyk_conv_wgrad(ctypes.byref(d), None), "dyk_conv_wgrad(part)")
            L.check(lib.dyk_grad_reduce(G.data_ptr(), part.data_ptr(), tab.data_ptr(), 1, (plane + 1023) // 1024, None), "dyk_grad_reduce")
            runs.append(G.clone())
        assert torch.equal(runs[0], runs[1]), "plane mode must be bit-reproducible"
        assert torch.all(runs[0][:64] == 1.0)
        got = (runs[0][64:] - 1.0).view(k * k, Cout, Cin)
        assert (got.cpu() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item(), hex(tune)
        outs.append(got)


@pytest.mark.parametrize("case", [(16, 128, 128, 64, 80, 3), (16, 1024, 512, 16, 20, 1), (16, 256, 256, 32, 40, 3)])
def test_conv_at_baseline_size(case):
    """forward, data gradient and weight gradient of three layers of the target cfg at BASELINE size (batch 16, 512x640
    input), bf16 operands, against torch CPU fp32 on the same rounded operands; plus linearity in the input
    (conv(2x) == 2 conv(x) bit for bit: scaling by a power of two commutes with every rounding)."""
    from dyk import ops
    B, Cin, Cout, H, W, k = case
    pad = k // 2
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    dy = torch.randn(B, Cout, H, W, generator=g).bfloat16().float()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    y_ref = F.conv2d(x, w, padding=pad)
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w, dy, padding=pad)
    dw_ref = torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=pad)
    xd, dyd = ops.to_nhwc(x.cuda(), torch.bfloat16), ops.to_nhwc(dy.cuda(), torch.bfloat16)
    wp = ops.pack_weight(w.cuda(), torch.bfloat16)
    wpt = ops.pack_weight(w.cuda(), torch.bfloat16, transposed=True)
    y = ops.conv2d_fwd(xd, wp, k, 1, pad, Cout)
    err = (ops.to_nchw(y).cpu() - y_ref).abs().max().item()
    assert err <= 1.2e-2 * max(1.0, y_ref.abs().max().item()), err
    y2 = ops.conv2d_fwd(ops.to_nhwc((2 * x).cuda(), torch.bfloat16), wp, k, 1, pad, Cout)
    assert torch.equal(y2.float(), 2 * y.float())
    dx = ops.conv2d_dgrad(dyd, wpt, k, 1, pad, H, W, Cin)
    err = (ops.to_nchw(dx).cpu() - dx_ref).abs().max().item()
    assert err <= 1.2e-2 * max(1.0, dx_ref.abs().max().item()), err
    dw = ops.conv2d_wgrad(xd, dyd, k, 1, pad)
    ref = dw_ref.permute(2, 3, 0, 1).reshape(k * k, Cout, Cin)
    err = (dw.cpu() - ref).abs().max().item()
    assert err <= 2e-3 * ref.abs().max().item(), err
    # round 6: the kernels the plan runs at this size -- row-block (3x3) / pixel-streaming (1x1) -- with one split (single writer,
    # read-add-write) and with the kernel's own split count (atomics); run twice, the single-split result is bit-reproducible;
    # linear in dy: wgrad(x, 2 dy) == 2 wgrad(x, dy) bit for bit
    tune = (2 | (1 << 8) | (2 << 28)) if k == 3 else ((3 << 28) | 3)
    one = [ops.conv2d_wgrad(xd, dyd, k, 1, pad, tune=tune | (1 << 20), splits=1) for _ in range(2)]
    assert torch.equal(one[0], one[1])
    assert (one[0].cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    auto = ops.conv2d_wgrad(xd, dyd, k, 1, pad, tune=tune)
    assert (auto.cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    two = ops.conv2d_wgrad(xd, ops.to_nhwc((2 * dy).cuda(), torch.bfloat16), k, 1, pad, tune=tune | (1 << 20), splits=1)
    assert torch.equal(two, 2 * one[0])


@pytest.mark.parametrize("case", [
    # (B, Cin, Cout, H, W, stride): H, W = input size
    (2, 32, 64, 12, 64, 1),       # KW 64, BN 32
    (2, 64, 64, 10, 32, 1),       # KW 32, BN 64
    (1, 64, 128, 9, 96, 1),       # KW 32 (96 % 64 != 0), two Cout tiles
    (2, 32, 64, 16, 128, 2),      # stride 2, KW 64
    (2, 64, 128, 12, 64, 2),      # stride 2, KW 32, BN 64
    (1, 24, 40, 7, 64, 1),        # channel counts off the tiles
    (1, 96, 64, 6, 64, 1),        # two Cin tiles, the second ragged
    (2, 128, 128, 16, 80, 1),     # round 3: rows that are not whole 32-pixel segments (the 64x80 maps: 2.5 segments)
    (2, 64, 128, 8, 40, 1),       # 32x40 maps: 1.25 segments
    (3, 64, 64, 5, 20, 1),        # 16x20 maps: one ragged segment per row
    (2, 64, 128, 16, 80, 2),      # stride 2 onto a 40-pixel row
])
def test_wgrad_multi_tap_kernel(case):
    """tune bit 28: dy and the x halo tile staged once for all nine taps (bf16, 3x3 / pad 1) -- atomic mode and plane
    mode against torch's conv2d_weight on the same rounded operands."""
    import ctypes
    from dyk import lib as L
    from dyk import ops
    B, Cin, Cout, H, W, s = case
    k, dtype = 3, torch.bfloat16
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    dy = torch.randn(B, Cout, Ho, Wo, generator=g).bfloat16().float()
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dy, stride=s, padding=1).permute(2, 3, 0, 1).reshape(9, Cout, Cin)
    xd, dyd = ops.to_nhwc(x.cuda(), dtype), ops.to_nhwc(dy.cuda(), dtype)
    tol = 2e-4 * ref.abs().max().item()
    base = ops.conv2d_wgrad(xd, dyd, k, s, 1)
    mt = ops.conv2d_wgrad(xd, dyd, k, s, 1, tune=2 | (1 << 28))
    assert (base.cpu() - ref).abs().max().item() <= tol
    assert (mt.cpu() - ref).abs().max().item() <= tol, (mt.cpu() - ref).abs().max().item()
    # plane mode
    d = L.DykWgradDesc()
    d.x, d.dy = xd.data_ptr(), dyd.data_ptr()
    d.dtype = ops.dtype_code(dtype)
    d.ldx, d.lddy = ops.nhwc_ld(xd), ops.nhwc_ld(dyd)
    d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, Cin, Ho, Wo, Cout
    d.isy = d.isx = s
    taps = ops.fwd_taps(k, 1)
    d.ntaps = len(taps)
    for i, (ty, tx, wt) in enumerate(taps):
        d.tdy[i], d.tdx[i], d.twt[i] = ty, tx, wt
    d.tune = 2 | (1 << 28)
    lib = L.load()
    splits = lib.dyk_conv_wgrad_splits(ctypes.byref(d))
    assert splits >= 1
    plane = 9 * Cout * Cin
    part = torch.full((splits * plane,), float("nan"), device="cuda")
    G = torch.zeros(plane, device="cuda")
    d.dw, d.part, d.part_stride, d.splits = G.data_ptr(), part.data_ptr(), plane, splits
    L.check(lib.dyk_conv_wgrad(ctypes.byref(d), None), "dyk_conv_wgrad(multi-tap, planes)")
    got = part.view(splits, 9, Cout, Cin).sum(0)
    assert bool(torch.isfinite(got).all())
    assert (got.cpu() - ref).abs().max().item() <= tol



@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # (B, Cin, Cout, Hi, Wi, stride, kkw): K step = nimg x rows x columns of output pixels
    (2, 32, 64, 16, 16, 1, 1),      # 16 x 8 rows ... one image per step
    (2, 128, 128, 16, 80, 1, 1),    # the 64x80 maps: 8-column blocks
    (2, 64, 128, 16, 40, 1, 1),     # the 32x40 maps
    (4, 64, 64, 16, 20, 1, 1),      # the 16x20 maps: 4-column blocks, two images per step
    (2, 96, 192, 8, 32, 1, 1),      # three Cin tiles, three Cout tiles
    (2, 24, 40, 16, 16, 1, 1),      # channel counts off the tiles (zero page, guarded stores)
    (2, 32, 64, 32, 32, 2, 1),      # stride 2 (halo of 2 x rows + 1)
    (2, 64, 128, 16, 80, 2, 1),     # stride 2 onto a 40-pixel row
    (2, 64, 64, 16, 32, 1, 2),      # 256-pixel steps
    (2, 128, 64, 32, 40, 1, 2),     # 256-pixel steps on a 40-pixel row
    (1, 32, 64, 128, 160, 1, 1),    # many steps per workgroup, K splits
])
def test_wgrad_row_block_kernel(case):
    """tune bits 28-30 == 2 (conv_wgrad_rb.hip): a K step is a block of 128 / 256 output pixels whose dy tile and x halo tile
    are staged once for all nine taps; 8 waves = 2 channel halves x 4 K-quarters folded through LDS.  Atomic mode and
    plane mode against torch's conv2d_weight on the same rounded operands; the kernel must be the one that ran."""
    import ctypes
    from dyk import lib as L
    from dyk import ops
    B, Cin, Cout, H, W, s, kkw = case
    k, dtype = 3, torch.bfloat16
    g = torch.Generator().manual_seed(37)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    dy = torch.randn(B, Cout, Ho, Wo, generator=g).bfloat16().float()
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dy, stride=s, padding=1).permute(2, 3, 0, 1).reshape(9, Cout, Cin)
    xd, dyd = ops.to_nhwc(x.cuda(), dtype), ops.to_nhwc(dy.cuda(), dtype)
    tol = 2e-4 * ref.abs().max().item()
    tune = 2 | (kkw << 8) | (2 << 28)
    d = L.DykWgradDesc()
    d.x, d.dy = xd.data_ptr(), dyd.data_ptr()
    d.dtype = ops.dtype_code(dtype)
    d.ldx, d.lddy = ops.nhwc_ld(xd), ops.nhwc_ld(dyd)
    d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, Cin, Ho, Wo, Cout
    d.isy = d.isx = s
    taps = ops.fwd_taps(k, 1)
    d.ntaps = len(taps)
    for i, (ty, tx, wt) in enumerate(taps):
        d.tdy[i], d.tdx[i], d.twt[i] = ty, tx, wt
    d.tune = tune
    lib = L.load()
    assert lib.dyk_conv_wgrad_variant(ctypes.byref(d)) == 2, "the row-block kernel does not cover this case"
    rb = ops.conv2d_wgrad(xd, dyd, k, s, 1, tune=tune)
    err = (rb.cpu() - ref).abs().max().item()
    assert err <= tol, (err, tol)
    # one K split with the caller's word that dw has no other writer (tune bit 20): read-add-write instead of atomics --
    # it ACCUMULATES like the atomic form
    acc0 = torch.full((9, Cout, Cin), 0.5, device="cuda")
    one = ops.conv2d_wgrad(xd, dyd, k, s, 1, tune=tune | (1 << 20), splits=1, dw=acc0.clone())
    assert ((one - 0.5).cpu() - ref).abs().max().item() <= tol
    for want in (0, 3):             # the kernel's own split count, then a given one
        d.splits, d.part, d.part_stride = want, None, 0
        splits = lib.dyk_conv_wgrad_splits(ctypes.byref(d))
        assert splits >= 1 and (want == 0 or splits <= want)
        plane = 9 * Cout * Cin
        part = torch.full((max(splits, want) * plane,), float("nan"), device="cuda")
        G = torch.zeros(plane, device="cuda")
        d.dw, d.part, d.part_stride, d.splits = G.data_ptr(), part.data_ptr(), plane, max(splits, want)
        L.check(lib.dyk_conv_wgrad(ctypes.byref(d), None), "dyk_conv_wgrad(row-block, planes)")
        got = part.view(-1, 9, Cout, Cin).sum(0)
        assert bool(torch.isfinite(got).all())
        assert (got.cpu() - ref).abs().max().item() <= tol



PS_CASES = [
    # (B, Cin, Cout, H, W)
    (2, 128, 128, 16, 20),      # one 128 x 128 tile, 640 pixels = 10 stages
    (2, 256, 128, 16, 20),      # two tiles along Cin
    (3, 128, 256, 9, 13),       # two tiles along Cout, ragged pixel count (351: the last stage is part zero page)
    (2, 64, 64, 32, 40),        # 64 x 64 tile, 128-pixel (and 64-pixel) stages
    (2, 64, 128, 16, 20),       # 128 x 64 tile
    (2, 128, 64, 16, 20),       # 64 x 128 tile
    (2, 72, 24, 8, 10),         # channel counts off the tiles (zero page rows, guarded stores); Cout = 18 in a padded row: test_conv_wgrad_head_padded_ld
    (1, 512, 256, 16, 20),      # 2 x 4 tiles
    (1, 32, 96, 8, 8),          # fewer pixels than one ring (64 pixels: a single stage)
    (2, 1024, 512, 4, 5),       # deep-stage channel counts, 40 pixels: less than one stage
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", PS_CASES, ids=lambda c: "b%d_c%d_%d_%dx%d" % c)
def test_wgrad_pixel_streaming_kernel(case):
    """tune bits 28-30 == 3 (conv_wgrad_ps.hip, round 6): the 1x1 weight gradient as a pixel stream -- 8-wave workgroups
    (2 K-halves x 2 x 2 waves), LDS-DMA ring of 2-8 stages, symmetric K-half exchange.  Every ring depth / tile cap / stage
    length, in atomic, single-writer, plane and in-launch-fold mode, against torch's conv2d_weight on the same rounded
    operands; the kernel must be the one that ran; plane and fold results are bit-reproducible."""
    import ctypes
    from dyk import lib as L
    from dyk import ops
    B, Cin, Cout, H, W = case
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    dy = torch.randn(B, Cout, H, W, generator=g).bfloat16().float()
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 1, 1), dy).reshape(Cout, Cin)
    xd, dyd = ops.to_nhwc(x.cuda(), dtype), ops.to_nhwc(dy.cuda(), dtype)
    tol = 2e-4 * ref.abs().max().item()
    lib = L.load()

    def desc(tune, splits=0):
        d = L.DykWgradDesc()
        d.x, d.dy = xd.data_ptr(), dyd.data_ptr()
        d.dtype = ops.dtype_code(dtype)
        d.ldx, d.lddy = ops.nhwc_ld(xd), ops.nhwc_ld(dyd)
        d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, Cin, H, W, Cout
        d.isy = d.isx = 1
        d.ntaps = 1
        d.tdy[0] = d.tdx[0] = d.twt[0] = 0
        d.tune, d.splits = tune, splits
        return d

    PS = 3 << 28
    tunes = [PS | ns for ns in (2, 3, 4, 6, 8)] + [PS | ns | (1 << 8) for ns in (2, 4)] + [PS | ns | (1 << 8) | (1 << 12) for ns in (2, 4, 8)]
    for tune in tunes:
        assert lib.dyk_conv_wgrad_variant(ctypes.byref(desc(tune))) == 3, "the pixel-streaming kernel does not cover this case"
        got = ops.conv2d_wgrad(xd, dyd, 1, 1, 0, tune=tune).view(Cout, Cin)              # atomics, the kernel's own split count
        assert (got.cpu() - ref).abs().max().item() <= tol, hex(tune)
        # one split + the caller's word that dw has no other writer: read-add-write, ACCUMULATES
        acc0 = torch.full((1, Cout, Cin), 0.5, device="cuda")
        one = ops.conv2d_wgrad(xd, dyd, 1, 1, 0, tune=tune | (1 << 20), splits=1, dw=acc0.clone()).view(Cout, Cin)
        assert ((one - 0.5).cpu() - ref).abs().max().item() <= tol, hex(tune)
    for tune in (PS | 4, PS | 4 | (1 << 8)):
        for want in (0, 3, 7):
            # planes: exactly `want` of them when given (trailing empty ones hold zeros)
            d = desc(tune, want)
            splits = lib.dyk_conv_wgrad_splits(ctypes.byref(d))
            assert splits >= 1 and (want == 0 or splits <= want)
            n = max(splits, want)
            plane = Cout * Cin
            runs = []
            for _ in range(2):
                part = torch.full((n * plane,), float("nan"), device="cuda")
                G = torch.zeros(plane, device="cuda")
                d.dw, d.part, d.part_stride, d.splits = G.data_ptr(), part.data_ptr(), plane, n
                L.check(lib.dyk_conv_wgrad(ctypes.byref(d), None), "dyk_conv_wgrad(pixel-streaming, planes)")
                got = part.view(n, Cout, Cin).sum(0)
                assert bool(torch.isfinite(got).all())
                assert (got.cpu() - ref).abs().max().item() <= tol
                assert float(G.abs().max()) == 0.0
                runs.append(part.clone())
            assert torch.equal(runs[0], runs[1])
        for want in (2, 3, 5):
            # in-launch fold: S slices per tile through slabs + ticket; the last arriver adds them in slice order into dw
            d = desc(tune | (1 << 20), want)
            nt = ctypes.c_int32(0)
            need = int(lib.dyk_conv_wgrad_fold_ws_bytes(ctypes.byref(d), ctypes.byref(nt)))
            if need == 0:
                continue                               # fewer stages than slices: nothing to fold (single split)
            assert need > 0 and nt.value >= 1
            ws = torch.full((need // 4,), float("nan"), device="cuda")
            cnt = torch.zeros(nt.value, dtype=torch.int32, device="cuda")
            d.sk_ws, d.sk_ws_bytes, d.sk_cnt, d.sk_cnt_n = ws.data_ptr(), need, cnt.data_ptr(), nt.value
            runs = []
            for _ in range(3):                         # (the counters re-arm themselves: three launches on one scratch set)
                G = torch.full((Cout, Cin), 0.25, device="cuda")
                d.dw = G.data_ptr()
                L.check(lib.dyk_conv_wgrad(ctypes.byref(d), None), "dyk_conv_wgrad(pixel-streaming, fold)")
                assert ((G - 0.25).cpu() - ref).abs().max().item() <= tol, (hex(tune), want)
                runs.append(G.clone())
            assert int(cnt.abs().max()) == 0
            assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    # not eligible: 3x3, stride 2, fp32 -> the tune word falls back to the per-tap kernel
    d3 = desc(PS | 4)
    d3.ntaps = 9
    assert lib.dyk_conv_wgrad_variant(ctypes.byref(d3)) == 0
    d2 = desc(PS | 4)
    d2.isy = d2.isx = 2
    assert lib.dyk_conv_wgrad_variant(ctypes.byref(d2)) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("k,case", [(1, (2, 128, 128, 16, 20)), (1, (2, 64, 64, 32, 40)), (1, (1, 256, 72, 9, 13)),
                                    (3, (2, 64, 128, 16, 40)), (3, (4, 64, 64, 16, 20)), (3, (2, 24, 40, 16, 16))])
def test_grouped_weight_gradient_launch(k, case):
    """DykWgradDesc.group (round 6): G problems of one geometry in ONE launch of the pixel-streaming (1x1) / row-block (3x3)
    kernel -- the repeated units of a stage.  Every member's planes (and, with one split, its read-add-written gradient) equal
    those of its own launch with the same split count BIT FOR BIT; the other kernels refuse a group."""
    import ctypes
    from dyk import lib as L
    from dyk import ops
    B, Cin, Cout, H, W = case
    dtype = torch.bfloat16
    G = 3
    g = torch.Generator().manual_seed(43)
    xs = [ops.to_nhwc(torch.randn(B, Cin, H, W, generator=g).cuda(), dtype) for _ in range(G)]
    dys = [ops.to_nhwc(torch.randn(B, Cout, H, W, generator=g).cuda(), dtype) for _ in range(G)]
    lib = L.load()
    tune = ((3 << 28) | 4 | (1 << 20)) if k == 1 else (2 | (1 << 8) | (2 << 28) | (1 << 20))
    plane = k * k * Cout * Cin

    def desc(i, splits):
        d = L.DykWgradDesc()
        d.x, d.dy = xs[i].data_ptr(), dys[i].data_ptr()
        d.dtype = ops.dtype_code(dtype)
        d.ldx, d.lddy = ops.nhwc_ld(xs[i]), ops.nhwc_ld(dys[i])
        d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, Cin, H, W, Cout
        d.isy = d.isx = 1
        taps = ops.fwd_taps(k, k // 2)
        d.ntaps = len(taps)
        for q, (ty, tx, wt) in enumerate(taps):
            d.tdy[q], d.tdx[q], d.twt[q] = ty, tx, wt
        d.tune, d.splits = tune, splits
        return d

    assert lib.dyk_conv_wgrad_variant(ctypes.byref(desc(0, 0))) == (3 if k == 1 else 2)
    for S in (1, 3):
        # every member alone
        alone, alone_g = [], []
        for i in range(G):
            d = desc(i, S)
            Gd = torch.full((plane,), 0.5, device="cuda")
            d.dw = Gd.data_ptr()
            if S > 1:
                part = torch.full((S * plane,), float("nan"), device="cuda")
                d.part, d.part_stride = part.data_ptr(), plane
                alone.append(part)
            L.check(lib.dyk_conv_wgrad(ctypes.byref(d), None), "single launch")
            alone_g.append(Gd)
        # the group
        members = [desc(i, S) for i in range(G)]
        grads = [torch.full((plane,), 0.5, device="cuda") for _ in range(G)]
        parts = [torch.full((S * plane,), float("nan"), device="cuda") for _ in range(G)] if S > 1 else None
        arr = (L.DykWgradGroupEntry * G)()
        for i, m in enumerate(members):
            m.dw = grads[i].data_ptr()
            if S > 1:
                m.part, m.part_stride = parts[i].data_ptr(), plane
            arr[i].x, arr[i].dy, arr[i].dw, arr[i].part = m.x, m.dy, m.dw, m.part
        tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
        lead = members[-1]
        lead.group, lead.group_n = tab.data_ptr(), G
        L.check(lib.dyk_conv_wgrad(ctypes.byref(lead), None), "grouped launch")
        for i in range(G):
            if S > 1:
                assert torch.equal(parts[i], alone[i]), (S, i)
                assert float((grads[i] - 0.5).abs().max()) == 0.0
            else:
                assert torch.equal(grads[i], alone_g[i]), (S, i)
                ref = torch.nn.grad.conv2d_weight(ops.to_nchw(xs[i]).cpu(), (Cout, Cin, k, k),
                                                  ops.to_nchw(dys[i]).cpu(), padding=k // 2)
                ref = ref.permute(2, 3, 0, 1).reshape(-1)
                assert ((grads[i] - 0.5).cpu() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
    # the per-tap kernel does not take groups
    bad = desc(0, 1)
    bad.tune = 2
    bad.dw = grads[0].data_ptr()
    bad.group, bad.group_n = tab.data_ptr(), G
    assert lib.dyk_conv_wgrad(ctypes.byref(bad), None) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("act", ["mish", "leaky", "relu6"])
def test_small_channel_dgrad_kernel(stride, act):
    """pixel-tile code 6 (conv_sc.hip): the 3x3 data gradient into a 32-channel tensor with the fused BatchNorm-backward
    epilogue -- resident weights, one gradient patch per tile for all taps and (stride 2) all four parity classes in one
    workgroup, epilogue from the accumulators.  Against torch autograd on the same rounded operands, and against the generic
    kernel's sums; bit 23 of the tune word forbids the fallback, so the kernel under test is the one that ran."""
    import ctypes
    from dyk import lib as L
    from dyk import ops
    acts = {"mish": F.mish, "leaky": lambda t: F.leaky_relu(t, 0.1), "relu6": F.relu6}
    dtype = torch.bfloat16
    B, Cin, Cout, k = 3, 32, 64, 3                           # the conv whose data gradient is launched: 32 -> 64
    H, W = (16, 48) if stride == 1 else (32, 64)             # its input; the gradient grid is 16 x 48 | 16 x 32
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    g = torch.Generator().manual_seed(41)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * 9) ** 0.5).bfloat16().float()
    dy = torch.randn(B, Cout, Ho, Wo, generator=g).bfloat16().float()
    y_prev = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    mean = y_prev.mean((0, 2, 3))
    rstd = (y_prev.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt()
    scale, shift = gamma * rstd, beta - mean * gamma * rstd
    dz = torch.nn.grad.conv2d_input((B, Cin, H, W), w, dy, stride=stride, padding=1).bfloat16().float()
    u = (y_prev * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).requires_grad_(True)
    acts[act](u).backward(dz)
    da_ref = u.grad
    xhat = (y_prev - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    s1_ref, s2_ref = da_ref.double().sum((0, 2, 3)), (da_ref.double() * xhat.double()).sum((0, 2, 3))

    dyd = ops.to_nhwc(dy.cuda(), dtype)
    wpt = ops.pack_weight(w.cuda(), dtype, transposed=True)
    yd = ops.to_nhwc(y_prev.cuda(), dtype)
    out = torch.empty((B, H, W, Cin), dtype=dtype, device="cuda")
    slots = 4
    red = torch.zeros(slots, 2, Cin, dtype=torch.float64, device="cuda")
    vec = [t.cuda().contiguous() for t in (scale, shift, mean, rstd)]
    classes = ops.dgrad_classes(k, 1, stride, H, W)
    d = ops.make_conv_desc(dyd, wpt, out, Hi=Ho, Wi=Wo, Cin=Cout, Cout=Cin, Hg=classes[0][2], Wg=classes[0][3], Ho=H, Wo=W,
                           taps=[t for c in classes for t in c[4]], osy=stride, osx=stride, act=act)
    if stride == 2:
        d.ncls, q0 = len(classes), 0
        for c, (py, px, _, _, taps) in enumerate(classes):
            d.cls_first[c], d.cls_ntaps[c], d.cls_ooy[c], d.cls_oox[c] = q0, len(taps), py, px
            q0 += len(taps)
    d.flags = L.EPI_BNBWD
    d.res, d.ldr = yd.data_ptr(), Cin
    d.scale, d.shift, d.aux0, d.aux1 = (t.data_ptr() for t in vec)
    d.stats, d.stats_slots = red.data_ptr(), slots
    res = {}
    for name, tune in (("generic", 0), ("sc", (6 << 12) | (1 << 23))):
        red.zero_()
        out.fill_(float("nan"))
        d.tune = tune
        L.check(L.load().dyk_conv_igemm(ctypes.byref(d), None), "dyk_conv_igemm(%s)" % name)
        got = ops.to_nchw(out).float().cpu()
        assert bool(torch.isfinite(got).all()), name
        err = (got - da_ref).abs().max().item()
        assert err <= 1.5e-2 * max(1.0, da_ref.abs().max().item()), (name, err)
        st = red.sum(0).cpu()
        n = B * H * W
        assert torch.allclose(st[0], s1_ref, rtol=2e-3, atol=5e-2 * n ** 0.5), name
        assert torch.allclose(st[1], s2_ref, rtol=2e-3, atol=5e-2 * n ** 0.5), name
        res[name] = (got, st)
    # same arithmetic as the generic kernel (gradient rounded to bf16 before act'; the MFMA summation order differs, which
    # flips single roundings): outputs agree to one bf16 ulp of the largest value, the fp64 sums accordingly
    assert (res["sc"][0] - res["generic"][0]).abs().max().item() <= 2.0 ** -7 * max(1.0, da_ref.abs().max().item())
    assert torch.allclose(res["sc"][1], res["generic"][1], rtol=1e-3, atol=2e-2 * n ** 0.5)


LT_CASES = [  # (B, Cin, Cout, H, W): forward conv Cin -> Cout, 3x3 / stride 1 / pad 1
    (2, 64, 128, 16, 20), (1, 96, 192, 8, 40), (2, 128, 64, 16, 40), (1, 160, 96, 16, 80), (3, 256, 128, 8, 20),
]


@pytest.mark.parametrize("case", LT_CASES)
def test_large_tile_conv_kernels(case):
    """csrc/conv_lt_kernel.h: the 8-wave large-tile 3x3 kernels (128 x 320, 256 x 160, 128 x 160 with two K-groups;
    halo patches of every admissible width) against torch CPU fp32 on the same bf16-rounded operands: forward with
    statistics, forward with affine + activation + residual, data gradient with the fused BatchNorm-backward epilogue (plain
    and residual-chain form).  Tune bit 23 makes the front end refuse instead of falling back to the generic tiles, so every
    configuration counted below really ran the large-tile kernel; ragged channel tiles (Cout = 192, 136) included."""
    import ctypes
    from dyk import lib as L
    from dyk import ops
    B, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).bfloat16().float()
    r = torch.randn(B, Cout, H, W, generator=g).bfloat16().float()
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    y_ref = F.conv2d(x, w, padding=1)
    z_ref = F.leaky_relu(y_ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.1) + r
    xd, rd = ops.to_nhwc(x.cuda(), torch.bfloat16), ops.to_nhwc(r.cuda(), torch.bfloat16)
    wp = ops.pack_weight(w.cuda(), torch.bfloat16)
    # data gradient of the same conv, producing dz of a BatchNorm + Mish block whose raw output is u
    dy = torch.randn(B, Cout, H, W, generator=g).bfloat16().float()
    u = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    gadd = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    mean, var = u.mean((0, 2, 3)), u.var((0, 2, 3), unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    sc2, sh2 = gamma * rstd, beta - mean * gamma * rstd
    xhat = (u - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    dz = torch.nn.grad.conv2d_input((B, Cin, H, W), w, dy, padding=1)

    def bn_ref(dzv):
        t = (u * sc2.view(1, -1, 1, 1) + sh2.view(1, -1, 1, 1)).requires_grad_(True)
        F.mish(t).backward(dzv)
        return t.grad, t.grad.double().sum((0, 2, 3)), (t.grad.double() * xhat.double()).sum((0, 2, 3))
    da_ref, s1_ref, s2_ref = bn_ref(dz.bfloat16().float())
    dzc = (dz + gadd).bfloat16().float()
    _, s1c, s2c = bn_ref(dzc)
    dyd, ud, addd = (ops.to_nhwc(t.cuda(), torch.bfloat16) for t in (dy, u, gadd))
    wpt = ops.pack_weight(w.cuda(), torch.bfloat16, transposed=True)
    out = torch.empty((B, H, W, Cin), dtype=torch.bfloat16, device="cuda")
    red = torch.zeros(4, 2, Cin, dtype=torch.float64, device="cuda")
    vec = [t.cuda().contiguous() for t in (sc2, sh2, mean, rstd)]
    (py, px, Hg, Wg, taps), = ops.dgrad_classes(3, 1, 1, H, W)
    d = ops.make_conv_desc(dyd, wpt, out, Hi=H, Wi=W, Cin=Cout, Cout=Cin, Hg=Hg, Wg=Wg, Ho=H, Wo=W, taps=taps, act="mish")
    d.res, d.ldr = ud.data_ptr(), Cin
    d.scale, d.shift, d.aux0, d.aux1 = (t.data_ptr() for t in vec)
    d.stats, d.stats_slots, d.add = red.data_ptr(), 4, addd.data_ptr()
    n = B * H * W
    ran_f = ran_b = 0
    for shape in (1, 2, 3):
        for tw in (0, 1, 2, 3, 4):
            tune = (5 << 12) | (shape << 8) | (tw << 24) | (1 << 23)
            # ---- forward
            stats = torch.zeros(4, 2, Cout, dtype=torch.float64, device="cuda")
            try:
                y = ops.conv2d_fwd(xd, wp, 3, 1, 1, Cout, stats=stats, stats_slots=4, tune=tune)
            except L.DykError:
                y = None                                   # this shape / patch width does not fit the problem
            if y is not None:
                ran_f += 1
                err = (ops.to_nchw(y).cpu() - y_ref).abs().max().item()
                assert err <= 1.2e-2 * max(1.0, y_ref.abs().max().item()), (hex(tune), err)
                st = stats.sum(0).cpu()
                assert torch.allclose(st[0], y_ref.double().sum((0, 2, 3)), rtol=1e-3, atol=2e-2 * n ** 0.5), hex(tune)
                assert torch.allclose(st[1], (y_ref.double() ** 2).sum((0, 2, 3)), rtol=2e-3, atol=1e-2), hex(tune)
                z = ops.conv2d_fwd(xd, wp, 3, 1, 1, Cout, act="leaky", scale=scale.cuda(), shift=shift.cuda(), res=rd, tune=tune)
                err = (ops.to_nchw(z).cpu() - z_ref).abs().max().item()
                assert err <= 1.5e-2 * max(1.0, z_ref.abs().max().item()), (hex(tune), err)
            # ---- data gradient + fused BatchNorm-backward reduce, plain and residual chain
            for flags, ref, r1, r2 in ((L.EPI_BNBWD, da_ref, s1_ref, s2_ref), (L.EPI_BNBWD | L.EPI_ADDEND, dzc, s1c, s2c)):
                d.flags, d.tune = flags, tune
                red.zero_()
                out.zero_()
                rc = L.load().dyk_conv_igemm(ctypes.byref(d), None)
                if rc == -3:                            # DYK_ERR_UNSUPPORTED: this shape / patch width does not fit
                    continue
                assert rc == 0, (hex(tune), rc)
                ran_b += 1
                err = (ops.to_nchw(out).cpu() - ref).abs().max().item()
                assert err <= 1.5e-2 * max(1.0, ref.abs().max().item()), (hex(tune), flags, err)
                st = red.sum(0).cpu()
                assert torch.allclose(st[0], r1, rtol=2e-3, atol=5e-2 * n ** 0.5), (hex(tune), flags)
                assert torch.allclose(st[1], r2, rtol=2e-3, atol=5e-2 * n ** 0.5), (hex(tune), flags)
    assert ran_f >= 2 and ran_b >= 4, (ran_f, ran_b)


SPLITK_CASES = [  # (B, Cin, Cout, H, W, k, dtype)
    (2, 256, 128, 16, 20, 3, torch.bfloat16), (1, 512, 96, 8, 20, 1, torch.bfloat16), (2, 192, 256, 8, 20, 3, torch.bfloat16),
    (1, 256, 64, 16, 40, 3, torch.bfloat16), (2, 128, 64, 8, 12, 1, torch.float32), (2, 256, 512, 8, 20, 1, torch.bfloat16),
    (1, 128, 256, 8, 12, 3, torch.float32),
]


@pytest.mark.parametrize("case", SPLITK_CASES, ids=lambda c: "b%d_c%d_%d_%dx%d_k%d_%s" % (c[:6] + (str(c[6]).split(".")[-1],)))
def test_conv_split_k_across_workgroups(case):
    """DykConvDesc.splitk (round 5; deep stages of reference models.py:34-62): S workgroups per output tile, slices folded through
    private fp32 slabs in slice order by whichever workgroup arrives last.  Every (tile configuration, S) the autotuner may
    pick -- generic tiles, K-grouped tiles, the 8-wave large-tile kernels -- against torch CPU fp32 on the same rounded
    operands: forward with statistics, forward with affine + activation + residual, data gradient with the fused
    BatchNorm-backward epilogue (plain and residual chain); results bit-identical from launch to launch while another stream
    keeps the chip busy (the sum must not depend on arrival order), tile counters back at zero."""
    import ctypes
    from dyk import lib as L
    from dyk import ops
    from dyk.plan import _conv_split_candidates
    B, Cin, Cout, H, W, k, dtype = case
    pad = k // 2
    g = torch.Generator().manual_seed(11)

    def rnd(*shape):
        t = torch.randn(*shape, generator=g)
        return t.bfloat16().float() if dtype == torch.bfloat16 else t
    x = rnd(B, Cin, H, W)
    w = rnd(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    w = w.bfloat16().float() if dtype == torch.bfloat16 else w
    r = rnd(B, Cout, H, W)
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    y_ref = F.conv2d(x, w, padding=pad)
    z_ref = F.leaky_relu(y_ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.1) + r
    xd, rd = ops.to_nhwc(x.cuda(), dtype), ops.to_nhwc(r.cuda(), dtype)
    wp = ops.pack_weight(w.cuda(), dtype)
    probe = ops.make_conv_desc(xd, wp, xd, Hi=H, Wi=W, Cin=Cin, Cout=Cout, Hg=H, Wg=W, Ho=H, Wo=W, taps=ops.fwd_taps(k, pad))
    cands = _conv_split_candidates(probe)
    assert len(cands) >= 2, cands
    tol = 1.2e-2 if dtype == torch.bfloat16 else 2e-4
    n = B * H * W
    # a busy neighbour: copies on a second stream while the split launches run (uneven arrival order of the slices)
    side = torch.cuda.Stream()
    junk_a, junk_b = torch.empty(64 << 20, dtype=torch.uint8, device="cuda"), torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    ran = 0
    for tune, S in cands:
        with torch.cuda.stream(side):
            for _ in range(8):
                junk_b.copy_(junk_a)
        first = None
        for rep in range(3):
            stats = torch.zeros(4, 2, Cout, dtype=torch.float64, device="cuda")
            keep = []
            try:
                y = ops.conv2d_fwd(xd, wp, k, 1, pad, Cout, stats=stats, stats_slots=4, tune=tune | (1 << 23), splitk=S, keep=keep)
            except L.DykError:
                y = None                                  # (large-tile shape that does not fit this map / K too short for S slices)
                break
            assert int(keep[0][1].abs().max()) == 0, "tile counters not re-armed"
            if first is None:
                first = (y.clone(), stats.clone())
            else:
                assert torch.equal(y, first[0]) and torch.equal(stats, first[1]), (hex(tune), S, "not reproducible")
        if y is None:
            continue
        ran += 1
        err = (ops.to_nchw(y).cpu().float() - y_ref).abs().max().item()
        assert err <= tol * max(1.0, y_ref.abs().max().item()), (hex(tune), S, err)
        st = stats.sum(0).cpu()
        assert torch.allclose(st[0], y_ref.double().sum((0, 2, 3)), rtol=1e-3, atol=2e-2 * n ** 0.5), (hex(tune), S)
        assert torch.allclose(st[1], (y_ref.double() ** 2).sum((0, 2, 3)), rtol=2e-3, atol=1e-2), (hex(tune), S)
        z = ops.conv2d_fwd(xd, wp, k, 1, pad, Cout, act="leaky", scale=scale.cuda(), shift=shift.cuda(), res=rd, tune=tune, splitk=S)
        err = (ops.to_nchw(z).cpu().float() - z_ref).abs().max().item()
        assert err <= 1.3 * tol * max(1.0, z_ref.abs().max().item()), (hex(tune), S, err)
    assert ran >= max(2, len(cands) // 3), (ran, len(cands))
    torch.cuda.synchronize()
    # ---- data gradient of the same conv into a BatchNorm + Mish block (fused reduce), split the same ways
    dy = rnd(B, Cout, H, W)
    u = rnd(B, Cin, H, W)
    gadd = rnd(B, Cin, H, W)
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    mean, var = u.mean((0, 2, 3)), u.var((0, 2, 3), unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    sc2, sh2 = gamma * rstd, beta - mean * gamma * rstd
    xhat = (u - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    dz = torch.nn.grad.conv2d_input((B, Cin, H, W), w, dy, padding=pad)

    def cast(t):
        return t.bfloat16().float() if dtype == torch.bfloat16 else t

    def bn_ref(dzv):
        t = (u * sc2.view(1, -1, 1, 1) + sh2.view(1, -1, 1, 1)).requires_grad_(True)
        F.mish(t).backward(dzv)
        return t.grad, t.grad.double().sum((0, 2, 3)), (t.grad.double() * xhat.double()).sum((0, 2, 3))
    da_ref, s1_ref, s2_ref = bn_ref(cast(dz))
    dzc = cast(dz + gadd)
    _, s1c, s2c = bn_ref(dzc)
    dyd, ud, addd = (ops.to_nhwc(t.cuda(), dtype) for t in (dy, u, gadd))
    wpt = ops.pack_weight(w.cuda(), dtype, transposed=True)
    out = torch.empty((B, H, W, Cin), dtype=dtype, device="cuda")
    red = torch.zeros(4, 2, Cin, dtype=torch.float64, device="cuda")
    vec = [t.cuda().contiguous() for t in (sc2, sh2, mean, rstd)]
    (py, px, Hg, Wg, taps), = ops.dgrad_classes(k, pad, 1, H, W)
    d = ops.make_conv_desc(dyd, wpt, out, Hi=H, Wi=W, Cin=Cout, Cout=Cin, Hg=Hg, Wg=Wg, Ho=H, Wo=W, taps=taps, act="mish")
    d.res, d.ldr = ud.data_ptr(), Cin
    d.scale, d.shift, d.aux0, d.aux1 = (t.data_ptr() for t in vec)
    d.stats, d.stats_slots, d.add = red.data_ptr(), 4, addd.data_ptr()
    ran_b = 0
    bcands = _conv_split_candidates(d)
    for tune, S in bcands:
        for flags, ref, r1, r2 in ((L.EPI_BNBWD, da_ref, s1_ref, s2_ref), (L.EPI_BNBWD | L.EPI_ADDEND, dzc, s1c, s2c)):
            d.flags, d.tune = flags, tune | (1 << 23)
            scratch = ops.attach_splitk(d, S, out.device)
            red.zero_()
            out.zero_()
            rc = L.load().dyk_conv_igemm(ctypes.byref(d), None)
            if rc == -3:
                continue
            assert rc == 0, (hex(tune), S, rc)
            ran_b += 1
            err = (ops.to_nchw(out).cpu().float() - ref).abs().max().item()
            assert err <= 1.5 * tol * max(1.0, ref.abs().max().item()), (hex(tune), S, flags, err)
            st = red.sum(0).cpu()
            assert torch.allclose(st[0], r1, rtol=2e-3, atol=5e-2 * n ** 0.5), (hex(tune), S, flags)
            assert torch.allclose(st[1], r2, rtol=2e-3, atol=5e-2 * n ** 0.5), (hex(tune), S, flags)
            assert int(scratch[1].abs().max()) == 0
    assert ran_b >= (8 if Cout >= 256 else min(2, len(bcands))), (ran_b, len(bcands))      # (K = Cout of the forward: short K, few ways to split)


@pytest.mark.parametrize("case", [(16, 512, 512, 16, 20, 3), (16, 1024, 512, 16, 20, 1), (16, 256, 256, 32, 40, 3)])
def test_split_k_conv_at_baseline_size(case):
    """the deep-stage layers of the target cfg at BASELINE size (batch 16): every split-K candidate the tuner would time,
    correct against torch CPU fp32 and bit-identical between two launches"""
    from dyk import lib as L
    from dyk import ops
    from dyk.plan import _conv_split_candidates
    B, Cin, Cout, H, W, k = case
    pad = k // 2
    g = torch.Generator().manual_seed(32)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    y_ref = F.conv2d(x, w, padding=pad)
    xd = ops.to_nhwc(x.cuda(), torch.bfloat16)
    wp = ops.pack_weight(w.cuda(), torch.bfloat16)
    probe = ops.make_conv_desc(xd, wp, xd, Hi=H, Wi=W, Cin=Cin, Cout=Cout, Hg=H, Wg=W, Ho=H, Wo=W, taps=ops.fwd_taps(k, pad))
    cands = _conv_split_candidates(probe)
    assert cands
    ran = 0
    for tune, S in cands:
        stats = torch.zeros(4, 2, Cout, dtype=torch.float64, device="cuda")
        try:
            y = ops.conv2d_fwd(xd, wp, k, 1, pad, Cout, stats=stats, stats_slots=4, tune=tune | (1 << 23), splitk=S)
        except L.DykError:
            continue
        ran += 1
        stats2 = torch.zeros_like(stats)
        y2 = ops.conv2d_fwd(xd, wp, k, 1, pad, Cout, stats=stats2, stats_slots=4, tune=tune | (1 << 23), splitk=S)
        assert torch.equal(y, y2) and torch.equal(stats, stats2), (hex(tune), S)
        err = (ops.to_nchw(y).cpu() - y_ref).abs().max().item()
        assert err <= 1.2e-2 * max(1.0, y_ref.abs().max().item()), (hex(tune), S, err)
    assert ran >= 4, ran


PW_CASES = [  # (B, Cin, Cout, H, W): 1x1 conv Cin -> Cout; channel counts of the target and the MobileNet cfgs, ragged pixel counts
    (2, 128, 128, 16, 20), (1, 256, 256, 13, 17), (3, 64, 32, 9, 25), (2, 72, 24, 12, 20), (1, 16, 64, 31, 33),
    (2, 256, 128, 8, 10), (1, 120, 40, 16, 20), (2, 64, 64, 16, 24), (1, 256, 512, 8, 20), (2, 96, 16, 10, 12),
]


@pytest.mark.parametrize("case", PW_CASES, ids=lambda c: "b%d_c%d_%d_%dx%d" % c)
def test_pointwise_conv_kernels(case):
    """csrc/conv_pw_kernel.h (round 5): the persistent resident-weight 1x1 kernels, every ring depth and pixel tile, against
    torch CPU fp32 on the same bf16-rounded operands -- forward with BatchNorm statistics, plain data gradient, data gradient
    with the fused BatchNorm-backward reduce (plain and residual chain, Mish / leaky / hard-swish) -- and BIT-identical raw
    outputs to the generic implicit-GEMM tile (same K walk, same MFMA).  Tune bit 23: no fallback, so every configuration
    counted really ran the pointwise kernel.  Channel counts that are not multiples of 32 run as the plan runs them: tight
    activation rows, K padded in the weight pack only (the K tail of a pixel meets zero weights)."""
    import ctypes
    from dyk import lib as L
    from dyk import ops
    B, Cin, Cout, H, W = case
    Kp = -(-Cin // 32) * 32
    g = torch.Generator().manual_seed(13)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).bfloat16().float()
    y_ref = F.conv2d(x, w)
    n = B * H * W

    def nhwc_tight(t, C):
        """channels-last rows of exactly C (multiple of 8) elements + spare zeros behind the tensor (the arena's K tail pad)"""
        flat = torch.zeros(n * C + 256, dtype=torch.bfloat16, device="cuda")
        v = flat[:n * C].view(t.shape[0], t.shape[2], t.shape[3], C)
        ops.to_nhwc(t.cuda(), torch.bfloat16, out=v)
        return flat, v
    xflat, xd = nhwc_tight(x, Cin)
    wp = ops.pack_weight(w.cuda(), torch.bfloat16, cin_pad=Kp)
    cands = [(7 << 12) | (nxs << 8) | (bn << 24) for nxs in (2, 3, 4) for bn in ((0, 1) if Cout <= 64 else (0, 2))]
    ran = 0
    y_gen = None
    for tune in [0] + cands:
        stats = torch.zeros(4, 2, Cout, dtype=torch.float64, device="cuda")
        y = torch.full((B, H, W, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
        d = ops.make_conv_desc(xd, wp, y, Hi=H, Wi=W, Cin=Kp, Cout=Cout, Hg=H, Wg=W, Ho=H, Wo=W, taps=ops.fwd_taps(1, 0), stats=stats,
                               ldx=Cin)
        d.tune, d.stats_slots = (tune | (1 << 23)) if tune else 0, 4
        rc = L.load().dyk_conv_igemm(ctypes.byref(d), None)
        if rc == -3:
            continue
        assert rc == 0, (hex(tune), rc)
        if tune == 0:
            y_gen = y.clone()
        else:
            ran += 1
            assert torch.equal(y, y_gen), (hex(tune), "raw output differs from the generic tile")
        err = (ops.to_nchw(y).cpu() - y_ref).abs().max().item()
        assert err <= 1.2e-2 * max(1.0, y_ref.abs().max().item()), (hex(tune), err)
        st = stats.sum(0).cpu()
        assert torch.allclose(st[0], y_ref.double().sum((0, 2, 3)), rtol=1e-3, atol=2e-2 * n ** 0.5), hex(tune)
        assert torch.allclose(st[1], (y_ref.double() ** 2).sum((0, 2, 3)), rtol=2e-3, atol=1e-2), hex(tune)
    assert ran >= 3, ran
    # ---- data gradient of the conv Cout -> Cin' where the roles swap: dx[n][ci] = sum_co dy[n][co] w[co][ci]
    Kb = -(-Cout // 32) * 32
    if Kb > 256 or Cin % 8:
        return
    dy = torch.randn(B, Cout, H, W, generator=g).bfloat16().float()
    u = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    gadd = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    mean, var = u.mean((0, 2, 3)), u.var((0, 2, 3), unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    sc2, sh2 = gamma * rstd, beta - mean * gamma * rstd
    xhat = (u - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    dz = torch.nn.grad.conv2d_input((B, Cin, H, W), w, dy)
    dyflat, dyd = nhwc_tight(dy, Cout)
    uflat, ud = nhwc_tight(u, Cin)
    aflat, addd = nhwc_tight(gadd, Cin)
    wpt = ops.pack_weight(w.cuda(), torch.bfloat16, transposed=True, cout_pad=Kb)
    vec = [t.cuda().contiguous() for t in (sc2, sh2, mean, rstd)]
    (py, px, Hg, Wg, taps), = ops.dgrad_classes(1, 0, 1, H, W)
    ran_b = 0
    for actname, actfn in (("mish", F.mish), ("leaky", lambda t: F.leaky_relu(t, 0.1)), ("hard-swish", F.hardswish)):
        def bn_ref(dzv):
            t = (u * sc2.view(1, -1, 1, 1) + sh2.view(1, -1, 1, 1)).requires_grad_(True)
            actfn(t).backward(dzv)
            return t.grad, t.grad.double().sum((0, 2, 3)), (t.grad.double() * xhat.double()).sum((0, 2, 3))
        da_ref, s1_ref, s2_ref = bn_ref(dz)
        dzc = (dz + gadd).bfloat16().float()
        _, s1c, s2c = bn_ref(dzc)
        for tune in [(7 << 12) | (nxs << 8) | (bn << 24) for nxs in (2, 3) for bn in (0, 1, 2)]:
            for flags, ref, r1, r2 in ((0, dz, None, None), (L.EPI_BNBWD, da_ref, s1_ref, s2_ref), (L.EPI_BNBWD | L.EPI_ADDEND, dzc, s1c, s2c)):
                if flags == 0 and actname != "mish":
                    continue
                out = torch.full((B, H, W, Cin), float("nan"), dtype=torch.bfloat16, device="cuda")
                red = torch.zeros(4, 2, Cin, dtype=torch.float64, device="cuda")
                d = ops.make_conv_desc(dyd, wpt, out, Hi=H, Wi=W, Cin=Kb, Cout=Cin, Hg=Hg, Wg=Wg, Ho=H, Wo=W, taps=taps, act=actname, ldx=Cout)
                if flags:
                    d.res, d.ldr = ud.data_ptr(), Cin
                    d.scale, d.shift, d.aux0, d.aux1 = (t.data_ptr() for t in vec)
                    d.stats, d.stats_slots, d.add = red.data_ptr(), 4, addd.data_ptr()
                d.flags, d.tune = flags, tune | (1 << 23)
                rc = L.load().dyk_conv_igemm(ctypes.byref(d), None)
                if rc == -3:
                    continue
                assert rc == 0, (hex(tune), flags, rc)
                ran_b += 1
                err = (ops.to_nchw(out).cpu() - ref).abs().max().item()
                assert err <= 1.5e-2 * max(1.0, ref.abs().max().item()), (hex(tune), flags, actname, err)
                if flags:
                    st = red.sum(0).cpu()
                    assert torch.allclose(st[0], r1, rtol=2e-3, atol=5e-2 * n ** 0.5), (hex(tune), flags, actname)
                    assert torch.allclose(st[1], r2, rtol=2e-3, atol=5e-2 * n ** 0.5), (hex(tune), flags, actname)
    assert ran_b >= 9, ran_b
