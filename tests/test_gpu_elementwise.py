"""GPU parity of the HBM-bound kernels (BN/activation fwd+bwd, axpby, upsample, max-pool, SE, head
permute, patch gather, decode) against torch CPU fp32 (the arithmetic the oracle uses)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {"leaky": lambda t: F.leaky_relu(t, 0.1), "mish": F.mish, "relu": F.relu, "relu6": F.relu6,
        "hard-swish": F.hardswish, "hard-sigmoid": F.hardsigmoid, "linear": lambda t: t}


def _tol(dtype):
    return 3e-5 if dtype == torch.float32 else 2e-2


def _close(got, ref, tol, what=""):
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), "%s max err %g (ref max %g)" % (what, err, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["mish", "leaky", "relu6", "hard-swish", "linear"])
def test_bn_act_train_fwd_bwd(dtype, act):
    """conv-output statistics -> finalize -> BN+act forward, then the three backward kernels, vs
    torch batch_norm(training=True) + activation under autograd."""
    from dyk import ops
    B, C, H, W = 3, 64, 10, 12
    g = torch.Generator().manual_seed(3)
    y = (torch.randn(B, C, H, W, generator=g) * 2 + 0.5)
    dz = torch.randn(B, C, H, W, generator=g)
    res = torch.randn(B, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        y, dz, res = y.bfloat16().float(), dz.bfloat16().float(), res.bfloat16().float()
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g).requires_grad_(True)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    yr = y.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z_ref = ACTS[act](F.batch_norm(yr, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)) + res
    z_ref.backward(dz)
    tol = _tol(dtype)

    yd = ops.to_nhwc(y.cuda(), dtype)
    n = B * H * W
    stats = torch.cat([y.double().sum((0, 2, 3)), (y.double() ** 2).sum((0, 2, 3))]).cuda()
    rmd, rvd = rm.cuda(), rv.cuda()
    scale, shift, mean, rstd = ops.bn_finalize(stats, n, gamma.detach().cuda(), beta.detach().cuda(), rmd, rvd)
    assert stats.abs().max().item() == 0.0                       # finalize re-arms the accumulator
    _close(rmd.cpu(), rm_ref, 1e-5, "running_mean")
    _close(rvd.cpu(), rv_ref, 1e-5, "running_var")
    resd = ops.to_nhwc(res.cuda(), dtype)
    z = torch.empty_like(yd)
    ops.call("dyk_bn_act_fwd", ops.ew_desc(a=yd, b=resd, out=z, act=act, p0=scale, p1=shift))
    _close(ops.to_nchw(z).cpu(), z_ref.detach(), tol, "fwd")
    # backward
    dzd = ops.to_nhwc(dz.cuda(), dtype)
    slots = 4                                                    # replicated reduction buffers
    red = torch.zeros(slots * 2 * C, dtype=torch.float64, device="cuda")
    rd = ops.ew_desc(a=dzd, b=yd, act=act, p0=scale, p1=shift, p2=mean, p3=rstd, red=red)
    rd.slots = slots
    ops.call("dyk_bn_act_bwd_reduce", rd)
    dgamma, dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    from dyk.lib import check, load
    check(load().dyk_bn_bwd_params(red.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), C, slots, None))
    _close(dgamma.cpu(), gamma.grad, 10 * tol, "dgamma")
    _close(dbeta.cpu(), beta.grad, 10 * tol, "dbeta")
    dy = torch.empty_like(dzd)
    ops.call("dyk_bn_act_bwd_apply", ops.ew_desc(a=dzd, b=yd, out=dy, act=act, p0=scale, p1=shift, p2=mean, p3=rstd, red=red))
    _close(ops.to_nchw(dy).cpu(), yr.grad, tol, "dy")
    # in-place form used by the plan (out aliases a)
    ops.call("dyk_bn_act_bwd_apply", ops.ew_desc(a=dzd, b=yd, out=dzd, act=act, p0=scale, p1=shift, p2=mean, p3=rstd, red=red))
    _close(ops.to_nchw(dzd).cpu(), yr.grad, tol, "dy in place")
    # fused form used by the plan: apply folds the replicas itself and adds the totals to dgamma / dbeta
    dzd = ops.to_nhwc(dz.cuda(), dtype)
    red.zero_()
    ops.call("dyk_bn_act_bwd_reduce", _redesc(ops, dzd, yd, act, scale, shift, mean, rstd, red, slots))
    dgamma2, dbeta2 = torch.ones(C, device="cuda"), torch.ones(C, device="cuda")       # accumulate onto existing values
    ap = ops.ew_desc(a=dzd, b=yd, out=dzd, act=act, p0=scale, p1=shift, p2=mean, p3=rstd, red=red)
    ap.slots, ap.aux, ap.aux2 = slots, dgamma2.data_ptr(), dbeta2.data_ptr()
    ops.call("dyk_bn_act_bwd_apply", ap)
    _close(ops.to_nchw(dzd).cpu(), yr.grad, tol, "dy fused")
    _close(dgamma2.cpu() - 1, gamma.grad, 10 * tol, "dgamma fused")
    _close(dbeta2.cpu() - 1, beta.grad, 10 * tol, "dbeta fused")


def _redesc(ops, dzd, yd, act, scale, shift, mean, rstd, red, slots):
    rd = ops.ew_desc(a=dzd, b=yd, act=act, p0=scale, p1=shift, p2=mean, p3=rstd, red=red)
    rd.slots = slots
    return rd


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_axpby_dot_wfuse(dtype):
    from dyk import ops
    from dyk.lib import EW_ACCUM, check, load
    B, C, H, W = 2, 48, 6, 10
    g = torch.Generator().manual_seed(4)
    a, b = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        a, b = a.bfloat16().float(), b.bfloat16().float()
    ad, bd = ops.to_nhwc(a.cuda(), dtype), ops.to_nhwc(b.cuda(), dtype)
    tol = _tol(dtype)
    # concat copy into a channel slice, then accumulate
    cat = torch.zeros(B, H, W, 128, dtype=dtype, device="cuda")
    ops.call("dyk_axpby", ops.ew_desc(a=ad, out=cat[..., 32:80]))
    ops.call("dyk_axpby", ops.ew_desc(a=bd, out=cat[..., 32:80], flags=EW_ACCUM))
    _close(ops.to_nchw(cat[..., 32:80]).cpu(), a + b, tol, "copy+accum")
    assert cat[..., :32].abs().max().item() == 0 and cat[..., 80:].abs().max().item() == 0
    # weighted fusion: w = sigmoid(w_raw) * 2/n
    w_raw = torch.tensor([0.3, -0.7], requires_grad=True)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    w = torch.sigmoid(w_raw) * (2 / 2)
    z_ref = ar * w[0] + br * w[1]
    dz = torch.randn(B, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        dz = dz.bfloat16().float()
    z_ref.backward(dz)
    weff = torch.zeros(2, device="cuda")
    check(load().dyk_wfuse_weights(w_raw.detach().cuda().data_ptr(), weff.data_ptr(), 2, None))
    z = torch.empty_like(ad)
    ops.call("dyk_axpby", ops.ew_desc(a=ad, b=bd, out=z, p0=weff[0:1], p1=weff[1:2]))
    _close(ops.to_nchw(z).cpu(), z_ref.detach(), tol, "wfuse fwd")
    dzd = ops.to_nhwc(dz.cuda(), dtype)
    red = torch.zeros(2, dtype=torch.float64, device="cuda")
    ops.call("dyk_dot", ops.ew_desc(a=dzd, b=ad, red=red[0:1]))
    ops.call("dyk_dot", ops.ew_desc(a=dzd, b=bd, red=red[1:2]))
    dw = torch.zeros(2, device="cuda")
    wr = w_raw.detach().cuda()
    check(load().dyk_wfuse_bwd_params(wr.data_ptr(), red.data_ptr(), dw.data_ptr(), 2, None))
    _close(dw.cpu(), w_raw.grad, 20 * tol, "dw")
    da = torch.empty_like(ad)
    ops.call("dyk_axpby", ops.ew_desc(a=dzd, out=da, p0=weff[0:1]))
    _close(ops.to_nchw(da).cpu(), ar.grad, tol, "da")
    # one pass per source (how the plan runs it): dyk_dot with `out` leaves the dot product AND the scaled copy -- the same bits
    # as the two separate calls, store and accumulate forms
    for accumulate in (False, True):
        base = ops.to_nhwc(torch.randn(B, C, H, W, generator=g).cuda(), dtype)
        ref = base.clone()
        ops.call("dyk_axpby", ops.ew_desc(a=dzd, out=ref, p0=weff[1:2], flags=EW_ACCUM if accumulate else 0))
        got, red2 = base.clone(), torch.zeros(1, dtype=torch.float64, device="cuda")
        ops.call("dyk_dot", ops.ew_desc(a=dzd, b=bd, out=got, p0=weff[1:2], red=red2, flags=EW_ACCUM if accumulate else 0))
        assert torch.equal(got, ref) and torch.equal(red2, red[1:2])


@pytest.mark.parametrize("weighted", [False, True])
def test_weighted_fusion_module_with_mismatched_channels(weighted):
    """WeightedFeatureFusion.forward called as a module (inference path of build_utils/layers.py) on tensors of different channel
    counts: the reference's three branches (layers.py:78-83), restated functionally; alpha = 0 of dyk_axpby never reads `a`."""
    from build_utils.layers import WeightedFeatureFusion
    from dyk import ops
    g = torch.Generator().manual_seed(9)
    outs = [torch.randn(2, 64, 6, 10, generator=g), torch.randn(2, 32, 6, 10, generator=g), torch.randn(2, 48, 6, 10, generator=g)]
    for xi, layers in [(0, [1]), (1, [0]), (2, [0, 1]), (0, [2, 1])]:
        mod = WeightedFeatureFusion(layers, weight=weighted)
        if weighted:
            with torch.no_grad():
                mod.w.copy_(torch.tensor([0.4, -0.6, 1.1][:mod.n]))
        x = outs[xi]
        ref = x.clone()
        if weighted:
            w = torch.sigmoid(mod.w.detach()) * (2 / mod.n)
            ref = ref * w[0]
        for q, j in enumerate(layers):
            a = outs[j] * w[q + 1] if weighted else outs[j]
            n = min(ref.shape[1], a.shape[1])
            ref = torch.cat((ref[:, :n] + a[:, :n], ref[:, n:]), 1)
        with torch.no_grad():
            got = mod.cuda()(x.cuda(), [o.cuda() for o in outs])
        assert got.shape == x.shape
        _close(got.cpu(), ref, 1e-5, "fusion %d <- %s" % (xi, layers))
    nan = torch.full((2, 6, 10, 32), float("nan"), device="cuda")
    ops.call("dyk_axpby", ops.ew_desc(a=nan, out=nan, alpha=0.0))
    assert nan.abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample_maxpool(dtype):
    from dyk import ops
    from dyk.lib import EW_ACCUM
    B, C, H, W = 2, 32, 7, 9
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xd = ops.to_nhwc(x.cuda(), dtype)
    tol = _tol(dtype)
    # upsample
    xr = x.clone().requires_grad_(True)
    up_ref = F.interpolate(xr, scale_factor=2, mode="nearest")
    dup = torch.randn(up_ref.shape, generator=g)
    if dtype == torch.bfloat16:
        dup = dup.bfloat16().float()
    up_ref.backward(dup)
    up = torch.empty(B, 2 * H, 2 * W, C, dtype=dtype, device="cuda")
    ops.call("dyk_upsample2x_fwd", ops.ew_desc(a=xd, out=up, B=B, H=H, W=W))
    assert torch.equal(ops.to_nchw(up).cpu(), up_ref.detach())
    dupd = ops.to_nhwc(dup.cuda(), dtype)
    dx = torch.empty_like(xd)
    ops.call("dyk_upsample2x_bwd", ops.ew_desc(a=dupd, out=dx, C=C, B=B, H=H, W=W))
    _close(ops.to_nchw(dx).cpu(), xr.grad, tol, "upsample bwd")
    # max pool 5 / 9 / 13 (SPP), values quantised so that ties occur
    xq = (x * 2).round() / 2
    xqd = ops.to_nhwc(xq.cuda(), dtype)
    for k in (3, 5, 9, 13):
        xr = xq.clone().requires_grad_(True)
        mp_ref = F.max_pool2d(xr, k, 1, (k - 1) // 2)
        dmp = torch.randn(mp_ref.shape, generator=g)
        if dtype == torch.bfloat16:
            dmp = dmp.bfloat16().float()
        mp_ref.backward(dmp)
        mp = torch.empty_like(xqd)
        amax = torch.zeros(B * H * W * C, dtype=torch.uint8, device="cuda")
        ops.call("dyk_maxpool_fwd", ops.ew_desc(a=xqd, out=mp, B=B, H=H, W=W, k=k), amax)
        assert torch.equal(ops.to_nchw(mp).cpu(), mp_ref.detach()), "maxpool k=%d" % k
        dmpd = ops.to_nhwc(dmp.cuda(), dtype)
        dxp = torch.empty_like(xqd)
        ops.call("dyk_maxpool_bwd", ops.ew_desc(a=dmpd, out=dxp, B=B, H=H, W=W, k=k), amax)
        _close(ops.to_nchw(dxp).cpu(), xr.grad, 4 * tol, "maxpool bwd k=%d (tie rule)" % k)
    # strided pools (nn.MaxPool2d(k, stride, padding=(k-1)//2), models.py:91-94): stride travels in `slots`
    for k, st in ((2, 2), (3, 2), (5, 2), (4, 1)):
        pad = (k - 1) // 2
        xr = xq.clone().requires_grad_(True)
        mp_ref = F.max_pool2d(xr, k, st, pad)
        dmp = torch.randn(mp_ref.shape, generator=g)
        if dtype == torch.bfloat16:
            dmp = dmp.bfloat16().float()
        mp_ref.backward(dmp)
        Ho, Wo = mp_ref.shape[2:]
        mp = torch.empty((B, Ho, Wo, C), dtype=dtype, device="cuda")
        amax = torch.zeros(B * H * W * C, dtype=torch.uint8, device="cuda")
        fd = ops.ew_desc(a=xqd, out=mp, B=B, H=H, W=W, k=k)
        fd.slots = st
        ops.call("dyk_maxpool_fwd", fd, amax)
        assert torch.equal(ops.to_nchw(mp).cpu(), mp_ref.detach()), "maxpool k=%d s=%d" % (k, st)
        dmpd = ops.to_nhwc(dmp.cuda(), dtype)
        dxp = torch.empty_like(xqd)
        bd = ops.ew_desc(a=dmpd, out=dxp, B=B, H=H, W=W, k=k)
        bd.slots = st
        ops.call("dyk_maxpool_bwd", bd, amax)
        _close(ops.to_nchw(dxp).cpu(), xr.grad, 4 * tol, "maxpool bwd k=%d s=%d" % (k, st))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_spp_pools_on_the_16x20_map(dtype):
    """the LDS-plane pool kernels (stride 1, odd window) at the SPP block's own map size: values equal torch's, the
    argmax codes are torch's first maximum in scan order (ties on purpose), gradient by the tie rule"""
    from dyk import ops
    B, C, H, W = 2, 64, 16, 20
    g = torch.Generator().manual_seed(15)
    xq = (torch.randn(B, C, H, W, generator=g) * 1.5).round() / 2
    xqd = ops.to_nhwc(xq.cuda(), dtype)
    tol = _tol(dtype)
    yy, xx = torch.arange(H).view(1, 1, H, 1), torch.arange(W).view(1, 1, 1, W)
    for k in (5, 9, 13):
        pad = (k - 1) // 2
        xr = xq.clone().requires_grad_(True)
        mp_ref, ind = F.max_pool2d(xr, k, 1, pad, return_indices=True)
        dmp = torch.randn(mp_ref.shape, generator=g)
        if dtype == torch.bfloat16:
            dmp = dmp.bfloat16().float()
        mp_ref.backward(dmp)
        mp = torch.empty_like(xqd)
        amax = torch.zeros(B * H * W * C, dtype=torch.uint8, device="cuda")
        ops.call("dyk_maxpool_fwd", ops.ew_desc(a=xqd, out=mp, B=B, H=H, W=W, k=k), amax)
        assert torch.equal(ops.to_nchw(mp).cpu(), mp_ref.detach()), "maxpool k=%d" % k
        code = (ind // W - (yy - pad)) * k + (ind % W - (xx - pad))
        assert torch.equal(amax.view(B, H, W, C).permute(0, 3, 1, 2).cpu().long(), code), "argmax codes k=%d" % k
        dxp = torch.empty_like(xqd)
        ops.call("dyk_maxpool_bwd", ops.ew_desc(a=ops.to_nhwc(dmp.cuda(), dtype), out=dxp, B=B, H=H, W=W, k=k), amax)
        _close(ops.to_nchw(dxp).cpu(), xr.grad, 8 * tol, "maxpool bwd k=%d" % k)


@pytest.mark.parametrize("B,C,Cs,H,W", [(3, 64, 16, 6, 8), (5, 520, 132, 2, 3)])       # (ragged row / column blocks)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_squeeze_excitation_fwd_bwd(dtype, B, C, Cs, H, W):
    from dyk import ops
    from dyk.lib import DykSeFcDesc
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, C, H, W, generator=g)
    dz = torch.randn(B, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        x, dz = x.bfloat16().float(), dz.bfloat16().float()
    w1 = (torch.randn(Cs, C, 1, 1, generator=g) * 0.2).requires_grad_(True)
    b1 = (torch.randn(Cs, generator=g) * 0.1).requires_grad_(True)
    w2 = (torch.randn(C, Cs, 1, 1, generator=g) * 0.5).requires_grad_(True)
    b2 = (torch.randn(C, generator=g) * 0.5).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    s = F.hardsigmoid(F.conv2d(F.relu(F.conv2d(F.adaptive_avg_pool2d(xr, 1), w1, b1)), w2, b2))
    z_ref = s * xr
    z_ref.backward(dz)
    tol = _tol(dtype)
    xd, dzd = ops.to_nhwc(x.cuda(), dtype), ops.to_nhwc(dz.cuda(), dtype)
    pooled = torch.zeros(B, C, device="cuda")
    scale = torch.zeros(B, C, device="cuda")
    ops.call("dyk_se_pool", ops.ew_desc(a=xd, B=B, H=H, W=W, alpha=1.0 / (H * W)), pooled)
    prm = [t.detach().cuda().contiguous() for t in (w1, b1, w2, b2)]
    fd = DykSeFcDesc()
    fd.pooled, fd.w1, fd.b1, fd.w2, fd.b2, fd.scale = pooled.data_ptr(), prm[0].data_ptr(), prm[1].data_ptr(), prm[2].data_ptr(), prm[3].data_ptr(), scale.data_ptr()
    fd.B, fd.C, fd.Cs = B, C, Cs
    fcws = torch.zeros(B * (C + 2 * Cs), device="cuda")      # h | dt1 | t2: the forward call parks h and t2 for the backward one
    fd.ws = fcws.data_ptr()
    ops.call("dyk_se_fc_fwd", fd)
    _close(scale.cpu(), s.detach().view(B, C), 1e-5, "se scale")
    z = torch.empty_like(xd)
    ops.call("dyk_se_scale", ops.ew_desc(a=xd, out=z, p0=scale, B=B, H=H, W=W))
    _close(ops.to_nchw(z).cpu(), z_ref.detach(), tol, "se fwd")
    # backward
    dscale = torch.zeros(B, C, device="cuda")
    dpooled = torch.zeros(B, C, device="cuda")
    ops.call("dyk_se_pool", ops.ew_desc(a=dzd, b=xd, B=B, H=H, W=W, alpha=1.0), dscale)
    grads = [torch.zeros_like(t) for t in prm]
    fd.dscale, fd.dpooled = dscale.data_ptr(), dpooled.data_ptr()
    fd.dw1, fd.db1, fd.dw2, fd.db2 = (t.data_ptr() for t in grads)
    ops.call("dyk_se_fc_bwd", fd)
    for got, ref, nm in zip(grads, (w1, b1, w2, b2), ("dw1", "db1", "dw2", "db2")):
        _close(got.cpu().view(-1), ref.grad.view(-1), 20 * tol, nm)
    dx = torch.empty_like(xd)
    ops.call("dyk_se_scale", ops.ew_desc(a=dzd, out=dx, p0=scale, p1=dpooled, alpha=1.0 / (H * W), B=B, H=H, W=W))
    _close(ops.to_nchw(dx).cpu(), xr.grad, tol, "se dx")
    # the two halves as separate calls (how the plan issues them: dpooled on the chain to dx, the parameter gradients as a
    # command of their own) give the same bits as the one call
    def clone(src):
        out = type(src)()
        ctypes.memmove(ctypes.byref(out), ctypes.byref(src), ctypes.sizeof(src))
        return out
    dpooled2 = torch.zeros_like(dpooled)
    grads2 = [torch.zeros_like(t) for t in prm]
    half = clone(fd)
    half.dpooled = dpooled2.data_ptr()
    half.dw1 = half.db1 = half.dw2 = half.db2 = None
    ops.call("dyk_se_fc_bwd", half)
    assert torch.equal(dpooled2, dpooled) and all(float(t.abs().max()) == 0.0 for t in grads2)
    half = clone(fd)
    half.dpooled = None
    half.dw1, half.db1, half.dw2, half.db2 = (t.data_ptr() for t in grads2)
    ops.call("dyk_se_fc_bwd", half)
    for a, b2_ in zip(grads2, grads):
        assert torch.equal(a, b2_)
    from dyk import lib as L
    half.dw2 = None                                           # three of four gradient pointers: refused
    assert L.load().dyk_se_fc_bwd(ctypes.byref(half), None) == -1          # DYK_ERR_ARG


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_se_pool_pixel_split_matches_single_stage(dtype):
    """dyk_se_pool with the aux2 scratch splits the pixels of an image over several workgroups (partial sums folded in a
    fixed order): same values as the one-workgroup reduction up to fp32 summation order, identical from run to run"""
    from dyk import ops
    from dyk.lib import SE_POOL_SPLITS
    B, C, H, W = 4, 120, 64, 80
    g = torch.Generator().manual_seed(16)
    x = torch.randn(B, C, H, W, generator=g)
    y = torch.randn(B, C, H, W, generator=g)
    xd, yd = ops.to_nhwc(x.cuda(), dtype), ops.to_nhwc(y.cuda(), dtype)
    xq, yq = ops.to_nchw(xd).cpu().double(), ops.to_nchw(yd).cpu().double()
    for second, ref, alpha in ((None, xq.mean((2, 3)), 1.0 / (H * W)), (yd, (xq * yq).sum((2, 3)), 1.0)):
        outs = []
        for split in (False, True, True):
            pooled = torch.full((B, C), float("nan"), device="cuda")
            d = ops.ew_desc(a=xd, b=second, B=B, H=H, W=W, alpha=alpha)
            scratch = torch.full((SE_POOL_SPLITS * B * C,), float("nan"), device="cuda")
            if split:
                d.aux2 = scratch.data_ptr()
            ops.call("dyk_se_pool", d, pooled)
            outs.append(pooled.cpu())
            if split:
                assert not bool(torch.isnan(scratch[:2 * B * C]).any())      # the split path ran
        _close(outs[0], ref.float(), 2e-5 * max(1.0, float(ref.abs().max())), "se pool single")
        _close(outs[1], ref.float(), 2e-5 * max(1.0, float(ref.abs().max())), "se pool split")
        assert torch.equal(outs[1], outs[2])


def test_head_permute_patch_gather_decode():
    from dyk import ops
    from dyk.lib import DYK_BF16, DYK_F32, DykDecodeDesc, check, load
    lib = load()
    B, na, no, ny, nx = 2, 3, 6, 5, 7
    g = torch.Generator().manual_seed(8)
    y = torch.randn(B, na * no, ny, nx, generator=g)
    yd = torch.zeros(B, ny, nx, 32, device="cuda")
    yd[..., :na * no] = y.permute(0, 2, 3, 1).cuda()
    p = torch.empty(B, na, ny, nx, no, device="cuda")
    check(lib.dyk_head_permute_fwd(yd.data_ptr(), p.data_ptr(), B, ny, nx, na, no, 32, None))
    p_ref = y.view(B, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
    assert torch.equal(p.cpu(), p_ref)
    dp = torch.randn(B, na, ny, nx, no, generator=g)
    dy = torch.full((B, ny, nx, 32), 7.0, device="cuda")
    db = torch.zeros(na * no, device="cuda")
    check(lib.dyk_head_permute_bwd(dp.cuda().data_ptr(), dy.data_ptr(), db.data_ptr(), B, ny, nx, na, no, 32, DYK_F32, None))
    dy_ref = dp.permute(0, 1, 4, 2, 3).reshape(B, na * no, ny, nx).permute(0, 2, 3, 1)
    assert torch.equal(dy[..., :na * no].cpu(), dy_ref) and dy[..., na * no:].abs().max().item() == 0
    _close(db.cpu(), dy_ref.sum((0, 1, 2)), 1e-5, "head bias grad")
    # patch gather = im2col of the 3-channel stem
    x = torch.rand(B, 3, 12, 20, generator=g)
    for (k, s) in ((3, 1), (3, 2)):
        pad = k // 2
        Ho, Wo = (12 + 2 * pad - k) // s + 1, (20 + 2 * pad - k) // s + 1
        out = torch.full((B, Ho, Wo, 32), 9.0, device="cuda")
        check(lib.dyk_patch_gather(x.cuda().data_ptr(), out.data_ptr(), B, 3, 12, 20, k, s, pad, 32, 0.5, DYK_F32, None))
        cols = F.unfold(x, k, padding=pad, stride=s).view(B, 3, k * k, Ho, Wo)          # [B, c, tap, Ho, Wo]
        ref = cols.permute(0, 3, 4, 2, 1).reshape(B, Ho, Wo, k * k * 3) * 0.5            # (tap, c) order
        assert torch.equal(out[..., :k * k * 3].cpu(), ref) and out[..., k * k * 3:].abs().max().item() == 0
    # decode v3 / v4 against the formulas of models.py:238-252
    for v4 in (0, 1):
        t = torch.randn(B, na, ny, nx, no, generator=g)
        anchors = torch.tensor([[16., 32.], [18., 42.], [22., 44.]])
        stride = 8.0
        av = anchors / stride
        io = torch.zeros(B, 200, no, device="cuda")
        d = DykDecodeDesc()
        td = t.cuda()
        d.p, d.io = td.data_ptr(), io.data_ptr()
        d.B, d.na, d.ny, d.nx, d.no, d.rows_total, d.row_offset, d.v4, d.stride = B, na, ny, nx, no, 200, 50, v4, stride
        for i, v in enumerate(av.reshape(-1).tolist()):
            d.anchor_vec[i] = v
        ops.call("dyk_yolo_decode", d)
        yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()
        awh = av.view(1, na, 1, 1, 2)
        if v4:
            s_ = t.sigmoid()
            ref = torch.cat(((s_[..., :2] * 2. - 0.5 + grid) * stride, ((s_[..., 2:4] * 2) ** 2 * awh) * stride, s_[..., 4:]), -1)
        else:
            ref = torch.cat(((t[..., :2].sigmoid() + grid) * stride, (t[..., 2:4].exp() * awh) * stride, t[..., 4:].sigmoid()), -1)
        got = io[:, 50:50 + na * ny * nx].cpu()
        err = ((got - ref.view(B, -1, no)).abs() / ref.view(B, -1, no).abs().clamp(min=1.0)).max().item()
        assert err < 2e-6, err
        assert io[:, :50].abs().max().item() == 0 and io[:, 50 + na * ny * nx:].abs().max().item() == 0


@pytest.mark.parametrize("act", ["mish", "hard-swish", "relu6", "leaky"])
@pytest.mark.parametrize("B,C,H,W", [(3, 64, 6, 8), (2, 72, 33, 17), (4, 960, 4, 5)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_se_scale_backward_carries_the_batchnorm_backward_reduce(dtype, B, C, H, W, act):
    """dyk_se_scale with `red` (the squeeze-excitation backward of a block fed by conv + BatchNorm, autograd of
    layers.py:188 behind models.py:47-56): the gradient it stores is the one the plain call stores, and the replica sums equal
    what dyk_bn_act_bwd_reduce computes from that stored gradient"""
    from dyk import ops
    g = torch.Generator().manual_seed(17)
    cpad = (C + 31) // 32 * 32
    dz = ops.to_nhwc(torch.randn(B, C, H, W, generator=g).cuda(), dtype, cpad=cpad)
    y = ops.to_nhwc((2.0 * torch.randn(B, C, H, W, generator=g)).cuda(), dtype, cpad=cpad)       # raw output of the producer
    old = ops.to_nhwc(torch.randn(B, C, H, W, generator=g).cuda(), dtype, cpad=cpad)
    s = torch.rand(B, C, generator=g).cuda()
    dpooled = torch.randn(B, C, generator=g).cuda()
    vecs = torch.cat([1.0 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g),
                      0.1 * torch.randn(C, generator=g), 1.0 + 0.1 * torch.rand(C, generator=g)]).cuda()   # scale | shift | mean | rstd
    slots = 8
    for accumulate in (False, True):
        plain = old.clone()
        ops.call("dyk_se_scale", ops.ew_desc(a=dz, out=plain, C=C, p0=s, p1=dpooled, alpha=1.0 / (H * W), B=B, H=H, W=W,
                                             flags=1 if accumulate else 0))
        fused = old.clone()
        red = torch.zeros(slots * 2 * C, dtype=torch.float64, device="cuda")
        d = ops.ew_desc(a=dz, b=y, out=fused, C=C, p0=s, p1=dpooled, p2=vecs, red=red, act=act, alpha=1.0 / (H * W), B=B, H=H, W=W,
                        flags=1 if accumulate else 0)
        d.slots = slots
        ops.call("dyk_se_scale", d)
        assert torch.equal(fused, plain)
        ref = torch.zeros_like(red)
        r = ops.ew_desc(a=plain, b=y, C=C, act=act, p0=vecs[:C], p1=vecs[C:2 * C], p2=vecs[2 * C:3 * C], p3=vecs[3 * C:], red=ref)
        r.slots = slots
        ops.call("dyk_bn_act_bwd_reduce", r)
        got, want = red.view(slots, 2, C).sum(0), ref.view(slots, 2, C).sum(0)
        scale = want.abs().max().item() + 1.0
        assert (got - want).abs().max().item() <= 1e-5 * scale, (got - want).abs().max().item() / scale
