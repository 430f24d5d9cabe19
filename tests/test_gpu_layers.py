"""Stand-alone calls of the operator surface (build_utils.layers classes, models.YOLOLayer, ConvBlock) on CUDA
NCHW tensors against the torch CPU modules the reference instantiates (layers.py:32-234, models.py:28-64,158-258)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, ref, tol=2e-5):
    err = (got.cpu() - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), err


def test_routing_ops():
    from build_utils.layers import FeatureConcat, WeightedFeatureFusion
    g = torch.Generator().manual_seed(1)
    outs = [torch.randn(2, c, 6, 8, generator=g) for c in (32, 64, 32)]
    cu = [o.cuda() for o in outs]
    assert FeatureConcat([1]).forward(None, cu) is cu[1]                       # single source: alias, no copy
    _close(FeatureConcat([0, 2, 1])(None, cu), torch.cat([outs[0], outs[2], outs[1]], 1))
    x = torch.randn(2, 32, 6, 8, generator=g)
    _close(WeightedFeatureFusion([0])(x.cuda(), cu), x + outs[0])
    wf = WeightedFeatureFusion([2], weight=True)
    with torch.no_grad():
        wf.w.copy_(torch.tensor([0.4, -0.9]))
    w = torch.sigmoid(wf.w.detach()) * (2 / 2)
    _close(wf.cuda()(x.cuda(), cu), x * w[0] + outs[2] * w[1])


def test_squeeze_excitation_and_make_divisible():
    from build_utils.layers import SqueezeExcitation, make_divisible
    assert make_divisible(256 // 4, 8) == 64 and make_divisible(72 // 4, 8) == 24
    torch.manual_seed(2)
    se = SqueezeExcitation(64, 4)
    x = torch.randn(2, 64, 6, 8)
    s = F.hardsigmoid(se.fc2(F.relu(se.fc1(F.adaptive_avg_pool2d(x, 1)))))
    ref = (s * x).detach()
    _close(se.cuda()(x.cuda()), ref)


@pytest.mark.parametrize("act", ["mish", "leaky", "linear"])
def test_conv_block_eval_and_train(act):
    from build_utils.layers import ConvBlock, make_activation
    torch.manual_seed(3)
    blk = ConvBlock()
    blk.add_module("Conv2d", nn.Conv2d(3, 32, 3, 1, 1, bias=False))          # Cin = 3: padded to 32 internally
    blk.add_module("BatchNorm2d", nn.BatchNorm2d(32))
    a = make_activation(act)
    if a is not None:
        blk.add_module("activation", a)
        blk.act_name = act
    with torch.no_grad():
        blk.BatchNorm2d.running_mean.normal_(0, 0.2)
        blk.BatchNorm2d.running_var.uniform_(0.5, 1.5)
        blk.BatchNorm2d.weight.uniform_(0.5, 1.5)
        blk.BatchNorm2d.bias.normal_(0, 0.2)
    ref_blk = nn.Sequential(nn.Conv2d(3, 32, 3, 1, 1, bias=False), nn.BatchNorm2d(32), *( [make_activation(act)] if a is not None else []))
    ref_blk[0].load_state_dict(blk.Conv2d.state_dict())
    ref_blk[1].load_state_dict(blk.BatchNorm2d.state_dict())
    x = torch.randn(2, 3, 16, 20)
    ref_blk.eval()
    ref = ref_blk(x).detach()
    blk = blk.cuda().eval()
    _close(blk(x.cuda()), ref, 5e-5)
    ref_blk.train()
    ref_t = ref_blk(x).detach()
    blk.train()
    _close(blk(x.cuda()), ref_t, 5e-5)
    _close(blk.BatchNorm2d.running_mean, ref_blk[1].running_mean, 1e-5)
    _close(blk.BatchNorm2d.running_var, ref_blk[1].running_var, 1e-5)
    assert int(blk.BatchNorm2d.num_batches_tracked) == 1


def test_yolo_layer_standalone():
    import numpy as np
    from models import YOLOLayer
    for bf in ("yolov3", "yolov4"):
        lay = YOLOLayer(np.array([[16., 32.], [18., 42.], [22., 44.]]), 1, (128, 160), 8, bf)
        p = torch.randn(2, 18, 16, 20)
        lay.train()
        out = lay.cuda()(p.cuda())
        ref = p.view(2, 3, 6, 16, 20).permute(0, 1, 3, 4, 2).contiguous()
        assert torch.equal(out.cpu(), ref)
        lay.eval()
        io, raw = lay(p.cuda())
        yv, xv = torch.meshgrid(torch.arange(16), torch.arange(20), indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, 1, 16, 20, 2).float()
        awh = (torch.tensor([[16., 32.], [18., 42.], [22., 44.]]) / 8).view(1, 3, 1, 1, 2)
        if bf == "yolov4":
            s = ref.sigmoid()
            want = torch.cat(((s[..., :2] * 2 - 0.5 + grid) * 8, ((s[..., 2:4] * 2) ** 2 * awh) * 8, s[..., 4:]), -1)
        else:
            want = torch.cat(((ref[..., :2].sigmoid() + grid) * 8, (ref[..., 2:4].exp() * awh) * 8, ref[..., 4:].sigmoid()), -1)
        assert io.shape == (2, 3 * 16 * 20, 6)
        err = ((io.cpu() - want.view(2, -1, 6)).abs() / want.view(2, -1, 6).abs().clamp(min=1)).max().item()
        assert err < 2e-6
        assert torch.equal(raw.cpu(), ref)


@pytest.mark.parametrize("stride", [1, 2])
def test_depthwise_separable_block(stride):
    """DepthwiseSeparableConv2d (reference layers.py:218-231), eval and train, C = 40 (rows padded to 64)"""
    from build_utils.layers import DepthwiseSeparableConv2d
    torch.manual_seed(5)
    blk = DepthwiseSeparableConv2d(40, 72, 3, stride)
    with torch.no_grad():
        for j in (1, 4):
            blk.conv[j].running_mean.normal_(0, 0.2)
            blk.conv[j].running_var.uniform_(0.5, 1.5)
            blk.conv[j].weight.uniform_(0.5, 1.5)
            blk.conv[j].bias.normal_(0, 0.2)
    ref_blk = nn.Sequential(nn.Conv2d(40, 40, 3, stride, 1, groups=40, bias=False), nn.BatchNorm2d(40), nn.ReLU6(),
                            nn.Conv2d(40, 72, 1, 1, 0, bias=False), nn.BatchNorm2d(72), nn.ReLU6())
    ref_blk.load_state_dict(blk.conv.state_dict())
    x = torch.randn(2, 40, 16, 20)
    ref_blk.eval()
    ref = ref_blk(x).detach()
    blk = blk.cuda().eval()
    _close(blk(x.cuda()), ref, 5e-5)
    ref_blk.train()
    ref_t = ref_blk(x).detach()
    blk.train()
    _close(blk(x.cuda()), ref_t, 5e-5)
    for j in (1, 4):
        _close(blk.conv[j].running_mean, ref_blk[j].running_mean, 1e-5)
        _close(blk.conv[j].running_var, ref_blk[j].running_var, 1e-5)


def test_compat_blocks_resblock_and_seinception_fusion():
    """ResBlock / SEInceptionFusion (reference layers.py:125-215; created by no cfg section) run on the HIP operator
    surface and agree with the same blocks built from torch modules; MixConv2d keeps the reference's channel split."""
    from build_utils.layers import MixConv2d, ResBlock, SEInceptionFusion
    torch.manual_seed(9)
    x = torch.randn(2, 32, 12, 20)

    def ref_cba(cba, t):                       # ConvBnActivation on the CPU, eval mode
        y = F.conv2d(t, cba.conv[0].weight, None, 1, cba.conv[0].padding)
        y = F.batch_norm(y, cba.conv[1].running_mean, cba.conv[1].running_var, cba.conv[1].weight, cba.conv[1].bias, False, 0.1, 1e-5)
        return F.mish(y) if cba.act_name == "mish" else F.leaky_relu(y, 0.1)

    rb = ResBlock(32, 16, 32, block_nums=2).eval()
    with torch.no_grad():
        for m in rb.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
        ref = x
        for pair in rb.module_list:
            ref = ref + ref_cba(pair[1], ref_cba(pair[0], ref))
        _close(rb.cuda()(x.cuda()), ref, 1e-4)
    fus = SEInceptionFusion(64, 64, [0, 1], inception=True, icp_param_list=(16, 24, 24, 12, 12, 12), tmse=True).eval()
    outs = [torch.randn(2, 32, 12, 20), torch.randn(2, 32, 12, 20)]
    with torch.no_grad():
        y = fus.cuda()(None, [o.cuda() for o in outs])
    assert y.shape == (2, 64, 12, 20) and bool(torch.isfinite(y).all())
    assert [c.out_channels for c in MixConv2d(32, 64).m] == [41, 15, 8]
    assert [c.out_channels for c in MixConv2d(32, 64, method="equal_ch").m] == [22, 21, 21]
    with pytest.raises(NotImplementedError):
        MixConv2d(32, 64)(x)


def test_cast_pad_table_matches_per_pack_casts():
    """dyk_cast_pad_table (round 5): every K-padded weight pack and the stems' transposes in one launch -- against
    dyk_cast_pad_rows / torch on the same data, ragged element counts included."""
    import ctypes
    from dyk import lib as L
    g = torch.Generator().manual_seed(3)
    shapes = [(27, 16, 32, 0), (9 * 40, 24, 32, 0), (72, 120, 128, 0), (16, 27, 27, 1), (5, 2049, 2080, 0), (333, 7, 7, 1)]
    srcs = [torch.randn(r, c, generator=g).cuda() for r, c, _, _ in shapes]
    dsts = [torch.full((c, r), 7.0, device="cuda") if tr else torch.full((r, cp), 7.0, device="cuda").bfloat16() for r, c, cp, tr in shapes]
    arr = (L.DykPadEntry * len(shapes))()
    blocks = 0
    for i, ((r, c, cp, tr), s, d) in enumerate(zip(shapes, srcs, dsts)):
        arr[i].src, arr[i].dst, arr[i].rows, arr[i].cols, arr[i].cpad, arr[i].blk_begin, arr[i].transpose_f32 = s.data_ptr(), d.data_ptr(), r, c, cp, blocks, tr
        blocks += (r * (c if tr else cp) + 2047) // 2048
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    L.check(L.load().dyk_cast_pad_table(tab.data_ptr(), len(shapes), blocks, L.DYK_BF16, None), "dyk_cast_pad_table")
    torch.cuda.synchronize()
    for (r, c, cp, tr), s, d in zip(shapes, srcs, dsts):
        if tr:
            assert torch.equal(d, s.t().contiguous())
        else:
            ref = torch.zeros(r, cp, device="cuda")
            ref[:, :c] = s
            assert torch.equal(d, ref.bfloat16())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_transpose_taps_table(dtype):
    """dyk_transpose_taps (the [tap][Cin][Cout] packs the data gradients read, rebuilt from the fp32 masters after every
    optimizer step): a table of ragged entries -- several taps, extents off the 32 x 32 tile, zero-filled row tails, more tiles
    than one workgroup's run and entries that end inside a run -- against torch."""
    import ctypes
    from dyk import lib as L
    g = torch.Generator().manual_seed(11)
    shapes = [(9, 64, 32, 0), (1, 33, 70, 64), (9, 255, 40, 256), (1, 8, 8, 0), (25, 16, 16, 32), (1, 300, 513, 0), (9, 32, 3, 32)]
    src = torch.randn(sum(t * r * c for t, r, c, _ in shapes) + 5, generator=g).cuda()
    arr = (L.DykTransposeEntry * len(shapes))()
    so, do, tiles = 5, 3 * 0, 0
    for i, (t, r, c, ld) in enumerate(shapes):
        arr[i].src_off, arr[i].dst_off, arr[i].taps, arr[i].rows, arr[i].cols, arr[i].dst_ld, arr[i].tile_begin = so, do, t, r, c, ld, tiles
        so += t * r * c
        do += t * (ld or r) * c
        tiles += t * ((r + 31) // 32) * ((c + 31) // 32)
    dst = torch.full((do,), 7.0, device="cuda").to(dtype)
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    code = L.DYK_BF16 if dtype == torch.bfloat16 else L.DYK_F32
    L.check(L.load().dyk_transpose_taps(src.data_ptr(), dst.data_ptr(), tab.data_ptr(), len(shapes), tiles, code, None), "dyk_transpose_taps")
    torch.cuda.synchronize()
    for i, (t, r, c, ld) in enumerate(shapes):
        s = src[arr[i].src_off:arr[i].src_off + t * r * c].view(t, r, c)
        ldd = ld or r
        d = dst[arr[i].dst_off:arr[i].dst_off + t * ldd * c].view(t, c, ldd)
        assert torch.equal(d[:, :, :r], s.transpose(1, 2).to(dtype)), i
        if ldd > r:                              # the tail of a padded row is zero-filled up to the next multiple of 32 it covers
            assert float(d[:, :, r:min(ldd, (r + 31) // 32 * 32)].float().abs().max()) == 0.0, i
