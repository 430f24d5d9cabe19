"""Box helpers of build_utils/utils.py against fixtures the REFERENCE generated
(tests/golden/make_golden_round2.py boxes): host form on CPU, HIP kernels on the GPU -- bit-exact both ways."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN

sys.path.insert(0, GOLDEN)
import make_golden_round2 as R2  # noqa: E402

GOLD = np.load(os.path.join(GOLDEN, "boxes.npz"))


def _run(dev):
    from build_utils import utils as U
    x = {k: v.to(dev) for k, v in R2.box_inputs().items()}
    out = {}
    out["xywh2xyxy"] = U.xywh2xyxy(x["xywh"].clone())
    out["xyxy2xywh"] = U.xyxy2xywh(x["xyxy"].clone())
    b = x["wild"][:, :4].clone()
    U.clip_coords(b, (512, 640))
    out["clip"] = b
    for ci, (s1, s0, rp) in enumerate(R2.SCALE_CASES):
        c = x["wild"].clone()
        r = U.scale_coords(s1, c, s0, rp)
        assert r is c
        out["scale%d" % ci] = c
    return x, out


def test_box_helpers_host_match_reference_bitwise():
    from build_utils import utils as U
    x, out = _run("cpu")
    for k, v in out.items():
        assert np.array_equal(v.numpy(), GOLD[k]), k
    assert np.array_equal(U.xywh2xyxy(x["xywh"].numpy().copy()), GOLD["xywh2xyxy_np"])
    for fmt in (True, False):
        b1 = (x["xyxy"] if fmt else x["xywh"]).t().clone()
        b2 = (x["xyxy2"] if fmt else x["xywh2"]).clone()
        for mode in ("IoU", "GIoU", "DIoU", "CIoU"):
            kw = {} if mode == "IoU" else {mode: True}
            got = U.bbox_iou(b1, b2, x1y1x2y2=fmt, **kw).numpy()
            assert np.array_equal(got, GOLD["bbox_iou|%d|%s" % (int(fmt), mode)], equal_nan=True), (fmt, mode)
    assert np.array_equal(U.box_iou(x["xyxy"][:33], x["xyxy2"][:57]).numpy(), GOLD["box_iou"], equal_nan=True)
    assert np.array_equal(U.wh_iou(x["wh1"], x["wh2"]).numpy(), GOLD["wh_iou"])


def test_focal_loss_and_smooth_bce():
    from build_utils import utils as U
    g = torch.Generator().manual_seed(2)
    pred, true = torch.randn(64, generator=g) * 3, (torch.rand(64, generator=g) > 0.7).float()
    for red in ("mean", "sum", "none"):
        fl = U.FocalLoss(torch.nn.BCEWithLogitsLoss(reduction=red), gamma=1.5, alpha=0.25)
        bce = torch.nn.functional.binary_cross_entropy_with_logits(pred, true, reduction="none")
        p = torch.sigmoid(pred)
        p_t = true * p + (1 - true) * (1 - p)
        ref = bce * ((true * 0.25 + (1 - true) * 0.75) * (1.0 - p_t) ** 1.5)
        ref = {"mean": ref.mean(), "sum": ref.sum(), "none": ref}[red]
        assert torch.equal(fl(pred, true), ref)
    assert U.smooth_BCE(0.1) == (0.95, 0.05)


@pytest.mark.gpu
def test_box_helpers_hip_match_reference_bitwise():
    _, out = _run("cuda")
    for k, v in out.items():
        assert v.is_cuda
        assert np.array_equal(v.cpu().numpy(), GOLD[k]), k


@pytest.mark.gpu
def test_scale_coords_on_detection_rows_in_place():
    """evaluate.py:82 hands the first four columns of the NMS output [n,6] (a strided view) to scale_coords"""
    from build_utils import utils as U
    x = R2.box_inputs()
    det = x["wild"].clone().cuda()
    view = det[:, :4]
    s1, s0, rp = R2.SCALE_CASES[4]
    U.scale_coords(s1, view, s0, rp)
    assert np.array_equal(det.cpu().numpy()[:, :4], GOLD["scale4"][:, :4])
    assert np.array_equal(det.cpu().numpy()[:, 4:], x["wild"].numpy()[:, 4:])          # score / class untouched
    empty = torch.zeros((0, 6), device="cuda")
    assert U.scale_coords((128, 160), empty[:, :4], (512, 640)).shape == (0, 4)


@pytest.mark.gpu
def test_box_helpers_accept_other_dtypes_and_one_row_views():
    """the reference's tensor expressions take any dtype and any view; the HIP entry points compute in fp32
    on a row-contiguous copy and hand back the caller's dtype / write through the caller's view."""
    from build_utils import utils as U
    x = R2.box_inputs()["wild"][:, :4].clone()
    want = U.xywh2xyxy(x)                                    # host form (bit-exact to the reference by boxes.npz)
    for dt, tol in ((torch.float64, 0.0), (torch.float16, 2e-3)):
        got = U.xywh2xyxy(x.to(dt).cuda())
        assert got.dtype == dt and got.is_cuda
        ref = want.to(dt).double() if dt == torch.float64 else U.xywh2xyxy(x.to(dt).float()).double()
        assert float((got.double().cpu() - ref).abs().max()) <= tol * float(ref.abs().max()) + (0 if tol else 0.0)
    # one row whose columns are NOT unit-stride (a transposed [4,1] tensor): read through the strides, not as contiguous
    col = torch.tensor([[10.0], [20.0], [4.0], [6.0]]).cuda()         # [4,1]; .t() -> [1,4] with stride (1,1)? no: (1, 1)
    wide = torch.arange(8.0).reshape(4, 2).cuda()                      # take column 1 of a [4,2] tensor as a [1,4] row
    row = wide[:, 1].unsqueeze(0)                                      # shape [1,4], stride (1, 2): values 1,3,5,7
    assert row.stride(1) == 2
    got = U.xywh2xyxy(row)
    assert torch.equal(got.cpu(), U.xywh2xyxy(torch.tensor([[1.0, 3.0, 5.0, 7.0]])))
    boxes = wide[:, 1].unsqueeze(0)
    U.clip_coords(boxes, (4, 4))
    assert wide[:, 1].tolist() == [1.0, 3.0, 4.0, 4.0] and wide[:, 0].tolist() == [0.0, 2.0, 4.0, 6.0]
    # no detections on a one-rank job: the empty result stays on the device
    from dyk.ddp import gather_detections
    assert gather_detections([None], [0]).shape == (0, 7)
    e = gather_detections([torch.zeros((0, 6), device="cuda")], [0])
    assert e.shape == (0, 7) and e.is_cuda
