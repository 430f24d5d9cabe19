"""Oracle restatements of build_targets / compute_loss / non_max_suppression pinned against the
reference's outputs (tests/golden/targets.npz, loss.npz, nms.npz), plus hand-computed known-answer
tests for the torchvision NMS semantics (the one 'parity unpinned' boundary)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from helpers import GOLDEN

sys.path.insert(0, GOLDEN)
import cases  # noqa: E402

from oracle import loss as oloss  # noqa: E402
from oracle import nms as onms  # noqa: E402


def _anchor_vecs(cfg):
    anchors, strides, v4 = cases.head_geometry(cfg)
    return [torch.tensor(a, dtype=torch.float32) / s for a, s in zip(anchors, strides)], v4


@pytest.mark.parametrize("cfg", ["kaist_yolov3.cfg", "kaist_dyolov4_fshare_global_concat_se3.cfg"])
def test_build_targets_exact(cfg):
    gold = np.load(os.path.join(GOLDEN, "targets.npz"))
    av, _ = _anchor_vecs(cfg)
    shapes = cases.head_shapes(cfg, 2, 512, 640, 6)
    for name, tg in cases.target_cases().items():
        tcls, tbox, indices, anch = oloss.build_targets(shapes, tg, av, 0.2)
        for h in range(3):
            key = "%s|%s|%d|" % (cfg, name, h)
            idx = torch.stack([t.long() for t in indices[h]]).numpy() if len(indices[h][0]) else np.zeros((4, 0), np.int64)
            assert np.array_equal(idx, gold[key + "idx"]), key
            assert np.array_equal(tbox[h].numpy(), gold[key + "tbox"]), key
            assert np.array_equal(anch[h].numpy(), gold[key + "anch"]), key
            assert np.array_equal(tcls[h].numpy(), gold[key + "tcls"]), key


@pytest.mark.parametrize("case", cases.loss_cases(), ids=lambda c: c["name"])
def test_compute_loss_matches_reference(case):
    gold = np.load(os.path.join(GOLDEN, "loss.npz"))
    av, v4 = _anchor_vecs(case["cfg"])
    p = cases.loss_preds(case)
    for t in p:
        t.requires_grad_(True)
    out = oloss.compute_loss(p, cases.loss_targets(case), av, cases.load_hyp(case["hyp"]), case["nc"], case["gr"], v4)
    got = np.array([out["box_loss"].item(), out["obj_loss"].item(), out["class_loss"].item()], np.float32)
    assert np.allclose(got, gold[case["name"] + "|losses"], rtol=1e-6, atol=1e-7)
    (out["box_loss"] + out["obj_loss"] + out["class_loss"]).backward()
    for i, t in enumerate(p):
        assert np.allclose(t.grad.numpy(), gold[case["name"] + "|dp%d" % i], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("case", cases.focal_loss_cases(), ids=lambda c: c["name"])
def test_focal_loss_matches_reference(case):
    """hyp['fl_gamma'] > 0: both BCE terms wrapped in the reference's FocalLoss (utils.py:174-201, :236-238); fixture
    loss_focal.npz = the reference's compute_loss on the same seeded inputs (tests/golden/make_golden_round3b.py)"""
    gold = np.load(os.path.join(GOLDEN, "loss_focal.npz"))
    av, v4 = _anchor_vecs(case["cfg"])
    p = cases.loss_preds(case)
    for t in p:
        t.requires_grad_(True)
    hyp = dict(cases.load_hyp(case["hyp"]), fl_gamma=case["fl_gamma"])
    out = oloss.compute_loss(p, cases.loss_targets(case), av, hyp, case["nc"], case["gr"], v4)
    got = np.array([out["box_loss"].item(), out["obj_loss"].item(), out["class_loss"].item()], np.float32)
    assert np.allclose(got, gold[case["name"] + "|losses"], rtol=1e-6, atol=1e-7)
    plain = np.load(os.path.join(GOLDEN, "loss.npz"))
    (out["box_loss"] + out["obj_loss"] + out["class_loss"]).backward()
    for i, t in enumerate(p):
        assert np.allclose(t.grad.numpy(), gold[case["name"] + "|dp%d" % i], rtol=1e-5, atol=1e-8)
    assert not np.allclose(got[1], plain["v4_ciou_nc2|losses"][1], rtol=1e-2)      # (the focal term really is on)


@pytest.mark.parametrize("case", cases.nms_cases(), ids=lambda c: c["name"])
def test_nms_matches_reference(case):
    gold = np.load(os.path.join(GOLDEN, "nms.npz"))
    pred = cases.nms_pred(case)
    out, rows = onms.non_max_suppression(pred, case["conf"], case["iou"], multi_label=case["multi"], classes=case["classes"],
                                         agnostic=case["agnostic"], return_indices=True)
    for b, o in enumerate(out):
        g = gold["%s|%d" % (case["name"], b)]
        if o is None:
            assert g.shape[0] == 0
        else:
            assert np.array_equal(o.numpy(), g), (case["name"], b)
            assert len(rows[b]) == len(o) <= 100


# ---------------------------------------------------------------- torchvision.ops.nms semantics
def test_nms_known_answers():
    # three boxes: B overlaps A with IoU exactly 1/3, C is disjoint
    boxes = np.array([[0, 0, 2, 2], [1, 0, 3, 2], [10, 10, 12, 12]], np.float32)
    scores = np.array([0.9, 0.8, 0.7], np.float32)
    assert onms.nms_numpy(boxes, scores, 0.5).tolist() == [0, 1, 2]
    assert onms.nms_numpy(boxes, scores, 0.3).tolist() == [0, 2]
    # IoU == threshold exactly is NOT suppressed (strict >): IoU(A, B) = 2/6
    assert onms.nms_numpy(boxes, scores, np.float32(2.0) / np.float32(6.0)).tolist() == [0, 1, 2]
    # equal scores: stable order -> lower index first
    boxes = np.array([[0, 0, 4, 4], [0, 0, 4, 4], [0, 0, 4, 4]], np.float32)
    assert onms.nms_numpy(boxes, np.array([0.5, 0.5, 0.5], np.float32), 0.5).tolist() == [0]
    assert onms.nms_numpy(boxes, np.array([0.5, 0.7, 0.5], np.float32), 0.5).tolist() == [1]
    # zero-area boxes: IoU is 0/0 = nan -> never greater than the threshold -> both kept
    boxes = np.array([[1, 1, 1, 1], [1, 1, 1, 1]], np.float32)
    assert onms.nms_numpy(boxes, np.array([0.6, 0.5], np.float32), 0.5).tolist() == [0, 1]
    # a chain of 200 boxes each overlapping only its neighbour (IoU 1/3 < 0.5): all kept, score order
    n = 200
    boxes = np.stack([np.array([i, 0, i + 2, 2], np.float32) for i in range(n)])
    scores = np.linspace(0.99, 0.01, n).astype(np.float32)
    assert onms.nms_numpy(boxes, scores, 0.5).tolist() == list(range(n))
    # same chain with threshold 0.3: every second box survives
    assert onms.nms_numpy(boxes, scores, 0.3).tolist() == list(range(0, n, 2))
    assert onms.nms_numpy(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5).tolist() == []


def test_box_helpers():
    x = torch.tensor([[10., 20., 4., 8.], [0., 0., 2., 2.]])
    y = onms.xywh2xyxy(x)
    assert torch.equal(y, torch.tensor([[8., 16., 12., 24.], [-1., -1., 1., 1.]]))
    assert torch.equal(onms.xyxy2xywh(y), x)
    c = torch.tensor([[-5., 10., 700., 600.]])
    onms.clip_coords(c, (512, 640))
    assert torch.equal(c, torch.tensor([[0., 10., 640., 512.]]))
    # letterbox 416x512 input of a 512x640 original: gain = 0.8, pad_y = (416 - 409.6) / 2 = 3.2
    c = torch.tensor([[80., 40., 160., 120.]])
    onms.scale_coords((416, 512), c, (512, 640))
    assert torch.allclose(c, torch.tensor([[100., 46., 200., 146.]]))


@pytest.mark.parametrize("case", cases.decode_cases(), ids=lambda c: c["name"])
def test_yolo_decode_matches_reference(case):
    """oracle YOLO head decode (models.py:234-258) against the reference YOLOLayer's eval output (fixture decode.npz)"""
    from oracle.model import OracleNet
    gold = np.load(os.path.join(GOLDEN, "decode.npz"))
    net = OracleNet.__new__(OracleNet)
    net.v4 = case["bf"] == "yolov4"
    L = dict(na=3, nc=case["nc"], stride=case["stride"], anchors=torch.tensor(case["anchors"]))
    io, p = net._yolo(L, cases.decode_logits(case), training=False)
    ref = gold[case["name"] + "|io"]
    assert io.shape == ref.shape
    assert np.allclose(io.numpy(), ref, rtol=1e-6, atol=1e-6)
