"""AP / LAMR evaluator: the oracle restatement and the product's vectorised version against the golden
vectors produced by the reference's compute_ap_lamr (tests/golden/make_golden.py ap), plus hand-computed cases."""
import os
import sys

import numpy as np
import pytest

from helpers import GOLDEN

sys.path.insert(0, GOLDEN)
import cases  # noqa: E402


def _impls():
    from oracle import metrics as om
    from other_utils import metrics as pm
    return [("oracle", om), ("product", pm)]


@pytest.mark.parametrize("which", ["oracle", "product"])
@pytest.mark.parametrize("case", ["clean", "noisy"])
def test_ap_lamr_matches_reference_golden(which, case):
    mod = dict(_impls())[which]
    gold = np.load(os.path.join(GOLDEN, "ap.npz"))
    preds, labels, shapes = cases.ap_cases()[case]
    before = [l.copy() for l in labels]
    out = mod.compute_ap_lamr(preds, labels, shapes)
    for k in ("recall", "precision", "fppi", "mr"):
        assert np.array_equal(out[k], gold["%s|%s" % (case, k)]), k       # curves identical
    assert abs(out["ap"] - gold[case + "|ap_lamr"][0]) < 1e-12
    assert abs(out["lamr"] - gold[case + "|ap_lamr"][1]) < 1e-12
    assert all(np.array_equal(a, b) for a, b in zip(before, labels)), "inputs must not be modified"


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_known_answers(which):
    mod = dict(_impls())[which]
    # precision envelope: points (r,p) = (.5,1), (.5,.5), (1,.667) -> AP = .5*1 + .5*.667
    ap = mod.voc_ap(np.array([0.5, 0.5, 1.0]), np.array([1.0, 0.5, 2 / 3]))
    assert abs(ap - (0.5 + 0.5 * 2 / 3)) < 1e-12
    # one image, one ground truth 100x200 at (100,100); a perfect hit, its duplicate, one miss
    labels = [np.array([[0, 150 / 640, 200 / 512, 100 / 640, 200 / 512]], dtype=np.float32)]
    shapes = np.array([(640.0, 512.0)])
    box = np.array([100, 100, 200, 300], dtype=np.float32)
    preds = [dict(img_id=0, conf=0.9, bbox=box), dict(img_id=0, conf=0.8, bbox=box + 1),
             dict(img_id=0, conf=0.1, bbox=np.array([400, 10, 430, 70], dtype=np.float32))]
    out = mod.compute_ap_lamr(preds, labels, shapes)
    assert out["recall"].tolist() == [1.0, 1.0, 1.0]
    assert np.allclose(out["precision"], [1.0, 0.5, 1 / 3])
    assert out["fppi"].tolist() == [0.0, 1.0, 2.0] and out["mr"].tolist() == [0.0, 0.0, 0.0]
    assert abs(out["ap"] - 1.0) < 1e-12
    assert abs(out["lamr"] - 1e-10) < 1e-15               # miss rate 0 at every reference point, floored at 1e-10
    # IoU exactly at the threshold counts as a hit (>=), pixel-inclusive extents
    gt = [np.array([[0, 0.5, 0.5, 0.5, 0.5]], dtype=np.float32)]       # on a 20x20 image: x 5..15, y 5..15 -> 11x11 px
    p = [dict(img_id=0, conf=1.0, bbox=np.array([5, 5, 15, 10], dtype=np.float32))]   # 11x6 inside -> IoU 66/121 > .5
    assert mod.compute_ap_lamr(p, gt, np.array([(20.0, 20.0)]))["recall"].tolist() == [1.0]
    p = [dict(img_id=0, conf=1.0, bbox=np.array([5, 5, 15, 9], dtype=np.float32))]    # 11x5 -> 55/121 < .5
    assert mod.compute_ap_lamr(p, gt, np.array([(20.0, 20.0)]))["recall"].tolist() == [0.0]


def test_box_iou_matrix_matches_scalar_oracle():
    from oracle import metrics as om
    from other_utils import metrics as pm
    rng = np.random.RandomState(3)
    a = rng.uniform(0, 300, (1, 4)).astype(np.float32)
    a[:, 2:] += a[:, :2]
    b = rng.randint(0, 300, (17, 4)).astype(np.int32)
    b[:, 2:] += b[:, :2]
    m = pm.box_iou(a, b)[0]
    for k in range(17):
        assert m[k] == om.iou_plus_one(a[0], b[k])
