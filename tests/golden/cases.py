"""Seeded input generators shared by the golden generator (run against the reference) and the tests
(run against the oracle and the HIP product).  Pure torch / numpy, no reference code."""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load_hyp(name):
    with open(os.path.join(HERE, name + ".json")) as f:
        return json.load(f)


def _sections(cfg_name):
    base = os.path.basename(cfg_name)
    if base.endswith(".cfg"):
        base = base[:-4]
    with open(os.path.join(HERE, "parse_%s.json" % base)) as f:
        return json.load(f)


def head_geometry(cfg_name):
    """(anchors per head [3][na][2], strides, v4) with the reference's cfg-path rules (models.py:124-131)"""
    anchors = []
    for s in _sections(cfg_name):
        if s["type"] == "yolo":
            a = np.array(s["anchors"]["__ndarray__"], dtype=np.float64)
            anchors.append(a[s["mask"]].tolist())
    base = os.path.basename(cfg_name)
    strides = [32, 16, 8] if any(x in base for x in ("yolov-tiny", "fpn", "yolov3")) else [8, 16, 32]
    return anchors, strides[:len(anchors)], "yolov4" in base


def head_shapes(cfg_name, B, H, W, no):
    anchors, strides, _ = head_geometry(cfg_name)
    return [(B, len(a), H // s, W // s, no) for a, s in zip(anchors, strides)]


# ------------------------------------------------------------------------------------------ targets
def target_cases():
    """[nt,6] = (image, class, xc, yc, w, h) normalised; B = 2, 512x640 heads"""
    g = torch.Generator().manual_seed(11)
    c = {}
    n = 12
    t = torch.zeros(n, 6)
    t[:, 0] = torch.randint(0, 2, (n,), generator=g).float().sort()[0]
    t[:, 2:4] = torch.rand(n, 2, generator=g) * 0.8 + 0.1
    t[:, 4] = (torch.rand(n, generator=g) * 60 + 16) / 640
    t[:, 5] = (torch.rand(n, generator=g) * 120 + 32) / 512
    c["random"] = t
    # centre exactly on cell boundaries of every head (x = k/80 .. ), and just below 1.0
    c["boundary"] = torch.tensor([[0, 0, 40 / 80, 32 / 64, 20 / 640, 44 / 512],
                                  [0, 0, 0.25, 0.75, 30 / 640, 60 / 512],
                                  [1, 0, 0.999999, 0.999999, 24 / 640, 50 / 512],
                                  [1, 0, 1e-6, 1e-6, 40 / 640, 90 / 512]])
    # two targets in the same cell that match the same anchors (last one must win the obj target)
    c["duplicate_cell"] = torch.tensor([[0, 0, 0.5031, 0.5021, 20 / 640, 44 / 512],
                                        [0, 0, 0.5032, 0.5022, 21 / 640, 45 / 512],
                                        [1, 0, 0.3, 0.3, 60 / 640, 140 / 512]])
    # sizes that match none / one / all anchors of some head
    c["anchor_counts"] = torch.tensor([[0, 0, 0.4, 0.4, 2 / 640, 2 / 512],
                                       [0, 0, 0.6, 0.6, 17 / 640, 33 / 512],
                                       [1, 0, 0.2, 0.7, 300 / 640, 400 / 512],
                                       [1, 0, 0.7, 0.2, 75 / 640, 150 / 512]])
    c["empty"] = torch.zeros(0, 6)
    return c


# ------------------------------------------------------------------------------------------ loss
def loss_cases():
    out = []
    for (name, cfg, nc, hyp, gr) in [
            ("v4_ciou_nc1", "kaist_dyolov4_fshare_global_concat_se3.cfg", 1, "hyp.scratch.4", 1.0),
            ("v4_ciou_nc1_gr0", "kaist_dyolov4_fshare_global_concat_se3.cfg", 1, "hyp.scratch.4", 0.0),
            ("v3_giou_nc1", "kaist_yolov3.cfg", 1, "hyp.scratch", 1.0),
            ("v4_ciou_nc2", "kaist_dyolov4_fshare_global_concat_se3.cfg", 2, "hyp.scratch.4", 0.5),
            ("v3_giou_nc2", "kaist_yolov3.cfg", 2, "hyp.scratch", 1.0),
            ("v4_ciou_empty", "kaist_dyolov4_fshare_global_concat_se3.cfg", 1, "hyp.scratch.4", 1.0)]:
        out.append(dict(name=name, cfg=cfg, nc=nc, hyp=hyp, gr=gr, B=3, H=128, W=160, seed=len(out) + 21))
    return out


def focal_loss_cases():
    """hyp['fl_gamma'] > 0 (reference utils.py:236-238: both BCE terms wrapped in FocalLoss) -- off in the two shipped hyp
    files, so the fixture overrides it (loss_focal.npz, tests/golden/make_golden_round3b.py)"""
    out = []
    for (name, cfg, nc, hyp, gr, gamma) in [
            ("v4_ciou_nc2_fl15", "kaist_dyolov4_fshare_global_concat_se3.cfg", 2, "hyp.scratch.4", 0.5, 1.5),
            ("v3_giou_nc1_fl15", "kaist_yolov3.cfg", 1, "hyp.scratch", 1.0, 1.5),
            ("v4_ciou_nc2_fl2", "kaist_dyolov4_fshare_global_concat_se3.cfg", 2, "hyp.scratch.4", 1.0, 2.0)]:
        out.append(dict(name=name, cfg=cfg, nc=nc, hyp=hyp, gr=gr, fl_gamma=gamma, B=3, H=128, W=160, seed=len(out) + 61))
    return out


def mismatch_inputs():
    """seeded image pair of the mismatched-channel [shortcut] fixture (mismatch.npz)"""
    g = torch.Generator().manual_seed(77)
    return torch.rand(2, 3, 32, 64, generator=g), torch.rand(2, 3, 32, 64, generator=g)


def loss_preds(case):
    g = torch.Generator().manual_seed(case["seed"])
    return [torch.randn(s, generator=g) * 1.5 for s in head_shapes(case["cfg"], case["B"], case["H"], case["W"], 5 + case["nc"])]


def loss_targets(case):
    if case["name"].endswith("empty"):
        return torch.zeros(0, 6)
    g = torch.Generator().manual_seed(case["seed"] + 1000)
    n = 14
    t = torch.zeros(n, 6)
    t[:, 0] = torch.randint(0, case["B"], (n,), generator=g).float().sort()[0]
    t[:, 1] = torch.randint(0, case["nc"], (n,), generator=g).float()
    t[:, 2:4] = torch.rand(n, 2, generator=g) * 0.9 + 0.05
    t[:, 4] = (torch.rand(n, generator=g) * 60 + 16) / 640
    t[:, 5] = (torch.rand(n, generator=g) * 120 + 32) / 512
    t[1, 2:6] = t[0, 2:6] + 1e-4          # a duplicate cell
    return t


# ------------------------------------------------------------------------------------------ nms
def nms_cases():
    return [
        dict(name="sparse", B=2, N=20160, nc=1, kind="sparse", conf=0.01, iou=0.6, multi=False, classes=None, agnostic=False, seed=31),
        dict(name="dense", B=1, N=20160, nc=1, kind="dense", conf=0.01, iou=0.6, multi=False, classes=None, agnostic=False, seed=32),
        dict(name="ties", B=2, N=600, nc=1, kind="ties", conf=0.1, iou=0.5, multi=True, classes=None, agnostic=False, seed=33),
        dict(name="empty", B=2, N=500, nc=1, kind="empty", conf=0.1, iou=0.6, multi=True, classes=None, agnostic=False, seed=34),
        dict(name="nc2_multi", B=2, N=3000, nc=2, kind="sparse", conf=0.1, iou=0.6, multi=True, classes=None, agnostic=False, seed=35),
        dict(name="nc2_best", B=2, N=3000, nc=2, kind="sparse", conf=0.1, iou=0.6, multi=False, classes=None, agnostic=False, seed=36),
        dict(name="nc2_agnostic", B=1, N=3000, nc=2, kind="sparse", conf=0.1, iou=0.6, multi=True, classes=None, agnostic=True, seed=37),
        dict(name="nc2_classes", B=1, N=3000, nc=2, kind="sparse", conf=0.1, iou=0.6, multi=True, classes=[1], agnostic=False, seed=38),
    ]


def nms_pred(case):
    """decoded predictions [B, N, 5+nc] (cx, cy, w, h, obj, cls...) in pixels of a 512x640 image"""
    g = torch.Generator().manual_seed(case["seed"])
    B, N, nc = case["B"], case["N"], case["nc"]
    p = torch.zeros(B, N, 5 + nc)
    if case["kind"] in ("sparse", "ties"):
        ncl = 25                                           # object clusters per image
        cx = torch.rand(B, ncl, generator=g) * 560 + 40
        cy = torch.rand(B, ncl, generator=g) * 430 + 40
        w = torch.rand(B, ncl, generator=g) * 60 + 16
        h = torch.rand(B, ncl, generator=g) * 120 + 32
        which = torch.randint(0, ncl, (B, N), generator=g)
        jit = torch.randn(B, N, 4, generator=g) * torch.tensor([4.0, 4.0, 3.0, 6.0])
        p[..., 0] = torch.gather(cx, 1, which) + jit[..., 0]
        p[..., 1] = torch.gather(cy, 1, which) + jit[..., 1]
        p[..., 2] = (torch.gather(w, 1, which) + jit[..., 2]).clamp(min=1.0)
        p[..., 3] = (torch.gather(h, 1, which) + jit[..., 3]).clamp(min=1.0)
        hot = torch.rand(B, N, generator=g) < (300.0 / N if case["kind"] == "sparse" else 0.5)
        p[..., 4] = torch.where(hot, torch.rand(B, N, generator=g) * 0.9 + 0.1, torch.rand(B, N, generator=g) * 0.005)
        p[..., 5:] = torch.rand(B, N, nc, generator=g) * 0.6 + 0.4
        if case["kind"] == "ties":
            p[..., 4] = (p[..., 4] * 8).round() / 8       # many equal scores
            p[..., 5:] = 1.0
            p[:, 1::3, :4] = p[:, 0:-1:3, :4][:, :p[:, 1::3].shape[1]]   # exact duplicate boxes
    elif case["kind"] == "dense":
        p[..., 0] = torch.rand(B, N, generator=g) * 600 + 20
        p[..., 1] = torch.rand(B, N, generator=g) * 470 + 20
        p[..., 2] = torch.rand(B, N, generator=g) * 60 + 16
        p[..., 3] = torch.rand(B, N, generator=g) * 120 + 32
        p[..., 4] = torch.rand(B, N, generator=g) * 0.5 + 0.5
        p[..., 5:] = torch.rand(B, N, nc, generator=g) * 0.5 + 0.5
    elif case["kind"] == "empty":
        p[..., :4] = torch.rand(B, N, 4, generator=g) * 100 + 10
        p[..., 4] = torch.rand(B, N, generator=g) * 0.05
        p[..., 5:] = 1.0
        p[1, :10, 4] = 0.9                                 # image 1: confident but too small (w,h <= 2)
        p[1, :10, 2:4] = 1.5
    return p


# ------------------------------------------------------------------------------------------ AP / LAMR
def ap_cases():
    """inputs of compute_ap_lamr (reference other_utils/metrics.py:81): preds = list of
    {img_id, conf, bbox xyxy} sorted by conf desc; labels = per image [k,5] (cls, xc, yc, w, h) relative;
    shapes = per image (w, h).  Every image has at least one ground truth (the reference crashes otherwise)."""
    rng = np.random.RandomState(5)
    out = {}
    W, H = 640.0, 512.0
    for name, nimg, noise in (("clean", 12, 2.0), ("noisy", 20, 14.0)):
        labels, preds, shapes = [], [], []
        for i in range(nimg):
            k = rng.randint(1, 4)
            gts = []
            for _ in range(k):
                x1, y1 = rng.uniform(10, 500), rng.uniform(10, 380)
                w, h = rng.uniform(16, 76), rng.uniform(32, 120)
                gts.append([0.0, (x1 + w / 2) / W, (y1 + h / 2) / H, w / W, h / H])
                if rng.rand() < 0.85:                      # a detection near the ground truth
                    d = rng.randn(4) * noise
                    preds.append(dict(img_id=i, conf=float(rng.uniform(0.3, 1.0)),
                                      bbox=np.array([x1 + d[0], y1 + d[1], x1 + w + d[2], y1 + h + d[3]], dtype=np.float32)))
                if rng.rand() < 0.3:                       # a duplicate detection
                    preds.append(dict(img_id=i, conf=float(rng.uniform(0.2, 0.9)),
                                      bbox=np.array([x1 + 1, y1 + 1, x1 + w - 1, y1 + h - 1], dtype=np.float32)))
            for _ in range(rng.randint(0, 3)):             # false positives
                x1, y1 = rng.uniform(10, 500), rng.uniform(10, 380)
                preds.append(dict(img_id=i, conf=float(rng.uniform(0.05, 0.6)),
                                  bbox=np.array([x1, y1, x1 + 30, y1 + 60], dtype=np.float32)))
            labels.append(np.array(gts, dtype=np.float32))
            shapes.append((W, H))
        preds.sort(key=lambda r: -r["conf"])
        out[name] = (preds, labels, np.array(shapes))
    return out


# ------------------------------------------------------------------------------------------ train steps
def step_batch(step):
    g = torch.Generator().manual_seed(500 + step)
    x = torch.rand(2, 3, 128, 160, generator=g)
    y = torch.rand(2, 3, 128, 160, generator=g)
    n = 6
    t = torch.zeros(n, 6)
    t[:, 0] = torch.tensor([0, 0, 0, 1, 1, 1.])
    t[:, 2:4] = torch.rand(n, 2, generator=g) * 0.8 + 0.1
    t[:, 4] = (torch.rand(n, generator=g) * 60 + 16) / 160
    t[:, 5] = (torch.rand(n, generator=g) * 60 + 32) / 128
    return x, y, t


def sgd_step_batch(step, B=4, H=256, W=320):
    """batches of the SGD trajectory fixture (step_sgd.npz): large enough that the stride-32 BatchNorm layers see 320
    samples per channel -- at 2 x 128 x 160 (40 samples) one 1e-6 SGD step changes the objectness loss by 11 %"""
    g = torch.Generator().manual_seed(700 + step)
    x = torch.rand(B, 3, H, W, generator=g)
    y = torch.rand(B, 3, H, W, generator=g)
    n = 3 * B
    t = torch.zeros(n, 6)
    t[:, 0] = torch.arange(B).repeat_interleave(3).float()
    t[:, 2:4] = torch.rand(n, 2, generator=g) * 0.8 + 0.1
    t[:, 4] = (torch.rand(n, generator=g) * 60 + 16) / W
    t[:, 5] = (torch.rand(n, generator=g) * 60 + 32) / H
    return x, y, t


def step_probe_names():
    return ["module_list.0.Conv2d.weight", "module_list.0.BatchNorm2d.weight", "module_list.55.Conv2d.weight",
            "module_list.111.Conv2d.weight", "module_list.112.fc1.weight", "module_list.113.w",
            "module_list.223.Conv2d.weight", "module_list.258.Conv2d.weight", "module_list.258.Conv2d.bias",
            "module_list.280.Conv2d.bias", "module_list.279.BatchNorm2d.running_var"]


# ------------------------------------------------------------------------------------------ decode (G7)
def decode_cases():
    """YOLOLayer eval decode (reference models.py:234-258) on fixed logits: (name, bf_type, stride, ny, nx, anchors, nc)"""
    anchors = [[16.0, 32.0], [18.0, 42.0], [22.0, 44.0]]
    out = []
    for bf, strides in (("yolov3", (32, 16, 8)), ("yolov4", (8, 16, 32))):
        for s in strides:
            out.append(dict(name="%s_s%d" % (bf, s), bf=bf, stride=s, ny=512 // s, nx=640 // s, anchors=anchors, nc=1))
    out.append(dict(name="yolov4_s16_nc2", bf="yolov4", stride=16, ny=8, nx=10, anchors=anchors, nc=2))
    return out


def decode_logits(case):
    g = torch.Generator().manual_seed(900 + case["stride"] + (7 if case["bf"] == "yolov4" else 0) + case["nc"])
    na, no = len(case["anchors"]), 5 + case["nc"]
    return torch.randn(1, na * no, case["ny"], case["nx"], generator=g) * 1.5
