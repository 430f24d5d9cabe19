"""Round-4 fixture, produced by RUNNING THE REFERENCE in this container (tests/golden/ref_import.py):

  evalap_trained.npz   the evaluation chain of evaluate.py:64-117 on a network whose SCORES ARE SEPARATED.

The round-2 AP fixture (evalap.npz) uses random weights: its scores are near-ties, and bf16 arithmetic alone moves the
reference's own AP from 0.546 to ~0.05 -- a bf16 AP comparison on it says nothing.  Here the reference is TRAINED until its
detections are confident:

  * data: 16 pairs of 128 x 160 uint8 images, noise background with 1-3 bright UPRIGHT rectangles each (same boxes in the visible and
    the LWIR image, different contrast) = the labels, plus 1-2 equally bright FLAT rectangles that are not labelled (seeded numpy generator, rebuilt by the tests);
  * network: the target cfg unchanged, weights conditioned_state(synth_state(5)) (the oracle's seeded recipe made well-conditioned:
    identity tap + 0.3 x He noise, see conditioned_state; identical in the tests), BatchNorm
    running statistics calibrated on the data by the reference (momentum 1, as in evalap.npz) -- stored;
  * training: the reference's own `compute_loss` (build_utils/utils.py:209-293, hyp.scratch.4) and torch.optim.Adam, 150
    full-batch steps on the THREE HEAD CONVS only (module_list.{j-1}.Conv2d of every [yolo] j; everything below them is frozen in
    eval mode, so their inputs are computed once).  The backbone's 116 M weights stay reproducible from the seed; what
    training changed -- 3 x (18 x Cin + 18) numbers -- is stored;
  * evaluation: reference YOLO eval forward -> non_max_suppression(0.1, 0.6, multi_label=False) -> scale_coords ->
    other_utils.metrics.compute_ap_lamr against the true labels: the stored AP / LAMR / detections.

    python tests/golden/make_golden_round4.py          (evalap_trained.npz)
    python tests/golden/make_golden_round4.py traj     (traj_trained.npz: three SGD steps of the reference from that state)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

CFG = "kaist_dyolov4_fshare_global_concat_se3"
NB, H, W = 16, 128, 160
SHAPE0, RATIO_PAD = (512, 640), ((0.25, 0.25), (0.0, 0.0))
SEED_W = 5
CONF, IOU = 0.1, 0.6
STEPS = 150


def dataset():
    """-> (visible uint8 [NB,3,H,W], lwir uint8 [NB,3,H,W], targets float32 [n,6] = (image, class, xc, yc, w, h) normalised)"""
    r = np.random.RandomState(404)
    v = r.randint(0, 90, size=(NB, 3, H, W)).astype(np.uint8)
    l = r.randint(0, 70, size=(NB, 3, H, W)).astype(np.uint8)
    tg = []
    for b in range(NB):
        taken = []
        for _ in range(int(r.randint(1, 4))):
            for _try in range(50):
                bw, bh = int(r.randint(14, 44)), int(r.randint(24, 72))
                x0, y0 = int(r.randint(2, W - bw - 2)), int(r.randint(2, H - bh - 2))
                box = (x0, y0, x0 + bw, y0 + bh)
                if all(box[2] + 6 < t[0] or t[2] + 6 < box[0] or box[3] + 6 < t[1] or t[3] + 6 < box[1] for t in taken):
                    break
            else:
                continue
            taken.append(box)
            lv, ll = int(r.randint(190, 256)), int(r.randint(150, 256))
            v[b, :, y0:y0 + bh, x0:x0 + bw] = np.clip(lv + r.randint(-12, 13, size=(3, bh, bw)), 0, 255).astype(np.uint8)
            l[b, :, y0:y0 + bh, x0:x0 + bw] = np.clip(ll + r.randint(-12, 13, size=(3, bh, bw)), 0, 255).astype(np.uint8)
            tg.append([b, 0, (x0 + bw / 2) / W, (y0 + bh / 2) / H, bw / W, bh / H])
        # distractors: equally bright, but flat (wider than tall) and NOT labelled -- the heads have to tell them from the upright
        # objects, so the trained network keeps a few confident false positives and AP stays off the saturated value 1.0
        for _ in range(int(r.randint(1, 3))):
            for _try in range(50):
                bw, bh = int(r.randint(28, 60)), int(r.randint(8, 18))
                x0, y0 = int(r.randint(2, W - bw - 2)), int(r.randint(2, H - bh - 2))
                box = (x0, y0, x0 + bw, y0 + bh)
                if all(box[2] + 6 < t[0] or t[2] + 6 < box[0] or box[3] + 6 < t[1] or t[3] + 6 < box[1] for t in taken):
                    break
            else:
                continue
            taken.append(box)
            lv, ll = int(r.randint(190, 256)), int(r.randint(150, 256))
            v[b, :, y0:y0 + bh, x0:x0 + bw] = np.clip(lv + r.randint(-12, 13, size=(3, bh, bw)), 0, 255).astype(np.uint8)
            l[b, :, y0:y0 + bh, x0:x0 + bw] = np.clip(ll + r.randint(-12, 13, size=(3, bh, bw)), 0, 255).astype(np.uint8)
    return torch.from_numpy(v), torch.from_numpy(l), torch.tensor(tg, dtype=torch.float32)


def conditioned_state(sd, a=0.3):
    """The weight recipe of this fixture: every convolution with more than 3 input channels becomes
    `identity tap + a x (seeded He noise)` -- W[co, co % Cin, centre] += 1 (Cout >= Cin) or W[ci % Cout, ci, centre] += Cout / Cin.
    Why: with PURE random weights the network is chaotic in the reference's own arithmetic -- bf16 rounding alone (oracle,
    emulate_bf16) grows from 0.2 % at the stem to 55-80 % of the tensor norm at the head inputs (2.5 % by section 48, 11 % by the
    SPP block, x 2 per conv through the PANet), and the trained heads then see noise: AP 0.92 -> 0.02 in bf16 whatever the
    kernels do.  A TRAINED network is not chaotic; neither is this recipe: the same measurement gives 3.7 / 4.6 / 6.6 % at the
    three head inputs, the conditioning one expects of trained weights, while every layer still mixes all its channels."""
    out = {}
    for k, w in sd.items():
        if k.endswith("Conv2d.weight") and w.dim() == 4 and w.shape[1] > 3:
            co, ci, kh, kw = w.shape
            eye = torch.zeros_like(w)
            if co >= ci:
                idx = torch.arange(co)
                eye[idx, idx % ci, kh // 2, kw // 2] = 1.0
            else:
                idx = torch.arange(ci)
                eye[idx % co, idx, kh // 2, kw // 2] = float(co) / ci
            out[k] = a * w + eye
        else:
            out[k] = w
    return out


def labels_of(targets):
    """per-image label arrays [n,5] = (class, xc, yc, w, h) and the (w, h) shapes compute_ap_lamr expects"""
    labels = []
    for b in range(NB):
        rows = targets[targets[:, 0] == b][:, 1:].numpy().astype(np.float32)
        labels.append(rows)
    shapes = np.array([(SHAPE0[1], SHAPE0[0])] * NB, dtype=np.int64)
    return labels, shapes


def main():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import cases
    from ref_import import import_reference
    torch.set_num_threads(8)
    ref_models, ref_utils, ref_parse, ref_metrics = import_reference()
    from oracle.model import OracleNet
    cfg = "config/%s.cfg" % CFG
    defs = ref_parse.parse_model_cfg(cfg)
    sd = conditioned_state(OracleNet(defs, cfg).synth_state(SEED_W))
    torch.manual_seed(0)
    m = ref_models.YOLO(cfg, (H, W))
    m.load_state_dict(sd)
    hyp = cases.load_hyp("hyp.scratch.4")
    m.nc, m.hyp, m.gr = 1, hyp, 1.0
    v, l, targets = dataset()
    x, y = v.float() / 255.0, l.float() / 255.0
    # ---- BatchNorm running statistics = statistics of the data set (momentum 1), as in evalap.npz
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 1.0
    with torch.no_grad():
        m(x, y)
    m.eval()
    calib = {k: t.clone() for k, t in m.state_dict().items() if k.endswith(("running_mean", "running_var"))}
    # ---- inputs of the three head convs (everything below them is frozen, eval mode): computed once
    feats = {}
    hooks = []
    for j in m.yolo_layers:
        conv = m.module_list[j - 1]
        hooks.append(conv.register_forward_hook(lambda mod, inp, out, j=j: feats.__setitem__(j, inp[0].detach())))
    with torch.no_grad():
        m(x, y)
    for h in hooks:
        h.remove()
    heads = [m.module_list[j - 1] for j in m.yolo_layers]
    params = [p for hd in heads for p in hd.parameters()]
    for p in m.parameters():
        p.requires_grad_(False)
    for p in params:
        p.requires_grad_(True)
    opt = torch.optim.Adam(params, lr=2e-3, betas=(hyp["momentum"], 0.999))
    for j in m.yolo_layers:
        m.module_list[j].train()                       # (YOLOLayer: training mode = raw p, models.py:229)
    for step in range(STEPS):
        p = [m.module_list[j](hd(feats[j])) for j, hd in zip(m.yolo_layers, heads)]
        ld = ref_utils.compute_loss(p, targets, m)
        loss = ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 50 == 0 or step == STEPS - 1:
            print("step %3d  box %.4f obj %.4f cls %.4f" % (step, ld["box_loss"].item(), ld["obj_loss"].item(), ld["class_loss"].item()), flush=True)
    m.eval()
    # ---- the reference's evaluation chain
    with torch.no_grad():
        pred = m(x, y)[0]
    dets = ref_utils.non_max_suppression(pred, conf_thres=CONF, iou_thres=IOU, multi_label=False)
    preds, per_image = [], []
    for idx, p in enumerate(dets):
        if p is None:
            per_image.append(np.zeros((0, 6), np.float32))
            continue
        boxes = ref_utils.scale_coords((H, W), p[:, :4].clone(), SHAPE0, RATIO_PAD)
        per_image.append(torch.cat([boxes, p[:, 4:6]], 1).numpy())
        for i in range(p.shape[0]):
            preds.append({"img_id": idx, "conf": p[i, 4].item(), "bbox": boxes[i].numpy()})
    preds.sort(key=lambda q: float(q["conf"]), reverse=True)
    labels, shapes = labels_of(targets)
    res = ref_metrics.compute_ap_lamr(preds, [lb.copy() for lb in labels], shapes)
    sc = pred[..., 4] * pred[..., 5:].max(-1).values
    rec = {"ap": np.float64(res["ap"]), "lamr": np.float64(res["lamr"]), "ndet": np.array([d.shape[0] for d in per_image]),
           "io": pred.numpy().astype(np.float32), "n_targets": np.int64(targets.shape[0]),
           "score_hist": np.histogram(sc.numpy().ravel(), bins=10, range=(0, 1))[0]}
    sdn = m.state_dict()
    for j in m.yolo_layers:
        for leaf in ("weight", "bias"):
            k = "module_list.%d.Conv2d.%s" % (j - 1, leaf)
            rec["head|" + k] = sdn[k].numpy()
    for k, t in calib.items():
        rec["bn|" + k] = t.numpy()
    for idx in range(NB):
        rec["det%d" % idx] = per_image[idx]
    np.savez_compressed(os.path.join(HERE, "evalap_trained.npz"), **rec)
    print("trained-head AP fixture: %d targets, %d detections, AP %.5f, LAMR %.5f; scores by decile %s"
          % (targets.shape[0], len(preds), res["ap"], res["lamr"], rec["score_hist"].tolist()))


TRAJ_LRS = (1e-6, 1e-5, 1e-4)
TRAJ_PROBES = ("module_list.1.Conv2d.weight", "module_list.1.BatchNorm2d.weight", "module_list.23.Conv2d.weight",
               "module_list.60.Conv2d.weight", "module_list.60.BatchNorm2d.bias", "module_list.113.w")


def trajectory():
    """traj_trained.npz: three steps of the reference's SGD branch (train.py:86-89: SGD + Nesterov momentum + weight decay) on
    the WHOLE network, starting from this fixture's state (conditioned weights, calibrated BatchNorm statistics, trained heads),
    full batch of the 16 synthetic pairs, train-mode BatchNorm (320+ samples per channel even at stride 32).  VERDICT r3 weak #1:
    the random-weight trajectories (step.npz, step_sgd.npz) can only be held to 10-30 %; this network is not chaotic, so the same
    comparison is sharp.  Stored per learning rate: the three loss triples and, for a few parameters, the update's norm / sum."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import cases
    from ref_import import import_reference
    torch.set_num_threads(8)
    ref_models, ref_utils, ref_parse, ref_metrics = import_reference()
    from oracle.model import OracleNet
    cfg = "config/%s.cfg" % CFG
    defs = ref_parse.parse_model_cfg(cfg)
    gold = np.load(os.path.join(HERE, "evalap_trained.npz"))
    sd0 = conditioned_state(OracleNet(defs, cfg).synth_state(SEED_W))
    for k in gold.files:
        if k.startswith(("bn|", "head|")):
            sd0[k.split("|", 1)[1]] = torch.from_numpy(gold[k])
    hyp = cases.load_hyp("hyp.scratch.4")
    v, l, targets = dataset()
    x, y = v.float() / 255.0, l.float() / 255.0
    rec = {"lrs": np.array(TRAJ_LRS), "momentum": np.float64(hyp["momentum"]), "weight_decay": np.float64(hyp["weight_decay"])}
    for q, lr in enumerate(TRAJ_LRS):
        torch.manual_seed(0)
        m = ref_models.YOLO(cfg, (H, W))
        m.load_state_dict(sd0)
        m.nc, m.hyp, m.gr = 1, hyp, 1.0
        m.train()
        p0 = {k: t.detach().clone().double() for k, t in m.state_dict().items() if k in TRAJ_PROBES}
        opt = torch.optim.SGD(m.parameters(), lr=lr, momentum=hyp["momentum"], weight_decay=hyp["weight_decay"], nesterov=True)
        losses = []
        for step in range(3):
            pred = m(x, y)
            ld = ref_utils.compute_loss(pred, targets, m)
            loss = ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append([ld["box_loss"].item(), ld["obj_loss"].item(), ld["class_loss"].item()])
            print("lr %g step %d losses %s" % (lr, step, losses[-1]), flush=True)
        rec["losses%d" % q] = np.array(losses)
        sd = m.state_dict()
        rec["delta%d" % q] = np.array([[(sd[k].double() - p0[k]).norm().item(), (sd[k].double() - p0[k]).sum().item()] for k in TRAJ_PROBES])
    np.savez_compressed(os.path.join(HERE, "traj_trained.npz"), **rec)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "traj":
        trajectory()
    else:
        main()
