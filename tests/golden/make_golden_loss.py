"""Second half of the golden generator (see make_golden.py): target assignment, loss, NMS, AP/LAMR and
3-step training fixtures, all produced by RUNNING THE REFERENCE's functions on seeded inputs.

Inputs are rebuilt by the tests from `cases.py` (same seeded generators), so the fixtures only hold
the reference's outputs.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402


def fake_model(ref_models, cfg_name, nc, hyp, gr, img_hw=(512, 640)):
    """the attributes the reference's compute_loss / build_targets read from `model`
    (utils.py:225,252,271,274,316,321)"""
    anchors, strides, v4 = cases.head_geometry(cfg_name)
    mods = []
    for a, s in zip(anchors, strides):
        lay = ref_models.YOLOLayer(np.array(a, dtype=np.float64), nc, img_hw, s, "yolov4" if v4 else "yolov3")
        mods.append(lay)
    m = types.SimpleNamespace()
    m.module_list = mods
    m.yolo_layers = [0, 1, 2]
    m.hyp, m.gr, m.nc, m.cfg = hyp, gr, nc, cfg_name
    return m


def run(what, ref_models, ref_utils, ref_parse, ref_metrics, OUT):
    if "targets" in what:
        rec = {}
        for cfg in ("kaist_yolov3.cfg", "kaist_dyolov4_fshare_global_concat_se3.cfg"):
            hyp = cases.load_hyp("hyp.scratch.4")
            model = fake_model(ref_models, cfg, 1, hyp, 1.0)
            shapes = cases.head_shapes(cfg, 2, 512, 640, 6)
            p = [torch.zeros(s) for s in shapes]
            for name, tg in cases.target_cases().items():
                tcls, tbox, indices, anch = ref_utils.build_targets(p, tg, model)
                for h in range(3):
                    key = "%s|%s|%d|" % (cfg, name, h)
                    rec[key + "idx"] = torch.stack([t.long() for t in indices[h]]).numpy() if len(indices[h][0]) else np.zeros((4, 0), np.int64)
                    rec[key + "tbox"] = tbox[h].numpy()
                    rec[key + "anch"] = anch[h].numpy()
                    rec[key + "tcls"] = tcls[h].numpy()
        np.savez_compressed(os.path.join(OUT, "targets.npz"), **rec)
        print("targets fixture written (%d arrays)" % len(rec))

    if "decode" in what:
        rec = {}
        for case in cases.decode_cases():
            lay = ref_models.YOLOLayer(np.array(case["anchors"]), case["nc"], (512, 640), case["stride"], case["bf"])
            lay.eval()
            with torch.no_grad():
                io, p = lay(cases.decode_logits(case))
            rec[case["name"] + "|io"] = io.numpy()
            assert torch.equal(p, cases.decode_logits(case).view(1, lay.na, lay.no, case["ny"], case["nx"]).permute(0, 1, 3, 4, 2))
        np.savez_compressed(os.path.join(OUT, "decode.npz"), **rec)
        print("decode fixture written (%d arrays)" % len(rec))

    if "loss" in what:
        rec = {}
        for case in cases.loss_cases():
            hyp = cases.load_hyp(case["hyp"])
            model = fake_model(ref_models, case["cfg"], case["nc"], hyp, case["gr"], (case["H"], case["W"]))
            p = cases.loss_preds(case)
            for t in p:
                t.requires_grad_(True)
            tg = cases.loss_targets(case)
            out = ref_utils.compute_loss(p, tg, model)
            total = out["box_loss"] + out["obj_loss"] + out["class_loss"]
            total.backward()
            k = case["name"] + "|"
            rec[k + "losses"] = np.array([out["box_loss"].item(), out["obj_loss"].item(), out["class_loss"].item()], np.float32)
            for i, t in enumerate(p):
                rec[k + "dp%d" % i] = t.grad.numpy()
        np.savez_compressed(os.path.join(OUT, "loss.npz"), **rec)
        print("loss fixture written")

    if "nms" in what:
        rec = {}
        for case in cases.nms_cases():
            pred = cases.nms_pred(case)
            out = ref_utils.non_max_suppression(pred.clone(), case["conf"], case["iou"], multi_label=case["multi"],
                                                classes=case["classes"], agnostic=case["agnostic"])
            for b, o in enumerate(out):
                rec["%s|%d" % (case["name"], b)] = o.numpy() if o is not None else np.zeros((0, 6), np.float32)
        np.savez_compressed(os.path.join(OUT, "nms.npz"), **rec)
        print("nms fixture written")

    if "ap" in what:
        rec = {}
        for name, (preds, labels, shapes) in cases.ap_cases().items():
            import copy
            r = ref_metrics.compute_ap_lamr(copy.deepcopy(preds), copy.deepcopy(labels), shapes)
            rec[name + "|ap_lamr"] = np.array([r["ap"], r["lamr"]], np.float64)
            for k in ("recall", "precision", "fppi", "mr"):
                rec[name + "|" + k] = np.asarray(r[k], np.float64)
        np.savez_compressed(os.path.join(OUT, "ap.npz"), **rec)
        print("ap fixture written")

    if "step" in what:
        from oracle.model import OracleNet
        cfg = "config/kaist_dyolov4_fshare_global_concat_se3.cfg"
        defs = ref_parse.parse_model_cfg(cfg)
        onet = OracleNet(defs, cfg)
        sd = onet.synth_state(seed=0)
        m = ref_models.YOLO(cfg)
        m.load_state_dict(sd)
        hyp = cases.load_hyp("hyp.scratch.4")
        m.nc, m.hyp, m.gr = 1, hyp, 1.0
        m.train()
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=hyp["lr0"],
                               betas=(hyp["momentum"], 0.999), weight_decay=hyp["weight_decay"])
        losses = []
        for step in range(3):
            x, y, tg = cases.step_batch(step)
            pred = m(x, y)
            ld = ref_utils.compute_loss(pred, tg, m)
            loss = ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]
            losses.append([ld["box_loss"].item(), ld["obj_loss"].item(), ld["class_loss"].item()])
            opt.zero_grad()
            loss.backward()
            opt.step()
        sdn = m.state_dict()
        names = cases.step_probe_names()
        np.savez_compressed(os.path.join(OUT, "step.npz"), losses=np.array(losses, np.float64),
                            probes=np.array([[sdn[k].double().sum().item(), sdn[k].double().abs().sum().item()] for k in names]))
        print("step fixture written", losses)
