"""Round-2 golden fixtures, produced by RUNNING THE REFERENCE (/root/reference) in the build container:

    python tests/golden/make_golden_round2.py [boxes] [evalap] [weights]

  boxes.npz       xywh2xyxy / xyxy2xywh / clip_coords / scale_coords (both ratio_pad forms) / bbox_iou (IoU, GIoU,
                  DIoU, CIoU, both box formats) / box_iou / wh_iou of build_utils/utils.py:40-171 on seeded boxes
                  
  evalap.npz      the evaluation chain of evaluate.py:64-117 on seeded weights and images: YOLO eval forward ->
                  non_max_suppression(conf, 0.6, multi_label=False) -> scale_coords -> other_utils.metrics.
                  compute_ap_lamr, with ground truth derived from the reference's own detections (every other
                  detection, jittered) so that AP is far from 0 and 1
  weights_load.json  models.load_darknet_weights (models.py:318-364) on a formula-generated .weights stream:
                  per-tensor float64 sums of the resulting state_dict (incl. the quirk that only [convolutional]
                  sections consume weights and `cutoff`)
Inputs are rebuilt by the tests from the same seeded generators (functions below are imported by the tests).
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


# ----------------------------------------------------------------------------------------------- seeded inputs
def box_inputs():
    g = torch.Generator().manual_seed(11)
    n = 257
    xywh = torch.rand(n, 4, generator=g) * torch.tensor([640.0, 512.0, 200.0, 300.0]) + torch.tensor([0.0, 0.0, 1.0, 1.0])
    xywh[5, 2:] = 0.0                                  # a zero-area box
    xywh2 = xywh + (torch.rand(n, 4, generator=g) - 0.5) * torch.tensor([40.0, 40.0, 60.0, 60.0])
    xywh2[:, 2:] = xywh2[:, 2:].abs() + 0.5
    xywh2[7] = xywh[7]                                 # identical pair
    xyxy = torch.stack([xywh[:, 0] - xywh[:, 2] / 2, xywh[:, 1] - xywh[:, 3] / 2,
                        xywh[:, 0] + xywh[:, 2] / 2, xywh[:, 1] + xywh[:, 3] / 2], 1)
    xyxy2 = torch.stack([xywh2[:, 0] - xywh2[:, 2] / 2, xywh2[:, 1] - xywh2[:, 3] / 2,
                         xywh2[:, 0] + xywh2[:, 2] / 2, xywh2[:, 1] + xywh2[:, 3] / 2], 1)
    wild = (torch.rand(n, 6, generator=g) - 0.25) * 1000.0        # boxes far outside the image (clip / scale)
    wh1 = torch.rand(3, 2, generator=g) * 8 + 0.2
    wh2 = torch.rand(41, 2, generator=g) * 12 + 0.1
    return dict(xywh=xywh, xywh2=xywh2, xyxy=xyxy, xyxy2=xyxy2, wild=wild, wh1=wh1, wh2=wh2)


SCALE_CASES = [            # (img1_shape (h,w), img0_shape (h,w), ratio_pad or None)
    ((512, 640), (512, 640), None),
    ((416, 512), (512, 640), None),
    ((384, 640), (480, 720), None),
    ((128, 160), (512, 640), ((0.25, 0.25), (0.0, 0.0))),
    ((416, 512), (500, 353), ((0.832, 0.832), (109.0, 0.0))),
    ((320, 416), (1080, 1920), ((0.21666, 0.21666), (0.5, 43.0))),
]

EVAL_CFG = "kaist_dyolov4_fshare_global_concat_se3"
EVAL_B, EVAL_H, EVAL_W = 8, 128, 160
EVAL_SHAPES = [((512, 640), ((0.25, 0.25), (0.0, 0.0)))] * 6 + [((480, 640), ((0.25, 0.25), (0.0, 4.0)))] * 2


def eval_images():
    g = torch.Generator().manual_seed(21)
    v = torch.randint(0, 256, (EVAL_B, 3, EVAL_H, EVAL_W), dtype=torch.uint8, generator=g)
    l = torch.randint(0, 256, (EVAL_B, 3, EVAL_H, EVAL_W), dtype=torch.uint8, generator=g)
    return v, l


def weights_stream(n):
    """the float32 payload of the synthetic .weights file (a pure function of the element index)"""
    i = np.arange(n, dtype=np.float64)
    return (np.sin(i * 0.37) * 0.5 + np.cos(i * 0.011) * 0.25).astype(np.float32)


WEIGHTS_CASES = [("kaist_dyolov4_mobilenetv3_fshare_global_cse3", -1), ("kaist_yolov3", 12), ("kaist_dyolov3_add_sl", 80)]


def write_weights_file(path, n):
    with open(path, "wb") as f:
        np.array([0, 2, 5], dtype=np.int32).tofile(f)
        np.array([12345], dtype=np.int64).tofile(f)
        weights_stream(n).tofile(f)


# ----------------------------------------------------------------------------------------------- generators
def main():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    from ref_import import import_reference
    what = sys.argv[1:] or ["boxes", "evalap", "weights"]
    ref_models, ref_utils, ref_parse, ref_metrics = import_reference()
    OUT = HERE

    if "boxes" in what:
        x = box_inputs()
        rec = {}
        rec["xywh2xyxy"] = ref_utils.xywh2xyxy(x["xywh"].clone()).numpy()
        rec["xyxy2xywh"] = ref_utils.xyxy2xywh(x["xyxy"].clone()).numpy()
        rec["xywh2xyxy_np"] = ref_utils.xywh2xyxy(x["xywh"].numpy().copy())
        b = x["wild"][:, :4].clone()
        ref_utils.clip_coords(b, (512, 640))
        rec["clip"] = b.numpy()
        for ci, (s1, s0, rp) in enumerate(SCALE_CASES):
            c = x["wild"].clone()
            out = ref_utils.scale_coords(s1, c, s0, rp)
            assert out is c
            rec["scale%d" % ci] = c.numpy()
        for fmt in (True, False):
            b1 = (x["xyxy"] if fmt else x["xywh"]).t().clone()
            b2 = (x["xyxy2"] if fmt else x["xywh2"]).clone()
            for mode in ("IoU", "GIoU", "DIoU", "CIoU"):
                kw = {} if mode == "IoU" else {mode: True}
                rec["bbox_iou|%d|%s" % (int(fmt), mode)] = ref_utils.bbox_iou(b1, b2, x1y1x2y2=fmt, **kw).numpy()
        rec["box_iou"] = ref_utils.box_iou(x["xyxy"][:33].clone(), x["xyxy2"][:57].clone()).numpy()
        rec["wh_iou"] = ref_utils.wh_iou(x["wh1"].clone(), x["wh2"].clone()).numpy()
        np.savez_compressed(os.path.join(OUT, "boxes.npz"), **rec)
        print("boxes fixture written (%d arrays)" % len(rec))

    if "evalap" in what:
        from oracle.model import OracleNet
        cfg = "config/%s.cfg" % EVAL_CFG
        defs = ref_parse.parse_model_cfg(cfg)
        sd = OracleNet(defs, cfg).synth_state(3)
        torch.manual_seed(0)
        m = ref_models.YOLO(cfg, (EVAL_H, EVAL_W))
        m.load_state_dict(sd)
        v8, l8 = eval_images()
        # calibrate the BatchNorm running statistics on the batch (momentum 1: running = batch statistics), as a trained
        # network's would be: with random running statistics the activations grow by 1e5 through the 100+ layers and
        # every rounding difference is amplified with them.  The calibrated statistics are stored in the fixture.
        m.train()
        bns = [mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
        for mod in bns:
            mod.momentum = 1.0
        with torch.no_grad():
            m(v8.float() / 255.0, l8.float() / 255.0)
        m.eval()
        calib = {k: v.clone() for k, v in m.state_dict().items() if k.endswith(("running_mean", "running_var"))}
        sd.update(calib)
        # random weights saturate the heads (every score 0 or 1): rescale the three head convs so that the logits have
        # a standard deviation of 1.5 around their bias (the factors are stored; tests apply them to synth_state(3))
        with torch.no_grad():
            raw = m(v8.float() / 255.0, l8.float() / 255.0)[1]
        head_scale = []
        for j, p_ in zip(m.yolo_layers, raw):
            k = "module_list.%d.Conv2d.weight" % (j - 1)
            b = sd["module_list.%d.Conv2d.bias" % (j - 1)].view(1, p_.shape[1], 1, 1, -1)
            f = 1.5 / float((p_ - b).std())
            head_scale.append(f)
            sd[k] = sd[k] * f
        m.load_state_dict(sd)
        with torch.no_grad():
            pred = m(v8.float() / 255.0, l8.float() / 255.0)[0]
        # untrained weights: choose the confidence threshold so that ~60 candidates per image survive
        sc = pred[..., 4] * pred[..., 5:].max(-1).values            # what the NMS thresholds (utils.py:408-409,416)
        ok = ((pred[..., 2:4] > 2) & (pred[..., 2:4] < 4096)).all(-1)
        conf = float(min((sc[b][ok[b]].sort().values[-60]) for b in range(EVAL_B)))
        dets = ref_utils.non_max_suppression(pred, conf_thres=conf, iou_thres=0.6, multi_label=False)
        preds, per_image = [], []
        for idx, p in enumerate(dets):
            assert p is not None
            boxes = ref_utils.scale_coords((EVAL_H, EVAL_W), p[:, :4].clone(), EVAL_SHAPES[idx][0], EVAL_SHAPES[idx][1])
            per_image.append(torch.cat([boxes, p[:, 4:6]], 1).numpy())
            for i in range(p.shape[0]):
                preds.append({"img_id": idx, "conf": p[i, 4].item(), "bbox": boxes[i].numpy()})
        # ground truth: every other detection of the reference (jittered by a few pixels), plus one box nothing detects
        gj = torch.Generator().manual_seed(5)
        labels, shapes = [], []
        for idx, d in enumerate(per_image):
            h0, w0 = EVAL_SHAPES[idx][0]
            gt = torch.from_numpy(d[::2, :4]).clone()
            gt += (torch.rand(gt.shape, generator=gj) - 0.5) * 6.0
            gt = torch.cat([gt, torch.tensor([[5.0, 7.0, 45.0, 99.0]])], 0)
            gt[:, [0, 2]] = gt[:, [0, 2]].clamp(0, w0)
            gt[:, [1, 3]] = gt[:, [1, 3]].clamp(0, h0)
            xc, yc = (gt[:, 0] + gt[:, 2]) / 2 / w0, (gt[:, 1] + gt[:, 3]) / 2 / h0
            bw, bh = (gt[:, 2] - gt[:, 0]) / w0, (gt[:, 3] - gt[:, 1]) / h0
            labels.append(torch.stack([torch.zeros_like(xc), xc, yc, bw, bh], 1).numpy().astype(np.float32))
            shapes.append((w0, h0))                # kaist_dataset: shapes are (w, h) (metrics.py:94-95 scales x by [0])
        preds.sort(key=lambda q: float(q["conf"]), reverse=True)
        res = ref_metrics.compute_ap_lamr(preds, [lb.copy() for lb in labels], np.array(shapes))
        rec = {"head_scale": np.array(head_scale, dtype=np.float64), "conf": np.float64(conf), "io": pred.numpy().astype(np.float32), "ap": np.float64(res["ap"]), "lamr": np.float64(res["lamr"]),
               "recall": np.asarray(res["recall"]), "precision": np.asarray(res["precision"]),
               "shapes": np.array(shapes, dtype=np.int64), "ndet": np.array([d.shape[0] for d in per_image])}
        for k, v in calib.items():
            rec["bn|" + k] = v.numpy()
        for idx in range(EVAL_B):
            rec["det%d" % idx] = per_image[idx]
            rec["labels%d" % idx] = labels[idx]
        np.savez_compressed(os.path.join(OUT, "evalap.npz"), **rec)
        print("evalap fixture: conf %.6g, %d detections, AP %.4f, LAMR %.4f" % (conf, len(preds), res["ap"], res["lamr"]))

    if "weights" in what:
        rec = {}
        for cfg_name, cutoff in WEIGHTS_CASES:
            cfg = "config/%s.cfg" % cfg_name
            torch.manual_seed(1)
            m = ref_models.YOLO(cfg)
            n = sum(p.numel() for p in m.state_dict().values())
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "synthetic.weights")
                write_weights_file(path, n)
                before = {k: v.clone() for k, v in m.state_dict().items()}
                ref_models.load_darknet_weights(m, path, cutoff)
            after = m.state_dict()
            changed = [k for k in after if not torch.equal(after[k], before[k])]
            rec["%s|%d" % (cfg_name, cutoff)] = {
                "n_stream": int(n), "changed": changed,
                "sums": {k: float(after[k].double().sum()) for k in changed},
                "first": {k: float(after[k].reshape(-1)[0]) for k in changed},
                "version": [int(q) for q in m.version], "seen": int(m.seen[0])}
            print(cfg_name, cutoff, "tensors loaded:", len(changed))
        with open(os.path.join(OUT, "weights_load.json"), "w") as f:
            json.dump(rec, f)


if __name__ == "__main__":
    main()
