"""Round-3 fixture (second), produced by RUNNING THE REFERENCE in this container (tests/golden/ref_import.py):

  loss_focal.npz  compute_loss (build_utils/utils.py:209-293) with hyp['fl_gamma'] > 0, i.e. both BCE terms wrapped in the
                  reference's FocalLoss (:174-201, :236-238), on the seeded head tensors / targets of cases.focal_loss_cases():
                  the three loss terms and the gradient of their sum w.r.t. every head tensor.  Both shipped hyp files have
                  fl_gamma = 0, so the case overrides that one key.

  mismatch.npz    the reference's YOLO on tests/golden/tiny_kaist_mismatch.cfg (builder-authored: [shortcut] sections between
                  tensors of different channel counts, build_utils/layers.py:78-83 -- 32 + 64 -> 32, weighted 64 + 32 -> 64,
                  48 + 64 -> 48): eval outputs, train-mode outputs, and the gradient checksums of a scalar functional of them.

    python tests/golden/make_golden_round3b.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
from make_golden_loss import fake_model  # noqa: E402
from ref_import import import_reference  # noqa: E402

OUT = HERE


def main():
    ref_models, ref_utils, _, _ = import_reference()
    rec = {}
    for case in cases.focal_loss_cases():
        hyp = dict(cases.load_hyp(case["hyp"]))
        hyp["fl_gamma"] = case["fl_gamma"]
        model = fake_model(ref_models, case["cfg"], case["nc"], hyp, case["gr"], (case["H"], case["W"]))
        p = cases.loss_preds(case)
        for t in p:
            t.requires_grad_(True)
        out = ref_utils.compute_loss(p, cases.loss_targets(case), model)
        (out["box_loss"] + out["obj_loss"] + out["class_loss"]).backward()
        k = case["name"] + "|"
        rec[k + "losses"] = np.array([out["box_loss"].item(), out["obj_loss"].item(), out["class_loss"].item()], np.float32)
        for i, t in enumerate(p):
            rec[k + "dp%d" % i] = t.grad.numpy()
        print(case["name"], rec[k + "losses"])
    np.savez_compressed(os.path.join(OUT, "loss_focal.npz"), **rec)
    print("loss_focal fixture written")

    from oracle.model import OracleNet
    _, _, ref_parse, _ = import_reference()
    cfg = os.path.join(HERE, "tiny_kaist_mismatch.cfg")
    onet = OracleNet(ref_parse.parse_model_cfg(cfg), cfg)
    sd = onet.synth_state(seed=3)
    m = ref_models.YOLO(cfg)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    with torch.no_grad():
        m.module_list[5].w.copy_(torch.tensor([0.3, -0.8]))           # (synth_state leaves the fusion weights at 0)
    x, y = cases.mismatch_inputs()
    rec = {}
    m.eval()
    with torch.no_grad():
        io, p = m(x, y)
    rec["eval_io"], rec["eval_p0"] = io.numpy(), p[0].numpy()
    m.train()
    out = m(x, y)
    rec["train_p0"] = out[0].detach().numpy()
    loss = sum((t ** 2).mean() for t in out)
    loss.backward()
    rec["train_loss"] = np.float32(loss.item())
    names = [k for k, _ in m.named_parameters()]
    rec["grad_sums"] = np.asarray([[q.grad.abs().sum().item(), q.grad.sum().item()] for _, q in m.named_parameters()], np.float64)
    rec["grad_w5"] = m.module_list[5].w.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "mismatch.npz"), **rec)
    print("mismatch fixture written: loss", loss.item(), len(names), "parameters")


if __name__ == "__main__":
    main()
