"""Round-3 fixture (second), produced by RUNNING THE REFERENCE in this container (tests/golden/ref_import.py):

  loss_focal.npz  compute_loss (build_utils/utils.py:209-293) with hyp['fl_gamma'] > 0, i.e. both BCE terms wrapped in the
                  reference's FocalLoss (:174-201, :236-238), on the seeded head tensors / targets of cases.focal_loss_cases():
                  the three loss terms and the gradient of their sum w.r.t. every head tensor.  Both shipped hyp files have
                  fl_gamma = 0, so the case overrides that one key.

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


if __name__ == "__main__":
    main()
