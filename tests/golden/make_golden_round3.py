"""Round-3 fixture, produced by RUNNING THE REFERENCE in this container (tests/golden/ref_import.py):

  step_sgd.npz  three optimizer steps of the target cfg with the reference's own SGD branch
                (train.py:86-89: optim.SGD(pg, lr=hyp['lr0'], momentum=hyp['momentum'], nesterov=True), weight decay
                hyp['weight_decay']) on the seeded batches of cases.step_batch: the three losses of every step, and
                per probed parameter (sum, sum|.|) plus the norm of its total update.

VERDICT r2 weak #2: the Adam fixture (step.npz) can only be held at 15 % on steps 2-3 -- Adam's first steps move every
weight by ~lr*sign(g), so gradient entries at rounding-noise level decide their direction.  With SGD the update is
proportional to g: rounding-level differences in g stay rounding-level in the parameters, and the trajectory can be
pinned tightly (tests/test_gpu_model.py::test_three_sgd_steps_match_reference).

    python tests/golden/make_golden_round3.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
from ref_import import import_reference  # noqa: E402

OUT = HERE


def main():
    torch.set_num_threads(8)
    ref_models, ref_utils, ref_parse, _ = import_reference()
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
    p0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    # lr0 of the hyp file (1e-3) moves the first conv's weights by half their norm in three steps on this random-weight net
    # (gradient norms ~1e3): the trajectory is then chaotic in the reference's own fp32 arithmetic (measured: HIP fp32 vs
    # reference 0.7 % / 20 % on the box / objectness terms of steps 2-3, step 1 at 2e-5).  LR keeps the same optimizer
    # branch in its linear regime, where a wrong gradient, momentum or decay term still shows at first order in every
    # parameter delta while rounding-level gradient differences stay rounding-level.
    LR = 1e-5
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=LR, momentum=hyp["momentum"],
                          weight_decay=hyp["weight_decay"], nesterov=True)
    losses = []
    for step in range(3):
        x, y, tg = cases.sgd_step_batch(step)
        pred = m(x, y)
        ld = ref_utils.compute_loss(pred, tg, m)
        loss = ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]
        losses.append([ld["box_loss"].item(), ld["obj_loss"].item(), ld["class_loss"].item()])
        opt.zero_grad()
        loss.backward()
        opt.step()
    sdn = m.state_dict()
    names = cases.step_probe_names()
    np.savez_compressed(
        os.path.join(OUT, "step_sgd.npz"), losses=np.array(losses, np.float64),
        probes=np.array([[sdn[k].double().sum().item(), sdn[k].double().abs().sum().item()] for k in names]),
        delta=np.array([[(sdn[k].double() - p0[k].double()).norm().item(), (sdn[k].double() - p0[k].double()).sum().item(),
                         ((sdn[k].double() - p0[k].double()) * p0[k].double().sign()).sum().item()] for k in names]),
        lr=np.array([LR, hyp["momentum"], hyp["weight_decay"]]))
    print("SGD step fixture written", losses)


if __name__ == "__main__":
    main()
