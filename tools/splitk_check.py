"""analysis (round 5): gradients and a three-step SGD trajectory of the target cfg (fp32 and bf16) with split-K convolutions on
(default) and off (DYK_CONV_SPLITK=0), each in its own process: per-step losses, gradient difference of every step relative to
the gradient norm, twice per variant (run-to-run reproducibility)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path[:0] = [%r, %r, %r, %r]
import numpy as np, torch
import cases
from helpers import hyp, C3
from build_utils.parse_config import materialize_cfg
from build_utils.utils import compute_loss
from dyk.optim import FusedSGD
from models import YOLO
dtype, out = sys.argv[1], sys.argv[2]
torch.manual_seed(0)
m = YOLO(materialize_cfg(C3))
m.dyk_dtype = dtype
m = m.cuda().train()
h = hyp("hyp.scratch.4")
m.nc, m.hyp, m.gr = 1, h, 1.0
opt = FusedSGD(m, lr=1e-5, momentum=h["momentum"], weight_decay=h["weight_decay"], nesterov=True)
opt.zero_in_step = False
res = {}
for step in range(3):
    x, y, tg = cases.sgd_step_batch(step)
    opt.zero_grad()
    m.engine.store.G.zero_() if m.engine.store.G is not None else None
    pred = m(x.cuda(), y.cuda())
    ld = compute_loss(pred, tg.cuda(), m)
    (ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]).backward()
    res["loss%%d" %% step] = np.array([ld["box_loss"].item(), ld["obj_loss"].item()])
    res["g%%d" %% step] = m.engine.store.G.detach().cpu().numpy().copy()
    res["p%%d" %% step] = [p.detach().cpu().numpy().copy() for p in pred][2]
    opt.step()
    m.engine.store.G.zero_()
ns = sum(1 for op, d in list(m.engine.plans.values())[0].fwd + list(m.engine.plans.values())[0].bwd if op == 1 and d.splitk > 1)
print("split convs in plan:", ns)
np.savez(out, **res)
''' % (ROOT, os.path.join(ROOT, "double-yolo-kaist_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"))


def run(dtype, split, tag):
    env = dict(os.environ)
    env["DYK_CONV_SPLITK"] = split
    out = "/tmp/sk_%s_%s_%s.npz" % (dtype, split, tag)
    r = subprocess.run([sys.executable, "-c", CHILD, dtype, out], env=env, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-2000:])
        raise SystemExit(1)
    print(dtype, "split", split, tag, r.stdout.strip().splitlines()[-1])
    return np.load(out)


for dtype in ("fp32", "bf16"):
    a0, a1, b0, b1 = run(dtype, "0", "a"), run(dtype, "0", "b"), run(dtype, "1", "a"), run(dtype, "1", "b")
    for s in range(3):
        n = np.linalg.norm(a0["g%d" % s])
        print("%s step %d: loss off %s on %s | |g_on - g_off| / |g| = %.2e | run-to-run off %.1e on %.1e | head diff %.2e"
              % (dtype, s, a0["loss%d" % s], b0["loss%d" % s], np.linalg.norm(b0["g%d" % s] - a0["g%d" % s]) / n,
                 np.linalg.norm(a1["g%d" % s] - a0["g%d" % s]) / n, np.linalg.norm(b1["g%d" % s] - b0["g%d" % s]) / n,
                 np.abs(b0["p%d" % s] - a0["p%d" % s]).max() / np.abs(a0["p%d" % s]).max()))
