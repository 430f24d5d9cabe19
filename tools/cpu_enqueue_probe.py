"""How long does the host need to ENQUEUE one train step (no synchronisation), phase by phase?  If this is close to
the GPU step time the path is host-bound."""
import os, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd")]
import torch
import bench
from build_utils.parse_config import materialize_cfg
from build_utils.utils import compute_loss
from dyk.optim import FusedAdam
from models import YOLO
dev = torch.device("cuda", 0)
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    model = YOLO(materialize_cfg(bench.CFG))
model.nc, model.hyp, model.gr = 1, bench.load_hyp(), 1.0
model.dyk_dtype = "bf16"
model = model.to(dev).train()
BATCH = int(os.environ.get("PROBE_BATCH", "16"))
v8, l8, targets = bench.synth_batch(BATCH, 512, 640, 0, dev)
opt = FusedAdam(model, lr=1e-5, betas=(0.9, 0.999), weight_decay=5e-4)
def step(rec=None):
    t = [time.time()]
    v = v8.float() / 255.0; l = l8.float() / 255.0; t.append(time.time())
    pred = model(v, l); t.append(time.time())
    ld = compute_loss(pred, targets, model); loss = ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]; t.append(time.time())
    loss.backward(); t.append(time.time())
    opt.step(); t.append(time.time())
    if rec is not None: rec.append([b - a for a, b in zip(t, t[1:])])
for _ in range(5): step()
torch.cuda.synchronize()
rec = []
t0 = time.time()
for _ in range(10): step(rec)
t_enq = time.time() - t0
torch.cuda.synchronize()
t_all = time.time() - t0
import numpy as np
m = np.array(rec).mean(0) * 1e3
print("host enqueue per step: %.2f ms (input %.2f, forward %.2f, loss %.2f, backward %.2f, optimizer %.2f); GPU-complete per step: %.2f ms" % (t_enq / 10 * 1e3, *m, t_all / 10 * 1e3))
