"""Where does the host block while it enqueues a train step?  Times Plan.run per command-list call and the Python around it."""
import os, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd")]
import torch
import bench
from build_utils.parse_config import materialize_cfg
from build_utils.utils import compute_loss
from dyk.optim import FusedAdam
from dyk import plan as P
from models import YOLO
dev = torch.device("cuda", 0)
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    model = YOLO(materialize_cfg(bench.CFG))
model.nc, model.hyp, model.gr = 1, bench.load_hyp(), 1.0
model.dyk_dtype = "bf16"
model = model.to(dev).train()
v8, l8, targets = bench.synth_batch(16, 512, 640, 0, dev)
opt = FusedAdam(model, lr=1e-5, betas=(0.9, 0.999), weight_decay=5e-4)
calls = []
orig = P.Plan.run
def timed(self, which, stream, *a):
    t0 = time.time(); r = orig(self, which, stream, *a); calls.append((which, a, time.time() - t0)); return r
P.Plan.run = timed
def step():
    pred = model(v8, l8)
    ld = compute_loss(pred, targets, model); loss = ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]
    loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
calls.clear()
t0 = time.time()
marks = []
for _ in range(6):
    step(); marks.append(time.time() - t0)
torch.cuda.synchronize()
print("step enqueue ends at (ms):", ["%.1f" % (1e3 * m) for m in marks], "all done %.1f" % (1e3 * (time.time() - t0)))
for w, a, dt in calls:
    print("  run %-4s %-14s %.2f ms" % (w, a, 1e3 * dt))
