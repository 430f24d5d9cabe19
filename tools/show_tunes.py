"""Print the autotuned tile configuration of every conv / weight-gradient problem of the target cfg's training plan."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "double-yolo-kaist_amd"))
import torch
from build_utils.parse_config import materialize_cfg
from models import YOLO
m = YOLO(materialize_cfg("kaist_dyolov4_fshare_global_concat_se3")).cuda().train()
m.dyk_dtype = "bf16"
B = int(os.environ.get("B", "16"))
x = torch.rand(B, 3, 512, 640, device="cuda")
out = m(x, x)
plan = list(m.engine.plans.values())[0]
for k, v in sorted(plan.tuned.items(), key=lambda kv: str(kv[0])):
    if k[0] == "c":
        print("conv  Cin %4d Cout %4d %3dx%-3d taps %2d isy %d osy %d flags %2d ncls %d -> bkb %3d pipe %d tile %d bm %d" % (
            k[3], k[4], k[5], k[6], k[7], k[8], k[9], k[10], k[11], v & 0xff, (v >> 8) & 0xf, (v >> 12) & 0xf, (v >> 24) & 0xf) + (" KG" if (v >> 28) & 7 else ""))
    else:
        v, sp = v if isinstance(v, tuple) else (v, 0)
        print("splits %3s " % (sp or "auto"), end="")
        print("wgrad Cin %4d Cout %4d %3dx%-3d taps %2d isy %d -> stages %d kg %d cap %d mt %d" % (
            k[3], k[4], k[5], k[6], k[7], k[8], v & 0xff, (v >> 8) & 0xff, (v >> 24) & 0xf, (v >> 28) & 7))
