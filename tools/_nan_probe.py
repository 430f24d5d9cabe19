import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd")]
import torch
import bench
from build_utils.parse_config import materialize_cfg
from build_utils.utils import compute_loss
from dyk.optim import FusedAdam
from models import YOLO
torch.manual_seed(0)
model = YOLO(materialize_cfg(bench.CFG)); model.nc, model.hyp, model.gr = 1, bench.load_hyp(), 1.0
model.dyk_dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
model = model.cuda().train()
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
v8, l8, t = bench.synth_batch(B, 512, 640, 0, "cuda")
opt = FusedAdam(model, lr=1e-3, betas=(0.937, 0.999), weight_decay=5e-4)
for step in range(12):
    pred = model(v8.float() / 255, l8.float() / 255)
    ld = compute_loss(pred, t, model)
    loss = ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]
    print("step", step, [float(x) for x in ld.values()], [bool(torch.isfinite(p).all()) for p in pred], flush=True)
    if step == 0:
        plan = list(model.engine.plans.values())[0]
        for i, (tr, rec) in enumerate(zip(plan.outs, plan.info)):
            if tr is None or rec["kind"] == "yolo": continue
            a = plan.arenas[tr.arena].tensor
            dt = torch.float32 if tr.esize == 4 else torch.bfloat16
            n = tr.npix * tr.ld
            flat = a[tr.off:tr.off + (n - (tr.ld - tr.C)) * tr.esize].view(dt)
            v = torch.as_strided(flat, (tr.npix, tr.C), (tr.ld, 1)).float()
            if not torch.isfinite(v).all():
                print("first non-finite layer output:", i, rec["kind"], "C", tr.C, "HxW", tr.H, tr.W); break
    loss.backward()
    g = model.engine.store.G
    fin = bool(torch.isfinite(g).all())
    print("   grad finite", fin, float(g.abs().max()))
    if not fin:
        st = model.engine.store
        bad = [e.name for e in st.entries if not torch.isfinite(g[e.offset:e.offset + e.numel]).all()]
        print("   non-finite grads in %d params; last (first produced in backward): %s ; first: %s" % (len(bad), bad[-3:], bad[:3]))
        print("   tuned:", {k: v for k, v in plan.tuned.items() if k[0] == "w"} )
        # which activation-gradient buffers are non-finite
        for i in range(len(plan.outs) - 1, -1, -1):
            tr = plan.outs[i]
            if tr is None or tr.tid not in plan.grads: continue
            gt = plan.grads[tr.tid]
            a = plan.arenas["grad"].tensor
            n = gt.npix * gt.ld
            flat = a[gt.off:gt.off + n * gt.esize].view(torch.bfloat16 if gt.esize == 2 else torch.float32)
            if not torch.isfinite(flat.float()).all():
                print("   last layer (in forward order) with non-finite dL/dout:", i, plan.info[i]["kind"], tr.C, tr.H, tr.W)
                break
        break
    opt.step()
