"""which BatchNorm-backward reduce launches are still commands of their own, and who writes their gradient last"""
import collections, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd"), os.path.join(ROOT, "tests")]
import torch
from build_utils.parse_config import materialize_cfg
from models import YOLO
from dyk import lib as L, sched
cfg = sys.argv[1] if len(sys.argv) > 1 else "kaist_dyolov4_fshare_global_concat_se3"
m = YOLO(materialize_cfg(cfg)); m.dyk_dtype = "bf16"; m = m.cuda().train()
x = torch.rand(4, 3, 128, 160).cuda()
m(x, x)
plan = list(m.engine.plans.values())[0]
store = m.engine.store
mem = sched.Memory(plan, store)
cmds = plan.bwd
names = {getattr(L, n): n for n in dir(L) if n.startswith("OP_")}
acc = [sched.accesses(op, d, mem, plan) for op, d in cmds]
why = collections.Counter()
for ri, (op, r) in enumerate(cmds):
    if op != L.OP_BN_BWD_REDUCE:
        continue
    target = mem.block(r.a, r.lda * 2, r.C * 2)
    writers = []
    for j in range(ri - 1, -1, -1):
        if target is not None and any(w.overlaps(target) for w in acc[j][1]):
            wop, w = cmds[j]
            tag = names.get(wop, str(wop))
            if wop == L.OP_CONV:
                tag += " taps%d flags%d Cout%d ncls%d full%s" % (w.ntaps, w.flags, w.Cout, w.ncls, w.Cout == r.C and w.y == r.a)
            writers.append(tag)
            if len(writers) == 3:
                break
    layer = max([l for c, l in plan.bwd_marks if c <= ri], default=-1) if os.environ.get("LAYERS") else -1
    lay = [l for c, l in plan.bwd_marks if c <= ri]
    why["C%d npix%d act%d <- %s%s" % (r.C, r.npix, r.act, " | ".join(writers), (" @layer %s" % (lay[-1] if lay else "?")) if os.environ.get("LAYERS") else "")] += 1
for k, v in sorted(why.items(), key=lambda kv: -kv[1]):
    print(v, k)
print("reduces left:", sum(why.values()), " ops:", collections.Counter(names.get(op, op) for op, _ in cmds).most_common(12))
