"""why does _fuse_late_reduces find nothing on the GPU?  prints the rejection reason of every BN_BWD_REDUCE"""
import collections, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd"), os.path.join(ROOT, "tests")]
MODE = sys.argv[1] if len(sys.argv) > 1 else "plain"
if MODE == "keep":
    os.environ["DYK_KEEP_DZ"] = "1"
os.environ["DYK_LATE_FUSE"] = "0"
import torch
from build_utils.parse_config import materialize_cfg
from models import YOLO
from dyk import lib as L, sched
m = YOLO(materialize_cfg("kaist_dyolov4_fshare_global_concat_se3")); m.dyk_dtype = "bf16"; m = m.cuda().train()
x = torch.rand(8, 3, 128, 160).cuda()
m(x, x)
plan = list(m.engine.plans.values())[0]
store = m.engine.store
mem = sched.Memory(plan, store)
cmds = plan.bwd
acc = [sched.accesses(op, d, mem, plan) for op, d in cmds]
why = collections.Counter()
for ri, (op, r) in enumerate(cmds):
    if op != L.OP_BN_BWD_REDUCE:
        continue
    es = 2
    target = mem.block(r.a, r.lda * es, r.C * es)
    if target is None:
        why["no target"] += 1; continue
    wi = None
    for j in range(ri - 1, -1, -1):
        if acc[j][2]:
            why["barrier op %d at %d" % (cmds[j][0], j)] += 1; break
        if any(w.overlaps(target) for w in acc[j][1]):
            wi = j; break
    if wi is None:
        continue
    wop, w = cmds[wi]
    if wop != L.OP_CONV:
        why["last writer op %d" % wop] += 1
    elif w.flags != L.EPI_ACCUM:
        why["flags %d" % w.flags] += 1
    elif w.ncls > 1 or w.dtype != r.dtype:
        why["ncls/dtype"] += 1
    elif w.y != r.a or w.ldy != r.lda or w.Cout != r.C or w.B * w.Ho * w.Wo != r.npix:
        why["geometry y %s a %s ldy %d lda %d C %d %d npix %d %d" % (w.y == r.a, 0, w.ldy, r.lda, w.Cout, r.C, w.B * w.Ho * w.Wo, r.npix)] += 1
    elif w.osy != 1 or w.osx != 1 or w.ooy or w.oox or w.Hg != w.Ho or w.Wg != w.Wo:
        why["strided"] += 1
    elif w.Cout % 8 or (w.ldy * es) % 16 or (r.ldb * es) % 16 or w.y % 16 or r.b % 16:
        why["alignment"] += 1
    else:
        why["fusable"] += 1
print(MODE, dict(why), "ops:", collections.Counter(op for op, _ in cmds).most_common(6))
