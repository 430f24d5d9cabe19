"""N > 1 path on CPU: two processes over gloo run the bucketed gradient exchange of dyk.ddp on a
dry-compiled plan.  The native backward is replaced by a stand-in that fills each finished gradient
segment with a rank-dependent value, so the test checks the host logic: the segments partition both
the command list and the flat gradient buffer, every bucket is reduced exactly once, asynchronously."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from build_utils.parse_config import materialize_cfg
        from dyk.ddp import GradAllReduce
        from dyk.plan import compile_plan
        from models import YOLO
        torch.manual_seed(0)
        m = YOLO(materialize_cfg("kaist_yolov3"))
        eng = m.engine
        eng.store.adopt(torch.device("cpu"))
        plan = compile_plan(m, eng.store, 1, 64, 64, torch.bfloat16, True, torch.device("cpu"), dry=True)
        red = GradAllReduce(m, dist, n_buckets=6)
        segs = red.segments(plan)
        # partition checks
        assert segs[0][0] == 0 and segs[-1][1] == len(plan.bwd)
        assert all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
        assert segs[0][3] == eng.store.total and segs[-1][2] == 0
        assert all(a[2] == b[3] for a, b in zip(segs, segs[1:]))
        assert 2 <= len(segs) <= 8
        # the default: geometric buckets (50 | 30 | 15 | 4 | 1 % of the gradient buffer in backward order) -- the same
        # partition properties, and the bucket that is exchanged after the last layer is a small one
        geo = GradAllReduce(m, dist).segments(plan)
        assert geo[0][0] == 0 and geo[-1][1] == len(plan.bwd) and geo[0][3] == eng.store.total and geo[-1][2] == 0
        assert all(a[1] == b[0] and a[2] == b[3] for a, b in zip(geo, geo[1:]))
        assert 2 <= len(geo) <= 5 and geo[-1][3] - geo[-1][2] <= 0.2 * eng.store.total
        assert geo[0][3] - geo[0][2] >= 0.45 * eng.store.total
        red = GradAllReduce(m, dist, n_buckets=6)
        G = eng.store.G
        G.fill_(-7.0)
        ran = []
        plan.run = lambda which, stream, c0=0, c1=None: ran.append((c0, c1))     # stand-in for the native backward
        stream = 0
        for (c0, c1, lo, hi) in segs:
            plan.run("bwd", stream, c0, c1)
            G[lo:hi] = float(rank + 1)                     # "gradient" of this segment on this rank
            red.bucket_ready(lo, hi)
        red.all_reduce()
        ok = bool((G == float(sum(range(1, world + 1)))).all())
        # the harness's other exchanges: per-step loss reduction (distributed_utils.py:117-142), object gather (:74-114)
        # and the padded detection gather of the sharded evaluation
        from dyk.ddp import all_gather, gather_detections, get_rank, get_world_size, reduce_dict
        assert get_world_size() == world and get_rank() == rank
        ld = {"obj_loss": torch.tensor([2.0 * (rank + 1)]), "box_loss": torch.tensor([1.0 * (rank + 1)]),
              "class_loss": torch.tensor([0.0])}
        red_ld = reduce_dict(ld)
        ok = ok and list(red_ld.keys()) == ["box_loss", "class_loss", "obj_loss"]
        ok = ok and float(red_ld["box_loss"]) == 1.5 and float(red_ld["obj_loss"]) == 3.0 and red_ld["box_loss"].shape == (1,)
        ok = ok and float(reduce_dict(ld, average=False)["obj_loss"]) == 6.0
        objs = all_gather({"rank": rank, "blob": list(range(rank * 100))})
        ok = ok and [o["rank"] for o in objs] == [0, 1] and len(objs[1]["blob"]) == 100
        dets = [torch.full((2 + rank, 6), float(rank)), None]          # rank 0: 2 rows, rank 1: 3 rows, one empty image each
        rows = gather_detections(dets, [10 * rank, 10 * rank + 1])
        ok = ok and tuple(rows.shape) == (5, 7) and rows[:, 0].tolist() == [0.0, 0.0, 10.0, 10.0, 10.0]
        ok = ok and rows[:2, 1:].eq(0).all().item() and rows[2:, 1:].eq(1).all().item()
        q.put((rank, ok, len(segs), ran == [(a, b) for a, b, _, _ in segs]))
    finally:
        dist.destroy_process_group()


def test_bucketed_gradient_allreduce_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), "all-reduced gradient buffer is wrong"
    assert all(r[3] for r in res)
