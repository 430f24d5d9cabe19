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


def _worker_real_plan(rank, world, port, q):
    """the REAL backward command list of the dual-stream target cfg (dry plan: resolved descriptors, no kernels): every
    command's writes into the flat gradient buffer are read off its descriptor (dyk/sched.py access sets) and replayed
    symbolically -- each write adds (rank + 1) to the bytes it covers -- with the all-reduce of a bucket issued exactly
    where dyk.ddp enqueues it.  A cut that closes a bucket before its last writer has run leaves a value that is not
    (number of writers) x sum(rank + 1): caught here without hardware."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from build_utils.parse_config import materialize_cfg
        from dyk import lib as L, sched
        from dyk.ddp import GradAllReduce
        from dyk.plan import compile_plan
        from models import YOLO
        torch.manual_seed(0)
        m = YOLO(materialize_cfg("kaist_dyolov4_fshare_global_concat_se3"))
        eng = m.engine
        eng.store.adopt(torch.device("cpu"))
        plan = compile_plan(m, eng.store, 2, 64, 96, torch.bfloat16, True, torch.device("cpu"), dry=True)
        mem = sched.Memory(plan, eng.store)
        G = eng.store.G
        g0 = G.data_ptr()
        writes = []                                   # per command: [(first float, last float + 1)] inside G
        for op, d in plan.bwd:
            _, W, barrier = sched.accesses(op, d, mem, plan)
            assert not barrier or op == L.OP_MEMSET
            writes.append([(r.lo // 4, (r.hi + 3) // 4) for r in W if r.key == "G"])
        nwrites = torch.zeros(eng.store.total)
        for ws in writes:
            for lo, hi in ws:
                nwrites[lo:hi] += 1
        assert float((nwrites > 0).float().mean()) > 0.99, "every parameter of the net gets a gradient"
        ok, nseg = True, []
        for nb, pair in ((0, "0"), (6, "0"), (0, "1")):
            os.environ["DYK_PAIR"] = pair
            plan.__dict__.pop("_ddp_segs", None)
            red = GradAllReduce(m, dist, n_buckets=nb)
            segs = red.segments(plan)
            nseg.append(len(segs))
            assert segs[0][0] == 0 and segs[-1][1] == len(plan.bwd) and segs[0][3] == eng.store.total and segs[-1][2] == 0
            assert all(a[1] == b[0] and a[2] == b[3] for a, b in zip(segs, segs[1:]))
            # static form of the check: nobody writes into a bucket after it was handed to the exchange
            for (c0, c1, lo, hi) in segs:
                for c in range(c1, len(plan.bwd)):
                    assert all(w1 <= lo or w0 >= hi for (w0, w1) in writes[c]), (
                        "command %d (op %d) writes into bucket [%d, %d) closed at command %d" % (c, plan.bwd[c][0], lo, hi, c1))
            # dynamic form, through the real collective
            G.zero_()
            for (c0, c1, lo, hi) in segs:
                for c in range(c0, c1):
                    for w0, w1 in writes[c]:
                        G[w0:w1] += float(rank + 1)
                red.bucket_ready(lo, hi)
            red.all_reduce()
            want = nwrites * float(sum(range(1, world + 1)))
            ok = ok and bool(torch.equal(G, want))
            if pair == "1":
                # two-problem launches: no bucket closes between the backward of a section and that of its twin
                marks = {cnt: layer for cnt, layer in plan.bwd_marks}
                for (c0, c1, lo, hi) in segs[:-1]:
                    done = min(l for cnt, l in plan.bwd_marks if cnt <= c1 and l >= 0) if any(cnt <= c1 for cnt, _ in plan.bwd_marks) else None
                    for l, t in plan.twin_layer.items():
                        if l < t and done is not None:
                            assert not (l < done <= t) or c1 == len(plan.bwd), ("cut inside twin span", l, t, done)
        q.put((rank, ok, nseg))
    finally:
        dist.destroy_process_group()


def test_bucket_cuts_against_the_real_backward_list_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30100 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker_real_plan, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), "a gradient bucket was exchanged before its last writer ran"
    assert all(2 <= n <= 8 for r in res for n in r[2]), res
