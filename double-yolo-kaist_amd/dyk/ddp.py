"""Data-parallel gradient exchange for one-process-per-GPU training (`torch.distributed`, backend "nccl"
= RCCL over xGMI on ROCm).  The reference is single-GPU (build_utils/torch_utils.py:35-36); this is the
north-star's added exchange step: one all-reduce(SUM) of the trainable gradients per optimizer step.

All gradients live in ONE flat fp32 buffer in layer order (dyk/params.py), and the backward command list
runs from the last layer to the first, so the tail of the buffer is final first.  The backward is cut into
a few segments; as soon as a segment's kernels are enqueued, the matching contiguous slice of the buffer
is all-reduced asynchronously (RCCL's stream waits on the compute stream at enqueue time), overlapping
the exchange of late-layer gradients with the differentiation of early layers.  xGMI is point-to-point, so
buckets are few and geometric (default 5: 50 | 30 | 15 | 4 | 1 % of the 464 MB of the target model, late layers
first -- the deep layers hold the parameters, the early layers the time of a backward pass): large, bandwidth-
bound collectives early, a small exposed tail.  The 1/world_size averaging is folded into the fused
optimizer step (FusedAdam.grad_scale).
"""
import os
import pickle

import torch


def _dist():
    import torch.distributed as dist
    return dist


def is_dist_avail_and_initialized():
    dist = _dist()
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return _dist().get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return _dist().get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def _comm_device():
    """where collective payloads live: the rank's GPU under nccl (= RCCL), host memory under gloo"""
    return torch.device("cuda", torch.cuda.current_device()) if _dist().get_backend() == "nccl" else torch.device("cpu")


def reduce_dict(input_dict, average=True):
    """The per-step loss exchange of the reference harness (train_utils/distributed_utils.py:117-142, called at
    kaist_train_eval_utils.py:82): the values of `input_dict` (the three [1]-shaped loss terms: 12 bytes) are
    stacked in key order, all-reduced and, with `average`, divided by the world size.  One rank: returned as is."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        _dist().all_reduce(values)
        if average:
            values /= world
        return {k: v for k, v in zip(names, values)}


def all_gather(data):
    """list over ranks of an arbitrary picklable object (train_utils/distributed_utils.py:74-114): sizes first,
    then the pickles padded to the longest."""
    world = get_world_size()
    if world == 1:
        return [data]
    dist, dev = _dist(), _comm_device()
    payload = torch.frombuffer(bytearray(pickle.dumps(data)), dtype=torch.uint8).to(dev)
    size = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    buf = torch.zeros(max(sizes), dtype=torch.uint8, device=dev)
    buf[:payload.numel()] = payload
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return [pickle.loads(t.cpu().numpy().tobytes()[:n]) for n, t in zip(sizes, parts)]


def gather_detections(dets, image_ids):
    """Evaluation shards images over ranks; every rank gets all detections as one float32 tensor of rows
    (image_id, x1, y1, x2, y2, conf, cls) in rank order, without pickling: the per-rank row counts are exchanged
    first, then the rows padded to the longest shard.  `dets`: the list non_max_suppression returns (tensor [n,6]
    or None per image); `image_ids`: the dataset index of each image of this rank's batch."""
    rows = []
    for d, i in zip(dets, image_ids):
        if d is not None and d.shape[0]:
            rows.append(torch.cat([torch.full((d.shape[0], 1), float(i), dtype=torch.float32, device=d.device), d.float()], 1))
    world = get_world_size()
    if world == 1:
        if rows:
            return torch.cat(rows, 0)
        dev = next((d.device for d in dets if d is not None), torch.device("cpu"))
        return torch.zeros((0, 7), device=dev)   # (empty, but on the device the detections live on)
    dist, dev = _dist(), _comm_device()
    mine = torch.cat(rows, 0).to(dev) if rows else torch.zeros((0, 7), device=dev)
    n = torch.tensor([mine.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    pad = torch.zeros((max(max(counts), 1), 7), dtype=torch.float32, device=dev)
    pad[:mine.shape[0]] = mine
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([t[:c] for c, t in zip(counts, parts)], 0)


class GradAllReduce:
    # cumulative fractions of the gradient buffer (in backward order: deep layers first) at which a bucket is closed.
    # Geometric, not uniform: the deep layers hold the parameters, the early layers the TIME of a backward pass -- with
    # equal buckets the last one spans most of the pass and its all-reduce starts when nothing is left to hide it.
    # 50 | 30 | 15 | 4 | 1 %: the exchange exposed after the last layer is ~1 % of the gradients (4.6 MB of 464 MB on
    # the target cfg), and five segments cost the one-rank step +0.15..0.3 ms where eight equal ones cost +0.9 ms.
    GEOMETRIC = (0.5, 0.8, 0.95, 0.99)

    def __init__(self, model, dist, n_buckets=None, attach=True):
        if n_buckets is None:
            n_buckets = int(os.environ.get("DYK_DDP_BUCKETS", "0"))       # 0: the geometric cuts above; n: n equal buckets
        self.model, self.dist, self.n_buckets = model, dist, n_buckets
        self.engine = model.engine
        if attach:                          # (attach=False: only the segment arithmetic is wanted, dyk/optim.py)
            self.engine.grad_sync = self
        self._works = []

    def segments(self, plan, fractions=None):
        """[(c0, c1, lo, hi)]: backward commands [c0, c1) complete the gradients G[lo, hi).  fractions: cumulative fractions of the
        gradient buffer at which a segment closes (default: this object's buckets); dyk/optim.py asks for one cut at 95 % to
        start the optimizer on the deep layers while the early layers are still being differentiated."""
        # cached on the plan object itself: a plan evicted under multi-scale training takes its cuts with it (an
        # id()-keyed table could hand them to a later plan that re-uses the address)
        ckey = self.n_buckets if fractions is None else tuple(fractions)
        segs = plan.__dict__.get("_ddp_segs", {}).get(ckey)
        if segs is None:
            store = self.engine.store
            total = store.total
            # first parameter offset of every layer (entries are in layer order)
            first_off = {}
            for e in store.entries:
                first_off.setdefault(e.layer, e.offset)
            if fractions is not None:
                cuts = [total - int(f * total) for f in fractions]
            elif self.n_buckets > 0:
                cuts = [total - (q * total) // self.n_buckets for q in range(1, self.n_buckets)]
            else:
                cuts = [total - int(f * total) for f in self.GEOMETRIC]
            cuts.append(-1)                            # `lo` at or below cuts[0] closes the open bucket
            segs = []
            c_prev, hi = 0, total
            marks = plan.bwd_marks                    # (commands emitted before layer i's backward, i), descending i
            # twin sections (dyk/twins.py) run as two-problem launches: a cut between the backward of section t and that
            # of its twin l < t would leave both halves unpaired, so no bucket closes at a layer in (l, t]
            # (only when the BACKWARD schedule really pairs launches: DYK_PAIR_WHICH=fwd / DYK_PAIR_OPS=none leave it unpaired, and
            # forbidding every cut inside the backbones would collapse the overlap into one or two buckets -- ADVICE r3)
            bwd_paired = (os.environ.get("DYK_PAIR", "0") != "0" and os.environ.get("DYK_PAIR_WHICH", "both") in ("both", "bwd")
                          and os.environ.get("DYK_PAIR_OPS", "ew") != "none")
            twin_spans = sorted((l, t) for l, t in getattr(plan, "twin_layer", {}).items() if l < t) if bwd_paired else []
            for k in range(1, len(marks)):
                c_end, layer_done = marks[k][0], marks[k - 1][1]     # commands [.., c_end) finish layer `layer_done`
                lo = min((o for l, o in first_off.items() if l >= layer_done), default=hi)
                cut_ok = getattr(plan, "bwd_cut_ok", None)          # atomic-free wgrads: cut only where the planes are folded
                if cut_ok is not None and c_end not in cut_ok and k != len(marks) - 1:
                    continue
                if k != len(marks) - 1 and any(l < layer_done <= t for l, t in twin_spans):
                    continue
                if (lo <= cuts[0] and lo < hi and c_end > c_prev) or k == len(marks) - 1:
                    lo = 0 if k == len(marks) - 1 else lo
                    segs.append((c_prev, c_end, lo, hi))
                    c_prev, hi = c_end, lo
                    while len(cuts) > 1 and lo <= cuts[0]:
                        cuts.pop(0)
            if c_prev < len(plan.bwd):
                segs.append((c_prev, len(plan.bwd), 0, hi))
            plan.__dict__.setdefault("_ddp_segs", {})[ckey] = segs
        return segs

    def bucket_ready(self, lo, hi):
        if hi > lo and not os.environ.get("DYK_DDP_NOREDUCE"):          # (analysis switch: segmentation without the collective)
            g = self.engine.store.G[lo:hi]
            self._works.append((self.dist.all_reduce(g, op=self.dist.ReduceOp.SUM, async_op=True), lo, hi))

    def all_reduce(self, optimizer=None):
        """wait for the bucketed exchange started during backward (call before optimizer.step()).
        optimizer: a dyk.optim fused optimizer whose step() is the NEXT thing done with the gradients -- the buckets are then
        handed to it instead of being waited for here: step() updates each bucket's parameter range on a side stream as soon as
        that bucket's all-reduce has completed, while the backward pass of the early layers is still running (the one-GPU step does
        the same with a single cut, dyk/optim.py); the caller's stream waits for the side stream at the end of step().  Anything
        else that reads the gradients between the two calls (clipping, logging) needs the plain form."""
        if (optimizer is not None and hasattr(optimizer, "_take_buckets") and os.environ.get("DYK_OPT_OVERLAP", "1") != "0"
                and self._works and optimizer._take_buckets(self._works, self.engine.store.total)):
            self._works = []
            return
        for w, _, _ in self._works:
            w.wait()
        self._works = []
