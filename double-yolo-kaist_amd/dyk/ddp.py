"""Data-parallel gradient exchange for one-process-per-GPU training (`torch.distributed`, backend "nccl"
= RCCL over xGMI on ROCm).  The reference is single-GPU (build_utils/torch_utils.py:35-36); this is the
north-star's added exchange step: one all-reduce(SUM) of the trainable gradients per optimizer step.

All gradients live in ONE flat fp32 buffer in layer order (dyk/params.py), and the backward command list
runs from the last layer to the first, so the tail of the buffer is final first.  The backward is cut into
a few segments; as soon as a segment's kernels are enqueued, the matching contiguous slice of the buffer
is all-reduced asynchronously (RCCL's stream waits on the compute stream at enqueue time), overlapping
the exchange of late-layer gradients with the differentiation of early layers.  xGMI is point-to-point, so
buckets are kept large (default 8 buckets of ~58 MB for the 464 MB of the target model): few, bandwidth-
bound collectives instead of many latency-bound ones.  The 1/world_size averaging is folded into the fused
optimizer step (FusedAdam.grad_scale).
"""
import torch


class GradAllReduce:
    def __init__(self, model, dist, n_buckets=8):
        self.model, self.dist, self.n_buckets = model, dist, n_buckets
        self.engine = model.engine
        self.engine.grad_sync = self
        self._segs = {}
        self._works = []

    def segments(self, plan):
        key = id(plan)
        segs = self._segs.get(key)
        if segs is None:
            store = self.engine.store
            total = store.total
            # first parameter offset of every layer (entries are in layer order)
            first_off = {}
            for e in store.entries:
                first_off.setdefault(e.layer, e.offset)
            target = max(total // self.n_buckets, 1)
            segs = []
            c_prev, hi = 0, total
            marks = plan.bwd_marks                    # (commands emitted before layer i's backward, i), descending i
            for k in range(1, len(marks)):
                c_end, layer_done = marks[k][0], marks[k - 1][1]     # commands [.., c_end) finish layer `layer_done`
                lo = min((o for l, o in first_off.items() if l >= layer_done), default=hi)
                cut_ok = getattr(plan, "bwd_cut_ok", None)          # atomic-free wgrads: cut only where the planes are folded
                if cut_ok is not None and c_end not in cut_ok and k != len(marks) - 1:
                    continue
                if (hi - lo >= target and c_end > c_prev) or k == len(marks) - 1:
                    lo = 0 if k == len(marks) - 1 else lo
                    segs.append((c_prev, c_end, lo, hi))
                    c_prev, hi = c_end, lo
            if c_prev < len(plan.bwd):
                segs.append((c_prev, len(plan.bwd), 0, hi))
            self._segs[key] = segs
        return segs

    def bucket_ready(self, lo, hi):
        if hi > lo:
            g = self.engine.store.G[lo:hi]
            self._works.append(self.dist.all_reduce(g, op=self.dist.ReduceOp.SUM, async_op=True))

    def all_reduce(self):
        """wait for the bucketed exchange started during backward (call before optimizer.step())"""
        for w in self._works:
            w.wait()
        self._works = []
