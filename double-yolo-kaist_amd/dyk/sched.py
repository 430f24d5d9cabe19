"""Multi-stream scheduling of a compiled command list.

The reference executes its module list strictly in order on one stream (models.py:291-305) and leaves the rest to
autograd.  A compiled plan knows every buffer each command reads and writes, so the true dependency graph is
available before the first launch: the two backbones of a dual-stream net are independent stage by stage, the two
branches of every CSP block are independent, weight gradients depend on nothing that follows them, the BatchNorm /
activation passes of one branch can run under the MFMA kernels of another.  On MI355X the individual kernels of this
path are short (10-100 us) and leave most of the 256 CUs idle during their ramp-up and tail, so concurrency across
HIP streams is worth more than any single-kernel tweak.

    accesses()   command -> regions read / written (from the resolved descriptor pointers and the plan's allocations)
    dependencies() RAW / WAR / WAW edges of a command range
    schedule()   list scheduling onto N in-order streams with per-command cost estimates (measured by the plan
                 compiler's autotuner for the convolution kernels, bytes-based for the streaming kernels); the result
                 is an issue-ordered table of (command, stream, events to wait for, record flag) that
                 dyk_run_schedule (csrc/exec.hip) replays with hipStreamWaitEvent / hipEventRecord -- no host sync.
"""
import bisect
import ctypes
import os

from . import lib as L


MAX_WAITS = 7


class Region:
    """a byte range of one tracked allocation; `rs` (row stride in bytes) and [c0, c1) (byte columns inside a row)
    narrow it to a channel slice of a channels-last tensor (two slices of one concat buffer do not conflict)"""
    __slots__ = ("key", "lo", "hi", "rs", "c0", "c1")

    def __init__(self, key, lo, hi, rs=0, c0=0, c1=0):
        self.key, self.lo, self.hi, self.rs, self.c0, self.c1 = key, lo, hi, rs, c0, c1

    def overlaps(self, o):
        if self.key != o.key or self.hi <= o.lo or o.hi <= self.lo:
            return False
        if self.rs and self.rs == o.rs:
            return self.c0 < o.c1 and o.c0 < self.c1
        return True


class Memory:
    """address -> tracked allocation.  Arena blocks (activations, gradients, workspace, statistics), the flat
    gradient buffer, the weight-gradient planes and the plan's output tensors are tracked; everything else a command
    points at (parameters, compute-dtype weight copies, running statistics, caller tensors) is either constant during
    a pass or private to one command."""

    def __init__(self, plan, store):
        self.spans = []                       # (lo, hi, key, blocks or None)
        for name, a in plan.arenas.items():
            if a.tensor is None:
                continue
            base = a.ptr()
            starts = [o for o, _ in a.blocks]
            self.spans.append((base, base + max(a.size, 1), name, (starts, a.blocks)))
        if store.G is not None:
            self.spans.append((store.G.data_ptr(), store.G.data_ptr() + 4 * store.total, "G", None))
        if getattr(plan, "part", None) is not None:
            self.spans.append((plan.part.data_ptr(), plan.part.data_ptr() + plan.part_bytes, "part", None))
        for i, p in enumerate(plan.p_out):
            self.spans.append((p.data_ptr(), p.data_ptr() + p.numel() * 4, "p_out%d" % i, None))
        if plan.io is not None:
            self.spans.append((plan.io.data_ptr(), plan.io.data_ptr() + plan.io.numel() * 4, "io", None))
        self.spans.sort()
        self._los = [s[0] for s in self.spans]

    def _span(self, ptr):
        i = bisect.bisect_right(self._los, ptr) - 1
        if i >= 0 and self.spans[i][0] <= ptr < self.spans[i][1]:
            return self.spans[i]
        return None

    def block(self, ptr, rs=0, width=0):
        """the whole arena block `ptr` lies in (optionally narrowed to the byte columns [ptr % rs, +width))"""
        if not ptr:
            return None
        sp = self._span(ptr)
        if sp is None:
            return None
        lo, hi, key, blocks = sp
        if blocks is None:
            return Region(key, 0, hi - lo)
        starts, blks = blocks
        off = ptr - lo
        j = bisect.bisect_right(starts, off) - 1
        b0, bn = blks[j]
        if rs and width and width < rs:
            c0 = (off - b0) % rs
            if c0 + width <= rs:
                return Region((key, b0), b0, b0 + bn, rs, c0, c0 + width)
        return Region((key, b0), b0, b0 + bn)

    def interval(self, ptr, nbytes):
        """exact byte interval of a flat buffer (gradient buffer, weight-gradient planes)"""
        if not ptr:
            return None
        sp = self._span(ptr)
        if sp is None:
            return None
        lo, hi, key, blocks = sp
        if blocks is not None:               # inside an arena: the block it starts in (callers pass single-block ranges)
            return self.block(ptr)
        return Region(key, ptr - lo, ptr - lo + nbytes)

    def tracked(self, ptr):
        return self._span(ptr) is not None


def _es(code):
    return 2 if code == L.DYK_BF16 else 4


def _v(p):
    return p if isinstance(p, int) else (p or 0)


def accesses(op, d, mem, plan):
    """(reads, writes, barrier) of one command.  barrier = orders against everything (unknown footprint)."""
    R, W = [], []

    def rd(r):
        if r is not None:
            R.append(r)

    def wr(r):
        if r is not None:
            W.append(r)

    def T(ptr, ld, C, es):
        return mem.block(_v(ptr), ld * es, C * es)

    def V(ptr):
        return mem.block(_v(ptr))

    extra = getattr(plan, "_rw_extra", {}).get(ctypes.addressof(d))
    if extra is not None:
        for (p, n) in extra[0]:
            rd(mem.interval(p, n))
        for (p, n) in extra[1]:
            wr(mem.interval(p, n))
        return R, W, False
    if op == L.OP_CONV:
        es = _es(d.dtype)
        eso = 4 if (d.flags & L.EPI_OUT_F32) else es
        rd(T(d.x, d.ldx, d.Cin, es))
        for p in (d.scale, d.shift, d.aux0, d.aux1):
            rd(V(p))
        if d.flags & (L.EPI_RESIDUAL | L.EPI_BNBWD):
            rd(T(d.res, d.ldr, d.Cout, es))
        if d.flags & L.EPI_ADDEND and d.add:          # (add == NULL: the keep-dz form without an addend)
            rd(T(d.add, d.ldy, d.Cout, es))
        wr(T(d.y, d.ldy, d.Cout, eso))
        if d.flags & (L.EPI_STATS | L.EPI_BNBWD):
            wr(V(d.stats))
        if d.flags & L.EPI_BNFWD:          # conv + BatchNorm forward in one launch: normalised output, BatchNorm vectors, arrival counter
            wr(T(d.y2, d.ldy2, d.Cout, es))
            for p in (d.scale, d.shift, d.bn_save_mean, d.bn_save_rstd, d.bn_counter):
                wr(V(p))
    elif op == L.OP_WGRAD:
        es = _es(d.dtype)
        # (a grouped launch -- DykWgradDesc.group, dyk/plan.py _group_wgrads -- reads and writes what its members do)
        for m in getattr(plan, "_wg_groups", {}).get(ctypes.addressof(d), [d]):
            rd(T(m.x, m.ldx, m.Cin, es))
            rd(T(m.dy, m.lddy, m.Cout, es))
            plane = m.ntaps * m.Cout * (m.lddw if m.lddw > 0 else m.Cin) * 4
            if m.part:
                wr(mem.interval(_v(m.part), max(m.splits, 1) * m.part_stride * 4))
            else:
                wr(mem.interval(_v(m.dw), plane))
    elif op in (L.OP_DW_FWD, L.OP_DW_DGRAD, L.OP_DW_WGRAD):
        es = _es(d.dtype)
        if d.pre and op != L.OP_DW_DGRAD:     # normalise + activation on load: scale | shift of the producer's BatchNorm
            rd(V(d.pre))
        if op == L.OP_DW_FWD:
            rd(T(d.x, d.ldx, d.C, es)); wr(T(d.y, d.ldy, d.C, es)); wr(V(d.stats))
        elif op == L.OP_DW_DGRAD:
            rd(T(d.y, d.ldy, d.C, es)); wr(T(d.x, d.ldx, d.C, es))
            if d.flags & L.EW_ACCUM:
                rd(T(d.x, d.ldx, d.C, es))
            if d.res:                      # fused BatchNorm-backward reduce of the producer
                rd(T(d.res, d.ldr, d.C, es)); rd(V(d.bn)); wr(V(d.stats))
        else:
            rd(T(d.x, d.ldx, d.C, es)); rd(T(d.y, d.ldy, d.C, es))
            ext = getattr(plan, "_part_extent", {}).get(ctypes.addressof(d))
            if d.part and ext is not None:
                wr(mem.interval(_v(d.part), ext))
            elif d.part:
                return R, W, True
            else:
                wr(mem.interval(_v(d.dw), d.k * d.k * d.C * 4))
    elif op == L.OP_STEM_FWD:
        es = _es(d.dtype)
        rd(V(d.scale)); rd(V(d.shift))
        wr(T(d.y, d.ldy, d.Cout, es)); wr(V(d.stats))
    elif op == L.OP_STEM_WGRAD:
        rd(T(d.dy, d.lddy, d.Cout, _es(d.dtype)))
        wr(V(d.part)); wr(mem.interval(_v(d.dw), d.Cout * 27 * 4))
        if d.bn_da:                        # fusable BatchNorm-backward apply (decided per call): the union of both forms' accesses
            rd(T(d.bn_da, d.lddy, d.Cout, _es(d.dtype))); rd(T(d.bn_yraw, d.lddy, d.Cout, _es(d.dtype)))
            rd(V(d.bn_vecs)); rd(V(d.bn_red))
            wr(mem.interval(_v(d.bn_dgamma), d.Cout * 4)); wr(mem.interval(_v(d.bn_dbeta), d.Cout * 4))
    elif op == L.OP_BN_FINALIZE:
        wr(V(d.stats)); wr(V(d.scale)); wr(V(d.shift)); wr(V(d.save_mean)); wr(V(d.save_rstd))
    elif op == L.OP_BN_FWD_FUSED:
        f = plan._desc_at[_v(d.p[0])]
        a = plan._desc_at[_v(d.p[1])]
        r1, w1, _ = accesses(L.OP_BN_FINALIZE, f, mem, plan)
        r2, w2, _ = accesses(L.OP_BN_ACT_FWD, a, mem, plan)
        # (the fused kernel only reads the replicas; keeping the finalize's write is merely conservative)
        return r1 + r2, w1 + w2, False
    elif op in (L.OP_BN_ACT_FWD, L.OP_BN_BWD_REDUCE, L.OP_BN_BWD_APPLY, L.OP_AXPBY, L.OP_DOT, L.OP_UPSAMPLE_FWD,
                L.OP_UPSAMPLE_BWD, L.OP_MAXPOOL_FWD, L.OP_MAXPOOL_BWD, L.OP_SE_POOL, L.OP_SE_SCALE):
        es = _es(d.dtype)
        rd(T(d.a, d.lda, d.C, es))
        if d.b:
            rd(T(d.b, d.ldb, d.C, es))
        for p in (d.p0, d.p1, d.p2, d.p3):
            rd(V(p))
        if op in (L.OP_BN_BWD_REDUCE, L.OP_DOT) or (op == L.OP_SE_SCALE and d.red):
            wr(V(d.red))                         # (dyk_se_scale with the fused BatchNorm-backward reduce adds to the replicas)
        else:
            rd(V(d.red))
        if op == L.OP_BN_BWD_APPLY:
            wr(mem.interval(_v(d.aux), d.C * 4)); wr(mem.interval(_v(d.aux2), d.C * 4))
        elif op in (L.OP_MAXPOOL_FWD, L.OP_SE_POOL):
            wr(V(d.aux))
            if op == L.OP_SE_POOL:
                wr(V(d.aux2))
        elif op == L.OP_MAXPOOL_BWD:
            rd(V(d.aux))
        if d.out:
            wr(T(d.out, d.ldo, d.C, es))
    elif op == L.OP_SE_FC_FWD:
        rd(V(d.pooled)); wr(V(d.scale)); wr(V(d.ws))
    elif op == L.OP_SE_FC_BWD:
        rd(V(d.dscale))
        if _v(d.dpooled):                   # data half: dt1 -> ws, dpooled
            wr(V(d.dpooled)); wr(V(d.ws))
        if _v(d.dw1):                       # parameter half: reads h / dt1 / t2 (ws) and the pooled activations
            rd(V(d.pooled)); rd(V(d.ws))
            for p, n in ((d.dw1, d.Cs * d.C), (d.db1, d.Cs), (d.dw2, d.C * d.Cs), (d.db2, d.C)):
                wr(mem.interval(_v(p), n * 4))
    elif op == L.OP_BN_FOLD:
        wr(V(d.p[4])); wr(V(d.p[5]))
    elif op == L.OP_WFUSE_WEIGHTS:
        wr(V(d.p[1]))
    elif op == L.OP_WFUSE_BWD_PARAMS:
        rd(V(d.p[1])); wr(mem.interval(_v(d.p[2]), d.i[0] * 4))
    elif op == L.OP_HEAD_PERMUTE_FWD:
        rd(V(d.p[0])); wr(mem.interval(_v(d.p[1]), d.i[0] * d.i[1] * d.i[2] * d.i[3] * d.i[4] * 4))
    elif op == L.OP_HEAD_PERMUTE_BWD:
        wr(V(d.p[1])); wr(mem.interval(_v(d.p[2]), d.i[3] * d.i[4] * 4))
    elif op == L.OP_PATCH_GATHER:
        wr(V(d.p[1]))
    elif op == L.OP_YOLO_DECODE:
        rd(mem.interval(_v(d.p), d.B * d.na * d.ny * d.nx * d.no * 4))
        wr(mem.interval(_v(d.io), d.B * d.rows_total * d.no * 4))
    else:                                   # MEMSET over a span of blocks, and anything not listed above
        return R, W, True
    return R, W, False


def dependencies(cmds, mem, plan):
    """deps[i] = sorted indices j < i that command i must follow (RAW, WAR, WAW on overlapping regions; a barrier
    command follows everything before it and precedes everything after it)"""
    n = len(cmds)
    deps = [set() for _ in range(n)]
    hist = {}                # key -> list of (region, index, is_write) still relevant
    last_barrier = -1
    for i, (op, d) in enumerate(cmds):
        R, W, barrier = accesses(op, d, mem, plan)
        if last_barrier >= 0:
            deps[i].add(last_barrier)
        if barrier:
            deps[i].update(range(max(last_barrier, 0), i))
            last_barrier = i
            hist = {}
            continue
        for r in R:
            for (q, j, w) in hist.get(r.key, ()):
                if w and q.overlaps(r):
                    deps[i].add(j)
        for r in W:
            lst = hist.get(r.key, ())
            keep = []
            for (q, j, w) in lst:
                if q.overlaps(r):
                    deps[i].add(j)
                    # an older access fully shadowed by this write need not be remembered (same block, same columns)
                    if q.lo >= r.lo and q.hi <= r.hi and (not r.rs or (q.rs == r.rs and q.c0 >= r.c0 and q.c1 <= r.c1)):
                        continue
                keep.append((q, j, w))
            hist[r.key] = keep
        for r in R:
            hist.setdefault(r.key, []).append((r, i, False))
        for r in W:
            hist.setdefault(r.key, []).append((r, i, True))
        deps[i].discard(i)
    return [sorted(s) for s in deps]


def estimate_cost_us(op, d, plan):
    """per-command duration estimate in microseconds: the autotuner's measurement for the MFMA kernels, a
    bytes / bandwidth model (3 TB/s effective + 4 us launch) for the streaming kernels"""
    t = getattr(plan, "_cmd_us", {}).get(ctypes.addressof(d))
    if t is not None:
        return max(t, 1.0)
    if op == L.OP_CONV:
        fl = 2.0 * d.B * d.Hg * d.Wg * d.Cin * d.Cout * d.ntaps
        return 8.0 + fl / 400e6
    if op == L.OP_WGRAD:
        fl = 2.0 * d.B * d.Ho * d.Wo * d.Cin * d.Cout * d.ntaps
        return 10.0 + fl / 350e6
    if op in (L.OP_DW_FWD, L.OP_DW_DGRAD, L.OP_DW_WGRAD):
        return 5.0 + 2.0 * d.B * d.Hi * d.Wi * d.C * _es(d.dtype) / 2e6
    if op in (L.OP_BN_ACT_FWD, L.OP_BN_BWD_REDUCE, L.OP_BN_BWD_APPLY, L.OP_AXPBY, L.OP_DOT, L.OP_UPSAMPLE_FWD,
              L.OP_UPSAMPLE_BWD, L.OP_MAXPOOL_FWD, L.OP_MAXPOOL_BWD, L.OP_SE_POOL, L.OP_SE_SCALE):
        passes = {L.OP_BN_BWD_APPLY: 3, L.OP_BN_ACT_FWD: 2, L.OP_BN_BWD_REDUCE: 2}.get(op, 2)
        return 4.0 + passes * float(d.npix) * d.C * _es(d.dtype) / 3e6
    if op == L.OP_BN_FWD_FUSED:
        a = plan._desc_at[_v(d.p[1])]
        return 6.0 + 2.0 * float(a.npix) * a.C * _es(a.dtype) / 3e6
    if op == L.OP_GRAD_REDUCE:
        return 50.0
    if op in (L.OP_STEM_FWD, L.OP_STEM_WGRAD):
        return 10.0 + float(d.B) * d.Ho * d.Wo * d.Cout * _es(d.dtype) / 3e6
    return 5.0


class Schedule:
    """issue-ordered table for dyk_run_schedule (+ the dependency lists in CSR form for dyk_dag_graph_create)"""

    def set_deps(self, deps):
        off = [0]
        flat = []
        for d in deps:
            flat.extend(d)
            off.append(len(flat))
        self.dep_off = (ctypes.c_int32 * len(off))(*off)
        self.dep_idx = (ctypes.c_int32 * max(len(flat), 1))(*flat)

    def __init__(self, entries, n_streams, makespan_us, serial_us):
        self.n = len(entries)
        self.n_streams = n_streams
        self.makespan_us, self.serial_us = makespan_us, serial_us
        self.entries = entries                        # list of dict(cmd, stream, waits, record)
        arr = (L.DykSchedEntry * max(self.n, 1))()
        for k, e in enumerate(entries):
            arr[k].cmd, arr[k].stream, arr[k].record = e["cmd"], e["stream"], 1 if e["record"] else 0
            arr[k].cmd2 = e.get("cmd2", -1)
            arr[k].nwait = len(e["waits"])
            for q, w in enumerate(e["waits"]):
                arr[k].wait[q] = w
        self.array = arr


def schedule(cmds, deps, costs, n_streams, first=0, filler=None, klass=None, policy="hlfet"):
    """List scheduling (highest bottom level first) onto `n_streams` in-order streams.  `policy`: "event" (default,
    round 4) advances a simulated clock -- among the commands that could start earliest (producers finished, a stream
    free) the one with the longest remaining dependency chain goes next; "hlfet" (rounds 2-3) ignores the clock: among the
    commands whose producers are all placed, the one with the longest remaining dependency chain goes next.  Either way
    the command goes onto the stream where it can start earliest; ties prefer the stream of its latest-finishing producer
    (no event needed), then the lowest index.  The critical chain therefore stays on one stream while independent branches and the weight gradients
    (which nothing in a pass reads) fill the others.  `filler`: optional set of command indices restricted to the last
    stream (a low-priority stream in the executor).  Returns a Schedule whose entries are sorted by simulated start
    time -- the host issues in that order so that no stream starves behind another one's commands.  `first` is added
    to the command indices (sub-range schedules).

    `klass` (resource-typed streams, DYK_SCHED_POLICY=typed): klass[i] = 0 for commands bound by the matrix pipe and its
    operand path (implicit-GEMM convolutions, weight gradients), 1 for streaming kernels (BatchNorm / activation passes,
    element-wise, pooling).  Measured (round 3, two-problem launches): a convolution launch with twice the workgroups takes
    exactly twice as long at every layer of the target cfg -- ONE 4-wave workgroup per CU already saturates the CU's
    operand path -- so two MFMA kernels on two streams only share the chip, while a streaming kernel (no LDS, HBM-bound)
    runs beside an MFMA kernel almost for free.  With `klass` all class-0 commands go to stream 0, in an order that never
    lets them compete with each other, and the class-1 commands to streams 1.. where they overlap with whatever stream 0
    runs; a ready command of lower priority is placed first only when it finishes before the highest-priority one of its
    class could start (backfill without delay)."""
    import heapq
    n = len(cmds)
    users = [[] for _ in range(n)]
    for i in range(n):
        for j in deps[i]:
            users[j].append(i)
    blevel = [0.0] * n
    for i in range(n - 1, -1, -1):
        blevel[i] = costs[i] + max((blevel[u] for u in users[i]), default=0.0)
    missing = [len(deps[i]) for i in range(n)]
    n_fill = 1 if (filler and n_streams > 1) else 0
    general = n_streams - n_fill
    avail = [0.0] * n_streams
    start, finish, stream_of = [0.0] * n, [0.0] * n, [0] * n
    pos_in_stream = [0] * n
    count = [0] * n_streams
    placed = 0

    def place(i):
        """command i onto the stream of its class where it starts earliest"""
        ready, prod = 0.0, -1
        for j in deps[i]:
            if finish[j] > ready:
                ready, prod = finish[j], j
        if filler and i in filler and n_streams > 1:
            cand = range(general, n_streams)
        elif klass is not None and n_streams > 1:
            cand = [0] if klass[i] == 0 else range(1, general)
        else:
            cand = range(general)
        pref = stream_of[prod] if prod >= 0 else 0
        best, best_t = None, None
        for s_ in cand:
            t = max(ready, avail[s_])
            if best is None or t < best_t - 1e-9 or (abs(t - best_t) <= 1e-9 and s_ == pref and best != pref):
                best, best_t = s_, t
        return best, best_t

    if klass is None and policy == "event":
        # time-driven list scheduling: the next command is the highest-priority one among those that could START earliest
        # (producers finished, a stream free).  The plain rule above places commands in priority order and only ever
        # appends to a stream, so every weight gradient -- low priority: nothing in the pass reads it -- is placed after the
        # whole critical chain, the expensive late ones first, and an in-order stream then holds the early cheap ones
        # behind them (MobileNetV3 cfg: the other streams idle through the neck's backward while its weight gradients wait)
        ready_t = {i: 0.0 for i in range(n) if missing[i] == 0}
        while ready_t:
            free_g = min(avail[:general])
            free_f = min(avail[general:]) if n_fill else free_g
            est = {j: max(r, free_f if (n_fill and j in filler) else free_g) for j, r in ready_t.items()}
            t_min = min(est.values())
            i = max((j for j, t in est.items() if t <= t_min + 1e-9), key=lambda j: (blevel[j], -j))
            del ready_t[i]
            best, best_t = place(i)
            start[i], finish[i], stream_of[i] = best_t, best_t + costs[i], best
            avail[best] = finish[i]
            pos_in_stream[i] = count[best]
            count[best] += 1
            placed += 1
            for u in users[i]:
                missing[u] -= 1
                if missing[u] == 0:
                    ready_t[u] = max(finish[j] for j in deps[u])
    elif klass is None:
        heap = [(-blevel[i], i) for i in range(n) if missing[i] == 0]
        heapq.heapify(heap)
        while heap:
            _, i = heapq.heappop(heap)
            best, best_t = place(i)
            start[i], finish[i], stream_of[i] = best_t, best_t + costs[i], best
            avail[best] = finish[i]
            pos_in_stream[i] = count[best]
            count[best] += 1
            placed += 1
            for u in users[i]:
                missing[u] -= 1
                if missing[u] == 0:
                    heapq.heappush(heap, (-blevel[u], u))
    else:
        ready_set = {i for i in range(n) if missing[i] == 0}
        while ready_set:
            # the highest-priority ready command, and where / when it could start
            top = max(ready_set, key=lambda i: (blevel[i], -i))
            s_top, t_top = place(top)
            pick, s_pick, t_pick = top, s_top, t_top
            # backfill: another ready command that fits in front of it on the same stream without delaying it
            if t_top > avail[s_top] + 1e-9:
                bestb = None
                for j in ready_set:
                    if j == top:
                        continue
                    sj, tj = place(j)
                    if sj == s_top and tj + costs[j] <= t_top + 1e-9 and (bestb is None or blevel[j] > blevel[bestb[0]]):
                        bestb = (j, sj, tj)
                if bestb is not None:
                    pick, s_pick, t_pick = bestb
            i = pick
            ready_set.discard(i)
            start[i], finish[i], stream_of[i] = t_pick, t_pick + costs[i], s_pick
            avail[s_pick] = finish[i]
            pos_in_stream[i] = count[s_pick]
            count[s_pick] += 1
            placed += 1
            for u in users[i]:
                missing[u] -= 1
                if missing[u] == 0:
                    ready_set.add(u)
    assert placed == n, "dependency cycle"
    # issue order: by simulated start time, but never ahead of an earlier command of the same stream
    order = sorted(range(n), key=lambda i: (start[i], pos_in_stream[i], i))
    issue_pos = {c: k for k, c in enumerate(order)}
    waited = {}                                   # (stream, other stream) -> highest position of `other` already waited for
    entries = []
    needs_record = set()
    for c in order:
        s_ = stream_of[c]
        latest = {}
        for j in deps[c]:
            t = stream_of[j]
            if t != s_ and (t not in latest or pos_in_stream[j] > pos_in_stream[latest[t]]):
                latest[t] = j
        waits = []
        for t, j in sorted(latest.items()):
            if waited.get((s_, t), -1) >= pos_in_stream[j]:
                continue
            waited[(s_, t)] = pos_in_stream[j]
            assert issue_pos[j] < issue_pos[c]
            waits.append(issue_pos[j])
            needs_record.add(j)
        entries.append({"cmd": c + first, "stream": s_, "waits": waits, "record": False, "_c": c})
    for e in entries:
        e["record"] = e["_c"] in needs_record
        if len(e["waits"]) > MAX_WAITS:
            raise RuntimeError("schedule entry with %d waits" % len(e["waits"]))
    sc = Schedule(entries, n_streams, max(finish) if n else 0.0, sum(costs))
    sc.entry_deps = [sorted(issue_pos[j] for j in deps[c]) for c in order]     # per entry: the entries it follows (issue positions)
    return sc


def build(plan, store, which, start, end, n_streams=None):
    """schedule of commands [start, end) of the plan's forward / backward list"""
    cmds = (plan.fwd if which == "fwd" else plan.bwd)[start:end]
    if n_streams is None:
        n_streams = int(os.environ.get("DYK_STREAMS_" + which.upper(), os.environ.get("DYK_STREAMS", "4")))
    if which == "fwd" and getattr(plan, "has_bnfwd", False):
        n_streams = min(n_streams, 2)      # DYK_EPI_BNFWD launches wait on their own workgroups: at most two of them side by side
    mem = Memory(plan, store)
    deps = dependencies(cmds, mem, plan)
    costs = [estimate_cost_us(op, d, plan) for op, d in cmds]
    filler = None        # (weight gradients restricted to dedicated filler streams: neutral, round 4 -- r04_ab_sched_event_filler_streams.log)
    # resource-typed streams (all MFMA kernels on stream 0, streaming kernels beside them): OFF by default -- measured
    # 43.0 ms vs 35.3 (batch 1: 18.1 vs 10.2): every conv -> BatchNorm -> conv hop then crosses streams, and a cross-stream
    # event dependency costs ~7-10 us on this stack, more than the overlap it buys
    klass = None
    policy = os.environ.get("DYK_SCHED_POLICY", "event")
    if policy == "typed" and n_streams > 1:
        klass = [0 if op in (L.OP_CONV, L.OP_WGRAD) else 1 for op, _ in cmds]
    # two-problem launches for the twin sections of a dual-stream net (dyk/twins.py): commands of twin sections with
    # equal descriptors that the dependency graph leaves unordered are contracted into one node each; the schedule is
    # built on the contracted graph and every entry carries its second command (DykSchedEntry.cmd2).
    # OFF by default (DYK_PAIR=1 enables): measured on MI355X, same box, target cfg at batch 16 (tools/ab.sh): everything
    # paired 37.2 ms vs 36.0 unpaired; only the streaming passes paired 39.0 vs 35.5; batch 1 10.5 vs 10.5; MobileNetV3
    # cfg 23.5 vs 21.5 -- a two-problem MFMA launch takes 1.8-2.1x a single one at every layer (one 4-wave workgroup per
    # CU already saturates the CU's operand path), and contracting the twins chains the two backbones into ONE
    # dependency chain whose cross-stream events cost more than the launches saved.
    pairs = []
    twin_of = getattr(plan, "twin_layer", None)
    layer_of = getattr(plan, which + "_layer", None)
    pair_which = os.environ.get("DYK_PAIR_WHICH", "both")      # "fwd" | "bwd" | "both": which pass of the step pairs its twins
    if twin_of and layer_of is not None and os.environ.get("DYK_PAIR", "0") != "0" and pair_which in ("both", which):
        from . import twins
        pairs = twins.find_pairs(cmds, deps, layer_of[start:end], twin_of, plan, os.environ.get("DYK_PAIR_OPS", "ew"),
                                 48e6)
    if pairs:
        members, ndeps, ncosts = twins.merge(len(cmds), deps, costs, pairs,
                                             0.8)
        if filler:
            filler = {k for k, m in enumerate(members) if m[0] in filler}
        if klass is not None:
            klass = [klass[m[0]] for m in members]
        sc = schedule(members, ndeps, ncosts, max(1, min(n_streams, 8)), first=0, filler=filler, klass=klass, policy=policy)
        for e, ent in zip(sc.entries, sc.array):
            m = members[e["cmd"]]
            e["cmd"] = ent.cmd = m[0] + start
            e["cmd2"] = ent.cmd2 = (m[1] + start) if len(m) > 1 else -1
    else:
        sc = schedule(cmds, deps, costs, max(1, min(n_streams, 8)), first=start, filler=filler, klass=klass, policy=policy)
    sc.n_pairs = len(pairs)
    sc.set_deps(_reduce(deps))
    return sc


def _reduce(deps):
    """drop dependencies implied by another one (j in deps[i] and j in deps[k] for some k in deps[i]): fewer graph edges"""
    out = []
    sets = [set(d) for d in deps]
    for i, d in enumerate(deps):
        implied = set()
        for k in d:
            implied |= sets[k]
        out.append([j for j in d if j not in implied])
    return out
