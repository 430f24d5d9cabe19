"""Data-parallel path on ONE GPU (`-m gpu`): torch.distributed backend "nccl" (= RCCL) with world_size 1 runs the REAL
segmented backward (dyk_run_commands* on sub-ranges of the command list, side streams included) with a REAL
asynchronous all-reduce enqueued behind every segment.  Weight gradients go through per-split planes and the
statistics through fixed-order / fp64 reductions, so the result must equal the monolithic backward bit for bit.
SURVEY 4's N-rank == 1-rank oracle is emulated with two micro-batches: what two ranks would each compute and the
all-reduce would sum is accumulated by two passes through the exchange path and must equal the sum of two
independent monolithic runs; the fused optimizer then applies the 1/N average."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import C3, C5, GOLDEN, hyp, oracle_net

sys.path.insert(0, GOLDEN)
import cases  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    port = 29600 + (os.getpid() % 300)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _model(name, dtype):
    from build_utils.parse_config import materialize_cfg
    from models import YOLO
    torch.manual_seed(0)
    m = YOLO(materialize_cfg(name))
    m.load_state_dict(oracle_net(name).synth_state(0))
    m.dyk_dtype = dtype
    m.nc, m.hyp, m.gr = 1, hyp("hyp.scratch.4"), 1.0
    return m.cuda().train()


def _batch(step, B=2, H=128, W=160):
    g = torch.Generator().manual_seed(100 + step)
    x, y = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
    tg = torch.zeros(B * 3, 6)
    tg[:, 0] = torch.arange(B).repeat_interleave(3).float()
    tg[:, 2:4] = torch.rand(B * 3, 2, generator=g) * 0.8 + 0.1
    tg[:, 4:6] = torch.rand(B * 3, 2, generator=g) * 0.3 + 0.05
    return x.cuda(), y.cuda(), tg.cuda()


def _backward(m, batch):
    from build_utils.utils import compute_loss
    x, y, tg = batch
    ld = compute_loss(m(x, y), tg, m)
    (ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]).backward()
    return ld


@pytest.mark.parametrize("name,dtype", [(C3, "fp32"), (C3, "bf16"), (C5, "bf16")])
def test_segmented_backward_with_nccl_allreduce_equals_monolithic_bitwise(nccl_world1, name, dtype):
    from dyk.ddp import GradAllReduce
    batch = _batch(0)
    ref = _model(name, dtype)
    _backward(ref, batch)
    g_ref = ref.engine.store.G.clone()
    m = _model(name, dtype)
    red = GradAllReduce(m, nccl_world1, n_buckets=8)
    calls = []
    orig = red.bucket_ready
    red.bucket_ready = lambda lo, hi: (calls.append((lo, hi)), orig(lo, hi))[1]
    _backward(m, batch)
    red.all_reduce()
    torch.cuda.synchronize()
    total = m.engine.store.total
    assert len(calls) >= 3 and calls[0][1] == total and calls[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(calls, calls[1:])), "buckets must tile the gradient buffer"
    assert bool(torch.isfinite(g_ref).all()) and float(g_ref.abs().sum()) > 0
    assert torch.equal(m.engine.store.G, g_ref), "segmented + all-reduced gradients differ from the monolithic backward"


def test_two_rank_emulation_matches_sum_of_independent_ranks(nccl_world1):
    """rank r of a 2-rank job sees micro-batch r; the SUM all-reduce + grad_scale = 1/2 of the fused step must equal
    Adam on the mean of the two per-rank gradients (per-rank BatchNorm statistics and per-rank loss normalisation,
    as stock DDP over the reference would do: SURVEY 8e)."""
    from dyk.ddp import GradAllReduce, reduce_dict
    from dyk.optim import FusedAdam
    b0, b1 = _batch(1), _batch(2)
    per_rank = []
    for b in (b0, b1):
        r = _model(C3, "fp32")
        _backward(r, b)
        per_rank.append(r.engine.store.G.clone())
    want_sum = per_rank[0] + per_rank[1]
    m = _model(C3, "fp32")
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    red = GradAllReduce(m, nccl_world1)
    opt = FusedAdam(m, lr=1e-3, betas=(0.937, 0.999), weight_decay=5e-4)
    opt.grad_scale = 0.5
    for b in (b0, b1):
        ld = _backward(m, b)
        red.all_reduce()
    # (G accumulates: single-split weight gradients add their tiles onto what is there, so the sum is associated
    # differently from gA + gB -- equal to fp32 rounding, not bit for bit)
    got = m.engine.store.G
    assert float((got - want_sum).abs().max()) <= 1e-5 * float(want_sum.abs().max())
    want_sum = got.clone()
    assert reduce_dict(ld) is ld                       # one rank: the loss dict is returned untouched (distributed_utils.py:127)
    opt.step()
    # the same update through torch.optim.Adam on the averaged gradient
    r = _model(C3, "fp32")
    r.load_state_dict(sd0)
    x, y, _ = b0
    r(x, y)                                            # adopt the store on the device
    st = r.engine.store
    st.attach_grads()
    st.G.copy_(want_sum * 0.5)
    topt = torch.optim.Adam(r.parameters(), lr=1e-3, betas=(0.937, 0.999), weight_decay=5e-4)
    topt.step()
    a, b = m.engine.store.P, st.P
    assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())


def test_optimizer_step_behind_each_bucket_equals_the_plain_order_bitwise(nccl_world1):
    """GradAllReduce.all_reduce(optimizer=opt): the fused step updates every bucket's parameter range on a side stream behind
    that bucket's collective instead of after all of them -- the same parameters, moments and zeroed gradients bit for bit,
    over two steps (the second one with warm moment buffers); a second backward before the step falls back to the plain wait."""
    from dyk.ddp import GradAllReduce
    from dyk.optim import FusedAdam
    res = []
    for deferred in (False, True):
        m = _model(C3, "bf16")
        red = GradAllReduce(m, nccl_world1)
        opt = FusedAdam(m, lr=1e-3, betas=(0.937, 0.999), weight_decay=5e-4)
        for step in range(2):
            _backward(m, _batch(step))
            red.all_reduce(optimizer=opt if deferred else None)
            assert (opt._buckets is not None) == deferred
            opt.step()
            assert opt._buckets is None
        torch.cuda.synchronize()
        res.append((m.engine.store.P.clone(), opt._m.clone(), opt._v.clone(), m.engine.store.G.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert float(res[0][3].abs().sum()) == 0.0
    # two backward passes before the step (gradient accumulation): the buckets no longer tile the buffer once -> plain wait
    _backward(m, _batch(0)); _backward(m, _batch(1))
    red.all_reduce(optimizer=opt)
    assert opt._buckets is None and not red._works


@pytest.mark.parametrize("name,dtype", [(C3, "bf16"), (C5, "bf16"), (C3, "fp32")])
def test_dependency_scheduled_streams_equal_serial_execution_bitwise(name, dtype, monkeypatch):
    """the multi-stream dependency schedule (dyk/sched.py + dyk_run_schedule) must not change a single bit relative to
    the same command lists enqueued in order on one stream: outputs, loss, every gradient, running statistics"""
    from build_utils.utils import compute_loss
    res = []
    for mode in ("serial", "dag", "dag_graph", "dag6", "dag_nopair", "dag_event"):
        monkeypatch.setenv("DYK_OVERLAP", "0" if mode == "serial" else "1")
        monkeypatch.setenv("DYK_PAIR", "0" if mode in ("dag_nopair", "dag_event") else "1")      # two-problem launches of the twin sections
        monkeypatch.setenv("DYK_PAIR_OPS", "all" if mode in ("dag", "dag_graph") else "ew")    # all: convolutions too
        monkeypatch.setenv("DYK_SCHED_POLICY", "hlfet" if mode == "dag6" else ("event" if mode == "dag_event" else "typed"))   # event: the default
        monkeypatch.setenv("DYK_STREAMS", "6" if mode == "dag6" else "4")
        monkeypatch.setenv("DYK_GRAPH", "1" if mode == "dag_graph" else "0")      # hipGraph of the dependency graph (optional path)
        m = _model(name, dtype)
        x, y, tg = _batch(5, B=4)
        outs = []
        for _ in range(3):                                   # later passes: re-armed statistics, accumulated gradients, graphs
            pred = m(x, y)
            ld = compute_loss(pred, tg, m)
            (ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]).backward()
            outs += [p.detach().clone() for p in pred] + [ld["box_loss"].detach().clone(), ld["obj_loss"].detach().clone()]
        torch.cuda.synchronize()
        if mode != "serial":
            plan = next(iter(m.engine.plans.values()))
            sc = plan.schedule("bwd", 0, len(plan.bwd))
            assert len({e["stream"] for e in sc.entries}) >= 3
            # the twin backbones ride in two-problem launches (serial = one command per launch: the comparison partner)
            assert sc.n == len(plan.bwd) - sc.n_pairs
            assert {"dag_nopair": sc.n_pairs == 0, "dag_event": sc.n_pairs == 0, "dag6": sc.n_pairs >= 20}.get(mode, sc.n_pairs >= 100 or name != C3)
            assert bool(plan._graphs) == (mode == "dag_graph"), "the forward graph is captured on the second pass with the same pointers"
        res.append((outs, m.engine.store.G.clone(), m.engine.store.R.clone()))
    for other in res[1:]:
        for a, b in zip(res[0][0], other[0]):
            assert torch.equal(a, b)
        assert torch.equal(res[0][1], other[1]), "gradients differ between serial and scheduled execution"
        assert torch.equal(res[0][2], other[2]), "running statistics differ"


@pytest.mark.parametrize("name", [C3, C5])
def test_grouped_weight_gradient_plan_equals_the_ungrouped_plan(name, monkeypatch):
    """Round 6 (dyk/plan.py _group_wgrads): the plan with grouped weight-gradient launches against the plan without them, same
    weights, same batch, pinned tiles.  Everything but the weight gradients of the grouped layers is bit-identical (grouping moves
    launches, it does not touch the data gradients, the BatchNorm passes or the heads); a grouped member runs the same kernel with
    a different number of K splits, i.e. another fp32 summation order: equal to 1e-3 of the tensor's scale."""
    monkeypatch.setenv("DYK_AUTOTUNE", "0")
    monkeypatch.setenv("DYK_TUNE_CACHE", "0")
    batch = _batch(3, B=4)
    grads, plans = [], []
    for g in ("0", "16"):
        monkeypatch.setenv("DYK_WGRAD_GROUP", g)
        m = _model(name, "bf16")
        _backward(m, batch)
        torch.cuda.synchronize()
        plan = next(p for k, p in m.engine.plans.items() if k[-1])
        grads.append(m.engine.store.G.clone())
        plans.append((plan, m))
    flat, grouped = plans[0][0], plans[1][0]
    assert not flat._wg_groups and len(grouped._wg_groups) >= 6
    assert len(grouped.bwd) < len(flat.bwd)
    g0, g1 = grads
    assert bool(torch.isfinite(g1).all())
    st = plans[1][1].engine.store
    base = st.G.data_ptr()
    member_ranges = []
    for lead_addr, ms in grouped._wg_groups.items():
        for d in ms:
            lo = (d.dw - base) // 4
            n = d.ntaps * d.Cout * (d.lddw if d.lddw > 0 else d.Cin)
            member_ranges.append((lo, lo + n))
    mask = torch.zeros(st.total, dtype=torch.bool, device=g0.device)
    for lo, hi in member_ranges:
        mask[lo:hi] = True
    assert torch.equal(g0[~mask], g1[~mask]), "grouping changed a gradient outside the grouped layers"
    for lo, hi in member_ranges:
        a, b = g0[lo:hi], g1[lo:hi]
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-3 * max(scale, 1e-20), (lo, hi, scale)
