"""The reference train harness's use of the optimizer (kaist_train_eval_utils.py:103-108, train.py:77-91,229) on the
HIP path: skipped steps + zero_grad, --freeze-layers, optimizer checkpoints, and the error for targets on the image
edge that the reference raises inside compute_loss."""
import io
import os

import numpy as np
import pytest
import torch

from helpers import C1, C3, hyp, oracle_net

pytestmark = pytest.mark.gpu


def _model(name, dtype="fp32"):
    from build_utils.parse_config import materialize_cfg
    from models import YOLO
    torch.manual_seed(0)
    m = YOLO(materialize_cfg(name))
    m.load_state_dict(oracle_net(name).synth_state(0))
    m.dyk_dtype = dtype
    m.nc, m.hyp, m.gr = 1, hyp("hyp.scratch.4"), 1.0
    return m.cuda().train()


def _batch(seed, B=2, H=96, W=128):
    g = torch.Generator().manual_seed(seed)
    x, y = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
    tg = torch.zeros(B * 2, 6)
    tg[:, 0] = torch.arange(B).repeat_interleave(2).float()
    tg[:, 2:4] = torch.rand(B * 2, 2, generator=g) * 0.8 + 0.1
    tg[:, 4:6] = torch.rand(B * 2, 2, generator=g) * 0.3 + 0.05
    return x.cuda(), y.cuda(), tg.cuda()


def _backward(m, batch):
    from build_utils.utils import compute_loss
    x, y, tg = batch
    ld = compute_loss(m(x, y), tg, m)
    (ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]).backward()


def test_zero_grad_after_a_skipped_step_drops_the_stale_gradients():
    """`scaler.step(optimizer)` skips optimizer.step() on inf/NaN gradients and the harness then calls
    optimizer.zero_grad(): the next backward must start from zeros (the clean flag was never cleared)."""
    from dyk.optim import FusedAdam
    m = _model(C1)
    opt = FusedAdam(m, lr=1e-3)
    b0, b1 = _batch(1), _batch(2)
    _backward(m, b0)
    opt.step()                                   # zeroes G in the step
    opt.zero_grad()
    _backward(m, b0)                             # ... this step is "skipped"
    assert float(m.engine.store.G.abs().sum()) > 0
    opt.zero_grad()
    assert float(m.engine.store.G.abs().sum()) == 0.0
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    _backward(m, b1)
    g_after = m.engine.store.G.clone()
    r = _model(C1)
    r.load_state_dict(sd)
    _backward(r, b1)
    assert torch.equal(g_after, r.engine.store.G)
    # GradScaler (kaist_train_eval_utils.py:103-108): an inf gradient makes scaler.step skip; zero_grad then clears it
    from build_utils.utils import compute_loss
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10)
    x, y, tg = b0
    ld = compute_loss(m(x, y), tg, m)
    scaler.scale(ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]).backward()
    m.engine.store.G[5] = float("inf")
    p_before = m.engine.store.P.clone()
    scaler.step(opt)
    scaler.update()
    assert torch.equal(m.engine.store.P, p_before), "step must be skipped on non-finite gradients"
    assert scaler.get_scale() == 2.0 ** 9
    opt.zero_grad()
    assert float(m.engine.store.G.abs().sum()) == 0.0


def test_freeze_layers_like_the_reference_harness():
    """train.py:77-91: parameters of module_list[0..k] get requires_grad_(False) and the optimizer is built from the
    rest.  Frozen tensors must stay bit-identical (no update, no weight decay), have no .grad, and the trainable
    layers must receive exactly the gradients of the unfrozen run."""
    from dyk.optim import FusedAdam, FusedSGD
    batch = _batch(3)
    full = _model(C3)
    _backward(full, batch)
    g_full = {e.name: full.engine.store._view(full.engine.store.G, e).clone() for e in full.engine.store.entries}
    for Opt in (FusedAdam, FusedSGD):
        m = _model(C3)
        cut = 224
        for idx in range(cut + 1):
            for p in m.module_list[idx].parameters():
                p.requires_grad_(False)
        opt = Opt(m, lr=1e-2, weight_decay=5e-4)
        assert len(opt.param_groups[0]["params"]) == sum(1 for p in m.parameters() if p.requires_grad)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        _backward(m, batch)                       # must not raise "does not require grad"
        st = m.engine.store
        for e in st.entries:
            if e.param.requires_grad:
                assert torch.equal(e.param.grad, g_full[e.name]), e.name
            else:
                assert e.param.grad is None
        opt.step()
        sd1 = m.state_dict()
        changed = 0
        for e in st.entries:
            same = torch.equal(sd1[e.name], sd0[e.name])
            if e.param.requires_grad:
                changed += int(not same)
            else:
                assert same, "frozen parameter %s moved" % e.name
        assert changed > 0
        # BatchNorm of frozen layers still runs in train mode (running statistics move), as in the reference
        assert not torch.equal(sd1["module_list.0.BatchNorm2d.running_mean"], sd0["module_list.0.BatchNorm2d.running_mean"])


def test_optimizer_state_dict_round_trip_and_format():
    """train.py:229 stores optimizer.state_dict(); a resumed FusedAdam must continue exactly, and the layout is
    torch.optim.Adam's (state[i] = {step, exp_avg, exp_avg_sq} with the parameter's shape)."""
    from dyk.optim import FusedAdam
    b = [_batch(10 + i) for i in range(3)]
    m = _model(C1)
    opt = FusedAdam(m, lr=1e-3, betas=(0.937, 0.999), weight_decay=5e-4)
    for i in range(2):
        _backward(m, b[i])
        opt.step()
    buf = io.BytesIO()
    torch.save({"model": m.state_dict(), "optimizer": opt.state_dict()}, buf)
    _backward(m, b[2])
    opt.step()
    want = m.engine.store.P.clone()
    buf.seek(0)
    ck = torch.load(buf)
    osd = ck["optimizer"]
    params = [p for p in m.parameters()]
    assert sorted(osd["state"].keys()) == list(range(len(params)))
    s0 = osd["state"][0]
    assert set(s0) == {"step", "exp_avg", "exp_avg_sq"} and float(s0["step"]) == 2.0
    assert tuple(s0["exp_avg"].shape) == tuple(params[0].shape)
    ref = torch.optim.Adam(params, lr=1e-3).state_dict()
    assert osd["param_groups"][0]["params"] == ref["param_groups"][0]["params"]
    m2 = _model(C1)
    m2.load_state_dict(ck["model"])
    x, y, _ = b[2]
    m2.train()
    opt2 = FusedAdam(m2, lr=1e-3, betas=(0.937, 0.999), weight_decay=5e-4)
    with torch.no_grad():
        m2.engine.store.adopt(torch.device("cuda", 0)) if m2.engine.store.P is None else None
    opt2.load_state_dict(osd)
    assert opt2._t == 2
    # (running statistics were checkpointed after step 2: the third step sees the same state)
    _backward(m2, b[2])
    opt2.step()
    assert torch.equal(m2.engine.store.P, want)
    # ... and a stock torch.optim.Adam accepts the same checkpoint
    topt = torch.optim.Adam([p for p in m2.parameters()], lr=1e-3)
    topt.load_state_dict(osd)


def test_sgd_resumes_from_a_torch_optim_sgd_checkpoint():
    """torch.optim.SGD's state holds only 'momentum_buffer' (no 'step'); a FusedSGD loading such a checkpoint
    must continue with the loaded momentum, not restart with m = g.  Two steps with torch.optim.SGD(nesterov) on the
    product's gradients, checkpoint, third step: FusedSGD from the checkpoint == torch.optim.SGD continuing."""
    from dyk.optim import FusedSGD
    b = [_batch(20 + i) for i in range(3)]
    m = _model(C1)
    kw = dict(lr=1e-3, momentum=0.937, weight_decay=5e-4, nesterov=True)
    topt = torch.optim.SGD(list(m.parameters()), **kw)
    for i in range(2):
        topt.zero_grad()
        _backward(m, b[i])
        topt.step()
        m.engine.store.mark_dirty()
    ck_model = {k: v.clone() for k, v in m.state_dict().items()}
    import copy
    osd = copy.deepcopy(topt.state_dict())              # (state_dict() hands out the live buffers: the next step would change them)
    assert "step" not in osd["state"][0] and "momentum_buffer" in osd["state"][0]
    topt.zero_grad()
    _backward(m, b[2])
    topt.step()
    want = m.engine.store.P.clone()
    m2 = _model(C1)
    m2.load_state_dict(ck_model)
    x, y, _ = b[2]
    m2(x, y)                                            # adopt the parameter store on the device
    opt2 = FusedSGD(m2, **kw)
    opt2.load_state_dict(osd)
    assert opt2._t >= 1, "a checkpoint with momentum buffers is past the first step"
    opt2.zero_grad()
    _backward(m2, b[2])
    opt2.step()
    got = m2.engine.store.P
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
    # FusedSGD -> FusedSGD round trip keeps its own step counter
    opt3 = FusedSGD(m2, **kw)
    opt3.load_state_dict(opt2.state_dict())
    assert opt3._t == opt2._t


def test_target_on_the_image_edge_raises_index_error():
    """a target with x == 1.0 indexes column nx of the grid: the reference raises IndexError inside compute_loss
    (utils.py:248); here the device flag is raised at the next optimizer step (or on demand)."""
    from build_utils.utils import compute_loss
    from dyk.detect import raise_if_target_outside_grid
    from dyk.optim import FusedAdam
    m = _model(C1)
    opt = FusedAdam(m, lr=1e-3)
    x, y, tg = _batch(4)
    _backward(m, (x, y, tg))
    torch.cuda.synchronize()
    opt.step()                                     # clean targets: nothing raised
    bad = tg.clone()
    bad[0, 2] = 1.0
    ld = compute_loss(m(x, y), bad, m)
    with pytest.raises(IndexError):
        raise_if_target_outside_grid(m)            # waits for the flag
    ld = compute_loss(m(x, y), bad, m)
    (ld["box_loss"] + ld["obj_loss"] + ld["class_loss"]).backward()
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        opt.step()


def test_gradient_edits_between_backward_and_step_reach_the_whole_fused_update():
    """with the fused optimizers the update of the deep layers runs on a side stream.  Whatever the harness
    enqueues on ITS stream between backward() and step() -- clip_grad_norm_ (reference users add it), manual scaling, a norm for
    logging -- must be ordered against that launch.  Here: the gradients are zeroed on the caller's stream behind a long-running
    kernel; Adam without weight decay must then leave EVERY parameter bit-unchanged (a side stream that only waited for the
    event in the middle of the backward pass would still see the old gradients of 95 % of the parameters), and a norm taken
    between the two calls sees the gradients, not the zeros the step leaves behind."""
    from dyk.optim import FusedAdam
    m = _model(C3, "bf16")
    opt = FusedAdam(m, lr=1e-6, weight_decay=0.0)        # (a small first step: the random-weight net must stay finite for the second backward)
    assert not opt.early_start
    _backward(m, _batch(3))
    opt.step()                                        # first step: creates the moment buffers (runs without the overlap)
    _backward(m, _batch(4))
    p_before = m.engine.store.P.clone()
    busy = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    for _ in range(6):
        busy.fill_(1)                                 # a few ms of work on the caller's stream in front of the edit
    gn = torch.nn.utils.clip_grad_norm_(list(m.parameters()), 1e30)       # (a norm over all gradients, no clipping)
    for p in m.parameters():
        p.grad.zero_()
    # moments of step 1 are not zero: neutralise them so that a zero gradient means a zero update
    opt._m.zero_()
    opt._v.zero_()
    opt.step()
    torch.cuda.synchronize()
    assert float(gn) > 0 and np.isfinite(float(gn))
    assert torch.equal(m.engine.store.P, p_before), "the fused step saw gradients the caller had already overwritten"
