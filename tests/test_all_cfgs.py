"""Every detector cfg of the reference (25: all files under its config/ that hold [yolo] sections) through the oracle
(CPU) and through the HIP path (GPU), against head tensors produced by the reference itself
(tests/golden/allcfg.npz, tests/golden/make_golden_allcfg.py).  Network definitions travel as parsed section tables
(config/netdefs/*.json)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

GOLD = np.load(os.path.join(GOLDEN, "allcfg.npz"))
NAMES = [str(n) for n in GOLD["names"]]


def _inputs():
    g = torch.Generator().manual_seed(77)
    return torch.rand(1, 3, 64, 96, generator=g), torch.rand(1, 3, 64, 96, generator=g)


def _oracle(name):
    from build_utils.parse_config import NETDEF_DIR, sections_from_json
    from oracle.model import OracleNet
    return OracleNet(sections_from_json(os.path.join(NETDEF_DIR, name + ".json")), "config/%s.cfg" % name)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def test_every_detector_cfg_is_present():
    from build_utils.parse_config import available_netdefs
    assert len(NAMES) == 25 and set(NAMES) <= set(available_netdefs())


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_heads(name):
    net = _oracle(name)
    sd = net.synth_state(seed=0)
    x, y = _inputs()
    with torch.no_grad():
        _, p = net.forward(sd, x, y, training=False)
        tp = net.forward({k: v.clone() for k, v in sd.items()}, x, y, training=True)
    for i in range(3):
        assert _rel(p[i].numpy(), GOLD["%s|eval_p%d" % (name, i)]) < 1e-5, (name, i)
        assert _rel(tp[i].numpy(), GOLD["%s|train_p%d" % (name, i)]) < 1e-4, (name, i)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_path_matches_reference_heads(name):
    """eval forward (running statistics): fp32 tolerance 5e-4 of the head range.  Train forward + backward at this size
    normalises over 6 samples at stride 32, which amplifies fp32 rounding by orders of magnitude (the reference's own
    fp32-vs-fp64 distance is percent-level there): the train heads are held to 5 % of their range at the two finer
    strides (15 % for the ReLU6 / hard-swish MobileNets, measured 7 %; tests/test_gpu_model.py holds the MobileNet
    cfgs to their fp32-vs-fp64 yardstick on better conditioned inputs) and the pass must produce finite gradients for
    every parameter."""
    from build_utils.parse_config import materialize_cfg
    from models import YOLO
    torch.manual_seed(0)
    m = YOLO(materialize_cfg(name))
    m.load_state_dict(_oracle(name).synth_state(seed=0))
    m = m.cuda()
    x, y = _inputs()
    m.eval()
    with torch.no_grad():
        _, p = m(x.cuda(), y.cuda())
    for i in range(3):
        assert _rel(p[i].cpu().numpy(), GOLD["%s|eval_p%d" % (name, i)]) < 5e-4, (name, i)
    m.train()
    out = m(x.cuda(), y.cuda())
    sum((t ** 2).mean() for t in out).backward()
    for i in range(3):
        if out[i].shape[2] > 2:
            tol = 0.15 if "mobilenet" in name else 5e-2
            assert _rel(out[i].detach().cpu().numpy(), GOLD["%s|train_p%d" % (name, i)]) < tol, (name, i)
    for k, prm in m.named_parameters():
        assert prm.grad is not None and bool(torch.isfinite(prm.grad).all()), (name, k)
