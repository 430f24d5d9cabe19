import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd"), os.path.join(ROOT, "tests")]
from helpers import *
from build_utils.parse_config import materialize_cfg
from models import YOLO
for name in (C3, C5, MNV2):
    gold = np.load(os.path.join(GOLDEN, "fwd_%s.npz" % name))
    torch.manual_seed(0)
    m = YOLO(materialize_cfg(name)); m.load_state_dict(oracle_net(name).synth_state(0)); m.dyk_dtype = "fp32"; m = m.cuda().train()
    g = torch.Generator().manual_seed(1234)
    x, y = torch.rand(2, 3, 128, 160, generator=g), torch.rand(2, 3, 128, 160, generator=g)
    out = m(x.cuda(), y.cuda())
    sd = m.state_dict()
    rs = np.array([[v.double().sum().item(), v.abs().max().item()] for k, v in sd.items() if k.endswith("running_mean") or k.endswith("running_var")])
    d = np.abs(rs - gold["running_sums"])
    print(name, "max abs dev", d.max(0), "max rel dev", (d / (np.abs(gold["running_sums"]) + 1e-30)).max(0), "rel w/ atol 1e-3", (d / (np.abs(gold["running_sums"]) + 1e-3)).max(0))
