"""Shared test helpers (fixture loading, oracle construction)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
CFGS = ["kaist_yolov3", "kaist_dyolov3_add_sl", "kaist_dyolov4_fshare_global_concat_se3",
        "kaist_dyolov4_mobilenetv3_fshare_global_cse3", "kaist_dyolov4_mobilenetv2_fshare_global_cse3",
        "kaist_dyolov3_concat_inc"]
C1, C2, C3, C5, MNV2, INC = CFGS


def golden_sections(name):
    from build_utils.parse_config import sections_from_json
    return sections_from_json(os.path.join(GOLDEN, "parse_%s.json" % name))


def same_sections(a, b):
    """exact equality of two parsed cfgs, including value types"""
    if len(a) != len(b):
        return False
    for da, db in zip(a, b):
        if list(da.keys()) != list(db.keys()):
            return False
        for k in da:
            va, vb = da[k], db[k]
            if isinstance(va, np.ndarray) or isinstance(vb, np.ndarray):
                if not (isinstance(va, np.ndarray) and isinstance(vb, np.ndarray)):
                    return False
                if va.dtype != vb.dtype or va.shape != vb.shape or not np.array_equal(va, vb):
                    return False
            elif type(va) is not type(vb) or va != vb:
                return False
    return True


def oracle_net(name):
    from oracle.model import OracleNet
    return OracleNet(golden_sections(name), "config/%s.cfg" % name)


def hyp(name="hyp.scratch.4"):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def tref_to_nchw(plan, t):
    """a plan tensor (dyk.plan.TRef: channels-last rows of `ld` elements inside an arena) as a float32 NCHW CPU tensor"""
    import torch
    es = t.esize
    dt = torch.float32 if es == 4 else torch.bfloat16
    a = plan.arenas[t.arena].tensor
    n = t.npix * t.ld
    flat = a[t.off:t.off + (n - (t.ld - t.C)) * es].view(dt)
    v = torch.as_strided(flat, (t.B, t.H, t.W, t.C), (t.H * t.W * t.ld, t.W * t.ld, t.ld, 1))
    return v.float().permute(0, 3, 1, 2).contiguous().cpu()
