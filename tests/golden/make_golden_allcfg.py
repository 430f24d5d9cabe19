"""All network definitions of the reference through the reference itself (run in the build container):
  * double-yolo-kaist_amd/config/netdefs/<cfg>.json  -- parse_model_cfg's section tables (value types preserved) for every
    detector cfg under /root/reference/config (the three *_backbone.cfg files have no [yolo] section: YOLO.forward
    cannot run them, skipped);
  * tests/golden/allcfg.npz -- per cfg: the three raw head tensors of an eval forward and of a train forward
    (synthetic parameters = OracleNet.synth_state(seed 0), inputs [1,3,64,96] per stream, generator seed 77) plus
    the loss-like scalar sum(mean(p^2)) of the train heads.
Run:  python tests/golden/make_golden_allcfg.py
"""
import glob
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from ref_import import import_reference, REF  # noqa: E402
from make_golden import encode_defs, load_oracle_state_into  # noqa: E402


def inputs():
    g = torch.Generator().manual_seed(77)
    return torch.rand(1, 3, 64, 96, generator=g), torch.rand(1, 3, 64, 96, generator=g)


def main():
    from oracle.model import OracleNet
    ref_models, ref_utils, ref_parse, ref_metrics = import_reference()
    torch.set_num_threads(8)
    netdefs = os.path.join(ROOT, "double-yolo-kaist_amd", "config", "netdefs")
    out = {}
    names = []
    for path in sorted(glob.glob(os.path.join(REF, "config", "*.cfg"))):
        name = os.path.basename(path)[:-4]
        cfg = "config/%s.cfg" % name
        defs = ref_parse.parse_model_cfg(cfg)
        if not any(d["type"] == "yolo" for d in defs):
            print("skip (no [yolo])", name)
            continue
        with open(os.path.join(netdefs, name + ".json"), "w") as f:
            json.dump(encode_defs(defs), f, separators=(",", ":"))
        sd = OracleNet(defs, cfg).synth_state(seed=0)
        m = ref_models.YOLO(cfg)
        load_oracle_state_into(m, sd)
        x, y = inputs()
        m.eval()
        with torch.no_grad():
            _, p = m(x, y)
        for i, t in enumerate(p):
            out["%s|eval_p%d" % (name, i)] = t.numpy()
        m.train()
        with torch.no_grad():
            tp = m(x, y)
        for i, t in enumerate(tp):
            out["%s|train_p%d" % (name, i)] = t.numpy()
        out["%s|train_loss" % name] = np.float32(sum((t ** 2).mean() for t in tp).item())
        names.append(name)
        print(name, [tuple(t.shape) for t in p])
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "allcfg.npz"), **out)


if __name__ == "__main__":
    main()
