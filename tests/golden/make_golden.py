"""Generate the golden fixtures under tests/golden/ by importing and RUNNING THE REFERENCE
(/root/reference) in the build container.  Run from the repo root:

    python tests/golden/make_golden.py [parse] [graph] [fwd] [targets] [loss] [decode] [nms] [ap] [step]

The reference never travels to the GPU box; only these small data files do.  Inputs are produced
by seeded torch CPU generators and parameters by oracle.model.OracleNet.synth_state (a pure
function of the parameter shapes and a seed), so tests can rebuild the exact same inputs.
"""
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

CFGS = ["kaist_yolov3", "kaist_dyolov3_add_sl", "kaist_dyolov4_fshare_global_concat_se3",
        "kaist_dyolov4_mobilenetv3_fshare_global_cse3", "kaist_dyolov4_mobilenetv2_fshare_global_cse3",
        "kaist_dyolov3_concat_inc"]

QUIRK_CFG = """[net]
batch = 64
# a comment
  # an indented comment survives the filter and breaks nothing? (no: it has no '=' -> not used here)
width = 512
;width=608
hue = .1
second_index = 2
scales = .1,.1

[convolutional]
batch_normalize = 1
filters=32
size = 3
stride = 1
pad=1
activation = mish

[convolutional]
filters = 18
size = 1
stride = 1
pad = 1
activation = linear

[shortcut]
from = -2
activation = linear
weights_type = 1.0

[route]
layers = -1, 0

[maxpool]
stride=1
size=5

[yolo]
mask = 0,1,2
anchors = 16,32,  18.5,42,  22,44, 30,61
classes = 1
num = 9
jitter = .3
ignore_thresh = .7
truth_thresh = 1
random = 1
scale_x_y = 1.05
iou_thresh = 0.213
"""


def encode_defs(defs):
    out = []
    for d in defs:
        e = {}
        for k, v in d.items():
            if isinstance(v, np.ndarray):
                e[k] = {"__ndarray__": v.tolist(), "dtype": str(v.dtype)}
            elif isinstance(v, (np.integer,)):
                e[k] = int(v)
            elif isinstance(v, (np.floating,)):
                e[k] = float(v)
            else:
                e[k] = v
        out.append(e)
    return out


def gen_parse(ref_parse):
    for name in CFGS:
        defs = ref_parse.parse_model_cfg("config/%s.cfg" % name)
        with open(os.path.join(HERE, "parse_%s.json" % name), "w") as f:
            json.dump(encode_defs(defs), f, separators=(",", ":"))
    qpath = os.path.join(HERE, "quirks.cfg")
    with open(qpath, "w") as f:
        f.write(QUIRK_CFG)
    with open(os.path.join(HERE, "parse_quirks.json"), "w") as f:
        json.dump(encode_defs(ref_parse.parse_model_cfg(qpath)), f, indent=1)
    with open(os.path.join(HERE, "parse_data_cfg.json"), "w") as f:
        json.dump(ref_parse.parse_data_cfg("data/kaist_data.data"), f, indent=1)
    # hyper-parameter files are plain data the loss needs
    import yaml
    for h in ("hyp.scratch.4.yaml", "hyp.scratch.yaml"):
        with open(os.path.join(REF, "config", h)) as f:
            hyp = yaml.safe_load(f)
        with open(os.path.join(HERE, h.replace(".yaml", ".json")), "w") as f:
            json.dump(hyp, f, indent=1)
    print("parse fixtures written")


def gen_graph(ref_models):
    for name in CFGS:
        torch.manual_seed(0)
        m = ref_models.YOLO("config/%s.cfg" % name)
        info = {
            "classes": [mod.__class__.__name__ for mod in m.module_list],
            "routs": [bool(r) for r in m.routs],
            "yolo_layers": m.yolo_layers,
            "state_shapes": [[k, list(v.shape)] for k, v in m.state_dict().items()],
            "n_params": sum(p.numel() for p in m.parameters()),
            "yolo": [{"stride": m.module_list[j].stride, "bf_type": m.module_list[j].bf_type,
                      "anchor_vec": m.module_list[j].anchor_vec.tolist(), "na": m.module_list[j].na,
                      "nc": m.module_list[j].nc} for j in m.yolo_layers],
            # head-bias init is added onto torch's random default; record the offset pattern only
            "net_info_keys": sorted(m.net_info.keys()),
            "second_index": m.net_info.get("second_index", None),
        }
        with open(os.path.join(HERE, "graph_%s.json" % name), "w") as f:
            json.dump(info, f, separators=(",", ":"))
    print("graph fixtures written")


def load_oracle_state_into(ref_model, sd):
    own = ref_model.state_dict()
    assert list(own.keys()) == list(sd.keys()), "state_dict key order differs"
    ref_model.load_state_dict(sd, strict=True)


def gen_fwd(ref_models, ref_parse):
    from oracle.model import OracleNet
    for name in CFGS:
        cfg = "config/%s.cfg" % name
        defs = ref_parse.parse_model_cfg(cfg)
        onet = OracleNet(defs, cfg)
        sd = onet.synth_state(seed=0)
        m = ref_models.YOLO(cfg)
        load_oracle_state_into(m, sd)
        g = torch.Generator().manual_seed(1234)
        H, W = (128, 160)
        x = torch.rand(2, 3, H, W, generator=g)
        y = torch.rand(2, 3, H, W, generator=g)
        rec = {}
        # eval
        m.eval()
        with torch.no_grad():
            io, p = m(x, y)
        rec["eval_io"] = io.numpy()
        for i, t in enumerate(p):
            rec["eval_p%d" % i] = t.numpy()
        # train (updates running stats)
        m.train()
        out = m(x, y)
        for i, t in enumerate(out):
            rec["train_p%d" % i] = t.detach().numpy()
        # a scalar loss-like functional of the outputs and its gradient checksums
        loss = sum((t ** 2).mean() for t in out)
        loss.backward()
        rec["train_loss"] = np.float32(loss.item())
        gsum, gnames = [], []
        for k, prm in m.named_parameters():
            gnames.append(k)
            gsum.append([prm.grad.abs().sum().item(), prm.grad.sum().item()])
        rec["grad_sums"] = np.asarray(gsum, dtype=np.float64)
        sd_after = m.state_dict()
        rs = []
        for k, v in sd_after.items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                rs.append([v.sum().item(), v.abs().max().item()])
        rec["running_sums"] = np.asarray(rs, dtype=np.float64)
        np.savez_compressed(os.path.join(HERE, "fwd_%s.npz" % name), **rec)
        with open(os.path.join(HERE, "fwd_%s_gradnames.json" % name), "w") as f:
            json.dump(gnames, f)
        print("fwd fixture", name, "loss", loss.item())


def main():
    what = sys.argv[1:] or ["parse", "graph", "fwd"]
    ref_models, ref_utils, ref_parse, ref_metrics = import_reference()
    torch.set_num_threads(8)
    if "parse" in what:
        gen_parse(ref_parse)
    if "graph" in what:
        gen_graph(ref_models)
    if "fwd" in what:
        gen_fwd(ref_models, ref_parse)
    import make_golden_loss as mgl
    mgl.run(what, ref_models, ref_utils, ref_parse, ref_metrics, HERE)


if __name__ == "__main__":
    main()
