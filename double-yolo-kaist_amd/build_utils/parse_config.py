"""Darknet `.cfg` / `.data` parsing with the exact value-typing behaviour of the reference
(build_utils/parse_config.py:5-90), plus the inverse (`dump_model_cfg`) used to materialise the
network definitions shipped as JSON tables under config/netdefs/.

Behaviour kept on purpose:
  * comment / blank filtering happens on the raw line, before stripping (ref :21-23);
  * only strings for which str.isnumeric() holds become numbers, i.e. non-negative integers;
    every float ('.1', '1.0') stays a str (ref :46-49);
  * `anchors` -> float64 ndarray [n,2]; `from` / `layers` / `mask` (and a comma-separated `size`)
    -> list[int]; `[convolutional]` starts with batch_normalize = 0;
  * an unknown key in any section but the first raises ValueError (ref :59-63).
"""
import json
import os

import numpy as np

_INT_LIST_KEYS = ("from", "layers", "mask")

SUPPORTED_KEYS = frozenset([
    "type", "batch_normalize", "filters", "size", "stride", "pad", "activation", "layers", "groups", "from", "mask",
    "anchors", "classes", "num", "jitter", "ignore_thresh", "truth_thresh", "random", "stride_x", "stride_y",
    "weights_type", "weights_normalization", "scale_x_y", "beta_nms", "nms_kind", "iou_loss", "iou_normalizer",
    "cls_normalizer", "iou_thresh", "probability", "max_delta", "atoms", "na", "nc", "squeeze_factor", "n1x1",
    "n3x3_reduce", "n3x3", "n5x5_reduce", "n5x5", "pool_proj"])


def _convert(key, val):
    if key == "anchors":
        nums = [float(tok) for tok in val.replace(" ", "").split(",")]
        return np.array(nums).reshape((-1, 2))
    if key in _INT_LIST_KEYS or (key == "size" and "," in val):
        return [int(tok) for tok in val.split(",")]
    if val.isnumeric():
        return int(val)
    return val


def parse_model_cfg(path: str):
    """cfg file -> list of section dicts ([net] first)."""
    if not path.endswith(".cfg") or not os.path.exists(path):
        raise FileNotFoundError("the cfg file not exist...")
    with open(path, "r", encoding="utf-8") as fh:
        raw = fh.read().split("\n")
    sections = []
    for line in raw:
        if not line or line.startswith("#"):
            continue
        line = line.strip()
        if line.startswith("["):
            sec = {"type": line[1:-1].strip()}
            if sec["type"] == "convolutional":
                sec["batch_normalize"] = 0
            sections.append(sec)
            continue
        key, val = line.split("=")            # ValueError for malformed lines, like the reference
        key, val = key.strip(), val.strip()
        sections[-1][key] = _convert(key, val)
    for sec in sections[1:]:
        for key in sec:
            if key not in SUPPORTED_KEYS:
                raise ValueError("Unsupported fields:{} in cfg".format(key))
    return sections


def parse_data_cfg(path):
    """`.data` file (key = value lines) -> dict of str (ref :68-90)."""
    if not os.path.exists(path) and os.path.exists("data" + os.sep + path):
        path = "data" + os.sep + path
    options = dict()
    with open(path, "r") as fh:
        for line in fh.readlines():
            line = line.strip()
            if line == "" or line.startswith("#"):
                continue
            key, val = line.split("=")
            options[key.strip()] = val.strip()
    return options


# ----------------------------------------------------------------------------- inverse
def _fmt(key, val):
    if isinstance(val, np.ndarray):
        return ", ".join(repr(float(v)) for v in val.reshape(-1))
    if isinstance(val, (list, tuple)):
        return ",".join(str(int(v)) for v in val)
    return str(val)


def dump_model_cfg(sections, path):
    """Write section dicts back as a Darknet cfg such that parse_model_cfg(path) == sections."""
    lines = []
    for sec in sections:
        lines.append("[%s]" % sec["type"])
        for key, val in sec.items():
            if key == "type":
                continue
            if sec["type"] == "convolutional" and key == "batch_normalize" and val == 0:
                continue                      # the parser's default
            lines.append("%s=%s" % (key, _fmt(key, val)))
        lines.append("")
    with open(path, "w", encoding="utf-8") as fh:
        fh.write("\n".join(lines))


def sections_from_json(path):
    with open(path) as fh:
        enc = json.load(fh)
    out = []
    for e in enc:
        d = {}
        for k, v in e.items():
            if isinstance(v, dict) and "__ndarray__" in v:
                d[k] = np.array(v["__ndarray__"], dtype=v.get("dtype", "float64"))
            else:
                d[k] = v
        out.append(d)
    return out


NETDEF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "netdefs")
CFG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config")


def materialize_cfg(name, out_dir=None):
    """Return the path of config/<name>.cfg, writing it from config/netdefs/<name>.json if missing.
    (The network definitions ship as parsed tables; the Darknet text is regenerated from them.)"""
    name = os.path.basename(name)
    if name.endswith(".cfg"):
        name = name[:-4]
    out_dir = out_dir or CFG_DIR
    path = os.path.join(out_dir, name + ".cfg")
    if not os.path.exists(path):
        src = os.path.join(NETDEF_DIR, name + ".json")
        if not os.path.exists(src):
            raise FileNotFoundError("no network definition %s" % src)
        os.makedirs(out_dir, exist_ok=True)
        dump_model_cfg(sections_from_json(src), path)
    return path


def available_netdefs():
    return sorted(f[:-5] for f in os.listdir(NETDEF_DIR) if f.endswith(".json"))
