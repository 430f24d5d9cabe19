"""Oracle-derived fixture (no reference import needed): gradients of sum(mean(p^2)) over the three heads of
the target cfg, evaluated by oracle.model in fp64 and in fp32, sampled at 64 fixed positions per parameter
tensor plus the tensor norms.  Lets the GPU test bound the HIP path's gradient error by torch-fp32's own
error against the fp64 truth without re-running a 10-minute fp64 CPU pass on the GPU box.
    python tests/golden/make_grad64.py [cfg name ...]      (default: the target cfg and the MobileNetV3 one)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd"), os.path.join(ROOT, "tests")]
from helpers import C3, C5, INC, oracle_net  # noqa: E402

K = 64


def sample_index(numel, name):
    g = np.random.RandomState(abs(hash(name)) % (2 ** 31)) if False else np.random.RandomState(sum(map(ord, name)))
    return g.randint(0, numel, size=K)


def main(name):
    torch.set_num_threads(8)
    net = oracle_net(name)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(2, 3, 128, 160, generator=g)
    y = torch.rand(2, 3, 128, 160, generator=g)
    out = {}
    names = None
    for dt, tag in ((torch.float64, "g64"), (torch.float32, "g32")):
        sd = net.synth_state(0)
        for k, v in list(sd.items()):
            if v.dtype.is_floating_point:
                sd[k] = v.to(dt)
                if not k.endswith(("running_mean", "running_var")):
                    sd[k].requires_grad_(True)
        for L in net.layers:
            if "anchors" in L:
                L["anchors"] = L["anchors"].to(dt)
        heads = net.forward(sd, x.to(dt), y.to(dt), training=True)
        sum((t ** 2).mean() for t in heads).backward()
        names = [k for k, v in sd.items() if v.grad is not None]
        out[tag] = np.stack([sd[k].grad.double().flatten()[sample_index(sd[k].numel(), k)].numpy() for k in names])
        out[tag + "_norm"] = np.array([float(sd[k].grad.double().norm()) for k in names])
        if tag == "g32":
            out["err32_norm"] = np.array([float((sd[k].grad.double() - g64_full[k]).norm()) for k in names])
        else:
            g64_full = {k: sd[k].grad.double().clone() for k in names}
    np.savez_compressed(os.path.join(HERE, "grad64_%s.npz" % name), names=np.array(names), **out)
    print("written", len(names))


if __name__ == "__main__":
    for cfg_name in (sys.argv[1:] or [C3, C5, INC]):
        main(cfg_name)
