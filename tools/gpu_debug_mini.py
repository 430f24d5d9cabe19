"""Tiny cfgs through the full plan machinery vs the oracle (debugging aid)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

from build_utils.parse_config import parse_model_cfg  # noqa: E402
from models import YOLO  # noqa: E402
from oracle.model import OracleNet  # noqa: E402

HEAD = """
[convolutional]
size=1
stride=1
pad=1
filters=18
activation=linear

[yolo]
mask = 0,1,2
anchors = 16,32, 18,42, 22,44
classes=1
num=3
"""


def conv(f, k, s=1, act="leaky"):
    return "\n[convolutional]\nbatch_normalize=1\nfilters=%d\nsize=%d\nstride=%d\npad=1\nactivation=%s\n" % (f, k, s, act)


CASES = {
    "a_stem_1x1": "[net]\nchannels=3\n" + conv(32, 3) + conv(64, 1) + HEAD,
    "b_stem_3x3": "[net]\nchannels=3\n" + conv(32, 3) + conv(64, 3) + HEAD,
    "c_down": "[net]\nchannels=3\n" + conv(32, 3) + conv(64, 3, 2) + conv(64, 3) + HEAD,
    "d_deep": "[net]\nchannels=3\n" + conv(32, 3) + conv(64, 3, 2) + conv(128, 3, 2) + conv(128, 3, 2) + conv(256, 3, 2, "mish") + conv(128, 1) + conv(256, 3) + HEAD,
}


def run(name, text, dtype, hw):
    d = tempfile.mkdtemp()
    path = os.path.join(d, "mini_yolov4_%s.cfg" % name)
    with open(path, "w") as f:
        f.write(text)
    defs = parse_model_cfg(path)
    net = OracleNet(defs, path)
    sd = net.synth_state(0)
    model = YOLO(path)
    model.load_state_dict(sd)
    model.dyk_dtype = dtype
    model = model.cuda().train()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(2, 3, hw[0], hw[1], generator=g)
    for k, v in sd.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    ref = net.forward(sd, x, None, training=True)
    out = model(x.cuda())
    loss_ref = sum((t ** 2).mean() for t in ref)
    loss_ref.backward()
    loss = sum((t ** 2).mean() for t in out)
    loss.backward()
    torch.cuda.synchronize()
    print("== %s %s %s loss ref %.6f got %.6f" % (name, dtype, hw, loss_ref.item(), loss.item()))
    for k, p in model.named_parameters():
        gr = sd[k].grad
        gg = p.grad.detach().cpu()
        err = (gg - gr).abs().max().item()
        rel = err / max(gr.abs().max().item(), 1e-12)
        print("   %-40s max|ref| %.3e err %.3e rel %.2e %s" % (k, gr.abs().max().item(), err, rel, "BAD" if rel > 2e-3 else ""))


if __name__ == "__main__":
    for name, text in CASES.items():
        for hw in ((32, 64), (64, 96)):
            run(name, text, "fp32", hw)
