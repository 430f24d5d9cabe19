"""Row-block 3x3 weight gradient (conv_wgrad_rb.hip) against the per-tap / multi-tap kernels on the target cfg's layer shapes:
isolated time of each variant in plane mode (as the step runs them), us and TF/s.  python tools/rb_probe.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "double-yolo-kaist_amd"))
from dyk import lib as L, ops  # noqa: E402

lib = L.load()
SHAPES = [(128, 128, 64, 80, 1), (256, 256, 32, 40, 1), (512, 512, 16, 20, 1), (128, 256, 64, 80, 1), (256, 512, 32, 40, 1),
          (512, 1024, 16, 20, 1), (1024, 512, 16, 20, 1), (64, 64, 128, 160, 1), (64, 128, 128, 160, 1), (32, 64, 256, 320, 1),
          (128, 256, 64, 80, 2), (256, 512, 32, 40, 2), (512, 1024, 16, 20, 2), (64, 128, 128, 160, 2)]
if len(sys.argv) > 1:
    SHAPES = SHAPES[:int(sys.argv[1])]
B = 16
for (ci, co, Ho, Wo, s) in SHAPES:
    Hi, Wi = Ho * s, Wo * s
    x = torch.randn(B, Hi, Wi, ci, device="cuda").bfloat16()
    dy = torch.randn(B, Ho, Wo, co, device="cuda").bfloat16()
    dw = torch.zeros(9, co, ci, device="cuda")
    plane = 9 * co * ci
    out = []
    for name, tune in [("tap", 2), ("tap3", 3), ("tapkg2", 2 | (2 << 8)), ("mt", 2 | (1 << 28)), ("rb", 2 | (1 << 8) | (2 << 28) | (1 << 20)), ("rb256", 2 | (2 << 8) | (2 << 28)),
                       ("rb-noloop", 2 | (1 << 8) | (2 << 28) | (1 << 17))]:
        if os.environ.get("RB_ONLY") and name != os.environ["RB_ONLY"]:
            continue
        d = L.DykWgradDesc()
        d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
        d.dtype = L.DYK_BF16
        d.ldx, d.lddy = ci, co
        d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, Hi, Wi, ci, Ho, Wo, co
        d.isy = d.isx = s
        taps = ops.fwd_taps(3, 1)
        d.ntaps = 9
        for i, (ty, tx, wt) in enumerate(taps):
            d.tdy[i], d.tdx[i], d.twt[i] = ty, tx, wt
        d.tune, d.splits = tune, 0
        var = lib.dyk_conv_wgrad_variant(C.byref(d))
        if name.startswith("rb") and var != 2 or name == "mt" and var != 1:
            out.append("%s n/a" % name)
            continue
        n = lib.dyk_conv_wgrad_splits(C.byref(d))
        part = torch.empty(max(n, 1) * plane, device="cuda")
        if n >= 2:
            d.part, d.part_stride, d.splits = part.data_ptr(), plane, n
        for _ in range(3):
            L.check(lib.dyk_conv_wgrad(C.byref(d), None), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.dyk_conv_wgrad(C.byref(d), None)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2.0 * B * Ho * Wo * ci * co * 9
        out.append("%s sp%d %.1f us (%d TF)" % (name, n, us, fl / us / 1e6))
    print("c%d>%d @%dx%d s%d | " % (ci, co, Ho, Wo, s) + " | ".join(out), flush=True)
