"""conv_sc.hip (3x3 data gradient into 32-channel tensors, fused BatchNorm-backward epilogue) at the target cfg's sizes against the generic kernel.  python tools/sc_probe.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "double-yolo-kaist_amd"))
from dyk import lib as L, ops  # noqa: E402

lib = L.load()
B = 16
for stride, (H, W) in ((2, (512, 640)), (1, (256, 320))):
    Cin, Cout, k = 32, 64, 3
    Ho, Wo = H // stride, W // stride
    dy = torch.randn(B, Ho, Wo, Cout, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    wpt = ops.pack_weight(w, torch.bfloat16, transposed=True)
    yd = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    out = torch.empty(B, H, W, Cin, device="cuda", dtype=torch.bfloat16)
    slots = 16
    red = torch.zeros(slots, 2, Cin, dtype=torch.float64, device="cuda")
    vec = [torch.rand(Cin, device="cuda") + 0.5 for _ in range(4)]
    classes = ops.dgrad_classes(k, 1, stride, H, W)
    d = ops.make_conv_desc(dy, wpt, out, Hi=Ho, Wi=Wo, Cin=Cout, Cout=Cin, Hg=classes[0][2], Wg=classes[0][3], Ho=H, Wo=W,
                           taps=[t for c in classes for t in c[4]], osy=stride, osx=stride, act="mish")
    if stride == 2:
        d.ncls, q0 = len(classes), 0
        for c, (py, px, _, _, taps) in enumerate(classes):
            d.cls_first[c], d.cls_ntaps[c], d.cls_ooy[c], d.cls_oox[c] = q0, len(taps), py, px
            q0 += len(taps)
    d.flags = L.EPI_BNBWD
    d.res, d.ldr = yd.data_ptr(), Cin
    d.scale, d.shift, d.aux0, d.aux1 = (t.data_ptr() for t in vec)
    d.stats, d.stats_slots = red.data_ptr(), slots
    rows = []
    for name, tune in [("generic", 128 | (2 << 8)), ("sc", 6 << 12)]:
        d.tune = tune
        for _ in range(2):
            L.check(lib.dyk_conv_igemm(C.byref(d), None), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.dyk_conv_igemm(C.byref(d), None)
        e1.record(); torch.cuda.synchronize()
        rows.append("%s %.1f" % (name, e0.elapsed_time(e1) / 10 * 1e3))
    print("stride %d -> %dx%dx%d | " % (stride, H, W, Cin) + " | ".join(rows), flush=True)
