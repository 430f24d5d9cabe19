"""Does a streaming BatchNorm pass run BESIDE an MFMA convolution launched on another stream?  (Round 4: the forward pass of the
step takes the SUM of its kernels' isolated times although two backbone streams alternate conv / normalise launches.)
Times N conv launches on stream A and M normalise passes on stream B, each alone and both together.
Usage: [DYK_LIB=...] python tools/overlap_probe.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "double-yolo-kaist_amd")
sys.path[:0] = [ROOT, PKG]
import torch  # noqa: E402

from dyk import lib as L  # noqa: E402
from dyk import ops  # noqa: E402

lib = L.load()
B, H, W, C = 16, 64, 80, 128
dt = torch.bfloat16
x = torch.randn(B, H, W, C, device="cuda").to(dt)
w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
wp = ops.pack_weight(w, dt)
y = torch.empty(B, H, W, C, device="cuda", dtype=dt)
stats = torch.zeros(32 * 2 * C, dtype=torch.float64, device="cuda")
dc = ops.make_conv_desc(x, wp, y, Hi=H, Wi=W, Cin=C, Cout=C, Hg=H, Wg=W, Ho=H, Wo=W, taps=ops.fwd_taps(3, 1), stats=stats)
dc.stats_slots = 32
u = torch.randn(B, H, W, C, device="cuda").to(dt)
z = torch.empty_like(u)
sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
db = ops.ew_desc(a=u, out=z, act="mish", p0=sc, p1=sh)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run(nc, nb, tune):
    dc.tune = tune
    pa, pb = ctypes.c_void_p(sa.cuda_stream), ctypes.c_void_p(sb.cuda_stream)
    torch.cuda.synchronize()
    e0, e1a, e1b = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    sa.wait_event(e0)
    sb.wait_event(e0)
    for i in range(max(nc, nb)):
        if i < nc:
            lib.dyk_conv_igemm(ctypes.byref(dc), pa)
        if i < nb:
            lib.dyk_bn_act_fwd(ctypes.byref(db), pb)
    e1a.record(sa)
    e1b.record(sb)
    torch.cuda.synchronize()
    return max(e0.elapsed_time(e1a), e0.elapsed_time(e1b)) * 1e3


for name, tune in (("generic 128x160", 0x2280), ("large tile 128x320", (5 << 12) | (1 << 8))):
    for _ in range(2):
        run(20, 40, tune)
    tc, tb, tboth = run(40, 0, tune), run(0, 80, tune), run(40, 80, tune)
    print("%-20s conv x40 alone %.0f us | normalise x80 alone %.0f us | together %.0f us (serial sum %.0f, ideal max %.0f)"
          % (name, tc, tb, tboth, tc + tb, max(tc, tb)), flush=True)
