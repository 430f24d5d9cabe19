"""Large-tile 3x3 kernels (csrc/conv_lt_kernel.h) against the best generic tile configuration, per problem of the target cfg
at BASELINE size.  Run on the GPU box; prints one line per (problem, epilogue): best generic candidate, every LT variant.
Usage: python tools/lt_probe.py [fwd] [bwd] [quick]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "double-yolo-kaist_amd")
sys.path[:0] = [ROOT, PKG]
import torch  # noqa: E402

from dyk import lib as L  # noqa: E402
from dyk import ops  # noqa: E402
from dyk.plan import _conv_candidates  # noqa: E402

what = sys.argv[1:] or ["fwd", "bwd"]
SHAPES = [  # (Cin, Cout, H, W) of the forward conv, B = 16
    (128, 128, 64, 80), (256, 256, 32, 40), (512, 512, 16, 20), (128, 256, 64, 80), (256, 512, 32, 40), (512, 1024, 16, 20),
    (1024, 512, 16, 20), (64, 64, 128, 160), (256, 128, 64, 80), (512, 256, 32, 40), (64, 128, 128, 160),
]
if "quick" in what:
    SHAPES = SHAPES[:3]
if "thin" in what:
    SHAPES = [(32, 64, 256, 320), (64, 128, 128, 160), (64, 64, 128, 160)]
B = 16
fn = L.load().dyk_conv_igemm


def timeit(d, reps=20):
    for _ in range(3):
        rc = fn(ctypes.byref(d), None)
        if rc != 0:
            return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(ctypes.byref(d), None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def lt_tunes():
    out = []
    for shape in (1, 2, 3):
        for tw in (0, 1, 2, 3, 4):
            out.append((5 << 12) | (shape << 8) | (tw << 24))
    return out


for (ci, co, H, W) in SHAPES:
    dt = torch.bfloat16
    x = torch.randn(B, H, W, ci, device="cuda").to(dt)
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weight(w, dt)
    wpt = ops.pack_weight(w, dt, transposed=True)
    y = torch.empty(B, H, W, co, device="cuda", dtype=dt)
    dy = torch.randn(B, H, W, co, device="cuda").to(dt)
    dx = torch.empty(B, H, W, ci, device="cuda", dtype=dt)
    uprev = torch.randn(B, H, W, ci, device="cuda").to(dt)
    fl = 2.0 * B * H * W * co * ci * 9
    cases = []
    if "fwd" in what:
        stats = torch.zeros(32 * 2 * co, dtype=torch.float64, device="cuda")
        d = ops.make_conv_desc(x, wp, y, Hi=H, Wi=W, Cin=ci, Cout=co, Hg=H, Wg=W, Ho=H, Wo=W, taps=ops.fwd_taps(3, 1), stats=stats)
        d.stats_slots = 32
        cases.append(("fwd+stats", d))
    if "bwd" in what:
        red = torch.zeros(16 * 2 * ci, dtype=torch.float64, device="cuda")
        vec = [torch.rand(ci, device="cuda") + 0.5 for _ in range(4)]
        (py, px, Hg, Wg, taps), = ops.dgrad_classes(3, 1, 1, H, W)
        d = ops.make_conv_desc(dy, wpt, dx, Hi=H, Wi=W, Cin=co, Cout=ci, Hg=Hg, Wg=Wg, Ho=H, Wo=W, taps=taps, act="mish")
        d.flags = L.EPI_BNBWD
        d.res, d.ldr = uprev.data_ptr(), ci
        d.scale, d.shift, d.aux0, d.aux1 = (t.data_ptr() for t in vec)
        d.stats, d.stats_slots = red.data_ptr(), 16
        d._keep = (red, vec)
        cases.append(("bwd+bnbwd(mish)", d))
    for name, d in cases:
        gen = []
        for c in _conv_candidates(d):
            if ((c >> 12) & 0xf) == 5:
                continue
            d.tune = c
            t = timeit(d, 8)
            if t is not None:
                gen.append((t, c))
        gen.sort()
        d.tune = gen[0][1]
        tg = timeit(d)
        if "thin" in what:
            print("   top generic:", " ".join("%#x:%.1f" % (c, t) for t, c in gen[:8]), flush=True)
        row = "%-16s c%d>%d @%dx%d  generic %#x %.1f us (%.0f TF) |" % (name, ci, co, H, W, gen[0][1], tg, fl / tg / 1e6)
        best = None
        for c in lt_tunes():
            d.tune = c | (1 << 23)          # bit 23: refuse instead of falling back to the generic tiles
            t = timeit(d)
            if t is None:
                continue
            row += " lt%d/tw%d %.1f" % ((c >> 8) & 0xf, (c >> 24) & 0xf, t)
            if best is None or t < best[0]:
                best = (t, c)
        if best:
            row += " | best LT %#x %.1f us (%.0f TF) x%.2f" % (best[1], best[0], fl / best[0] / 1e6, tg / best[0])
            if "ablate" in what:
                for nm, base in (("LT", best[1] | (1 << 23)), ("gen", gen[0][1])):
                    ts = []
                    for bits in (1 << 17, 1 << 20, (1 << 17) | (1 << 20)):
                        d.tune = base | bits
                        ts.append(timeit(d))
                    row += " | %s noloop %.1f noepi %.1f neither %.1f" % ((nm,) + tuple(ts))
        print(row, flush=True)
