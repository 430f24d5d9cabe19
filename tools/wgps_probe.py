"""Round 6: the 1x1 weight gradients of the target cfg (B = 16), per-tap kernel (conv_wgrad.hip) against the pixel-streaming
kernel (conv_wgrad_ps.hip): every ring depth / tile cap x split counts, plane mode (as the step runs them), timed warm
(back to back) and cold (a 1 GB fill between launches).  Usage: python tools/wgps_probe.py [quick]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd")]
import torch  # noqa: E402
from dyk import lib as L  # noqa: E402
from dyk import ops  # noqa: E402

SHAPES = [  # (H, W, Cin, Cout, launches per step)
    (64, 80, 128, 128, 18), (32, 40, 256, 256, 18), (32, 40, 512, 256, 11), (16, 20, 1024, 512, 10), (64, 80, 256, 128, 8),
    (16, 20, 512, 512, 10), (256, 320, 64, 64, 6), (128, 160, 64, 64, 6), (128, 160, 128, 64, 4), (256, 320, 128, 64, 2),
    (128, 160, 128, 128, 2), (16, 20, 1024, 1024, 2), (16, 20, 2048, 512, 1), (64, 80, 256, 256, 2), (32, 40, 512, 512, 2),
]
B = 16
lib = L.load()
flush = torch.empty(256 << 20, dtype=torch.float32, device="cuda")


def timed(d, reps, cold):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.dyk_conv_wgrad(ctypes.byref(d), st), "warm-up")
    if not cold:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.dyk_conv_wgrad(ctypes.byref(d), st)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    tot = 0.0
    for _ in range(reps):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.dyk_conv_wgrad(ctypes.byref(d), st)
        e1.record()
        e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


def main():
    quick = "quick" in sys.argv
    PS = 3 << 28
    total = {"old": 0.0, "new": 0.0}
    for (H, W, Cin, Cout, n) in SHAPES:
        x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
        dy = torch.randn(B, H, W, Cout, device="cuda").bfloat16()
        plane = Cout * Cin
        G = torch.zeros(plane, device="cuda")
        part = torch.empty(600 * plane if plane * 600 * 4 < (3 << 30) else 128 * plane, device="cuda")
        maxp = part.numel() // plane
        d = L.DykWgradDesc()
        d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), G.data_ptr()
        d.dtype = ops.dtype_code(torch.bfloat16)
        d.ldx, d.lddy = Cin, Cout
        d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, Cin, H, W, Cout
        d.isy = d.isx = 1
        d.ntaps = 1
        mb = B * H * W * (Cin + Cout) * 2 / 1e6
        rows = []
        olds = [2, 3, 2 | (2 << 8), 2 | (1 << 24), 3 | (1 << 24), 2 | (2 << 8) | (1 << 24)]
        news = [PS | 2, PS | 3, PS | 4, PS | 6, PS | 4 | (1 << 8), PS | 8 | (1 << 8), PS | 8 | (1 << 8) | (1 << 12)]
        for fam, tunes in (("old", olds), ("new", news)):
            for tune in tunes:
                d.tune, d.part, d.part_stride, d.splits = tune, None, 0, 0
                auto = lib.dyk_conv_wgrad_splits(ctypes.byref(d))
                opts = sorted({auto, max(1, auto // 2), max(1, auto // 4), max(1, auto // 8)} | ({2 * auto} if fam == "new" else set()))
                for o in opts:
                    d.tune, d.part, d.part_stride, d.splits = tune, None, 0, o
                    sp = lib.dyk_conv_wgrad_splits(ctypes.byref(d))
                    if sp < 1 or sp > maxp:
                        continue
                    if sp >= 2:
                        d.part, d.part_stride, d.splits = part.data_ptr(), plane, sp
                    tw = timed(d, 5, False)
                    tc = timed(d, 3, True) if not quick else tw
                    fold = (sp + 1) * plane * 4 / 2.7e6 if sp >= 2 else 0.0     # us at 2.7 TB/s (the tuner's fold model, weight 1)
                    rows.append((tc + fold, tc, tw, fold, fam, tune, sp))
        rows.sort()
        best = {f: min(r for r in rows if r[4] == f) for f in ("old", "new")}
        print("%3dx%-3d c%d>%d  %.0f MB  x%d" % (H, W, Cin, Cout, mb, n))
        for r in rows[:6] + [best["old"]]:
            print("    %-3s tune %#10x sp %3d: cold %6.1f us (%.2f TB/s) warm %6.1f  fold %5.1f  cold+fold %6.1f" % (
                r[4], r[5], r[6], r[1], mb / r[1], r[2], r[3], r[0]))
        for f in ("old", "new"):
            total[f] += best[f][0] * n
        sys.stdout.flush()
    print("sum over the step's 1x1 launches (cold + fold model): per-tap kernel %.0f us, pixel-streaming %.0f us" % (total["old"], total["new"]))


if __name__ == "__main__":
    main()
