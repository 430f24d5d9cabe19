"""Hardware probes + conv micro-benchmarks, run on the GPU box.  Writes gpurun_out/probe.json.
Usage: python tools/gpu_probe.py [tr16] [mfma] [convbench]"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "double-yolo-kaist_amd")
sys.path[:0] = [ROOT, PKG]
import torch  # noqa: E402

from dyk import ops  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
res = {}
what = sys.argv[1:] or ["tr16", "mfma", "convbench"]
probe = ctypes.CDLL(os.path.join(PKG, "csrc", "libdyk_probe.so"))


def tr16(addrs):
    a = torch.tensor(addrs, dtype=torch.int32, device="cuda")
    o = torch.zeros(256, dtype=torch.int16, device="cuda")
    rc = probe.dyk_probe_tr16(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(o.data_ptr()), None)
    torch.cuda.synchronize()
    assert rc == 0
    return o.cpu().view(64, 4).tolist()


if "tr16" in what:
    # LDS holds u16 element index i at byte 2*i.  Pattern A: lane l -> byte 8*l (contiguous).
    res["tr16_contig8"] = tr16([8 * l for l in range(64)])
    # Pattern B: rows of 64 elements (128 B): lane l -> row (l%16)//4 ... see analysis in DESIGN
    res["tr16_row128_q4"] = tr16([((l % 16) // 4) * 128 + (l % 4) * 8 + (l // 16) * 512 for l in range(64)])
    # Pattern C: lane l -> row l%16 (128 B rows), col block l//16
    res["tr16_row128_l16"] = tr16([(l % 16) * 128 + (l // 16) * 8 for l in range(64)])
    # Pattern D: rows of 32 B (16 elements) : lane -> row (l%16)//4, 8-byte piece l%4, group l//16 -> +128 B
    res["tr16_row32"] = tr16([((l % 16) // 4) * 32 + (l % 4) * 8 + (l // 16) * 128 for l in range(64)])

if "mfma" in what:
    g = torch.Generator().manual_seed(0)
    A = torch.randint(-4, 5, (16, 32), generator=g).float()
    B = torch.randint(-4, 5, (32, 16), generator=g).float()
    D = torch.zeros(16, 16, device="cuda")
    Ad, Bd = A.cuda(), B.cuda()
    probe.dyk_probe_mfma_layout(ctypes.c_void_p(Ad.data_ptr()), ctypes.c_void_p(Bd.data_ptr()),
                                ctypes.c_void_p(D.data_ptr()), None)
    torch.cuda.synchronize()
    res["mfma_layout_maxerr"] = (D.cpu() - A @ B).abs().max().item()

if "convbench" in what:
    shapes = [  # (Cin, Cout, H, W, k, s) at B=16 : the dominant C3 problems
        (128, 128, 64, 80, 3, 1), (256, 256, 32, 40, 3, 1), (512, 512, 16, 20, 3, 1), (64, 64, 128, 160, 3, 1),
        (512, 1024, 16, 20, 3, 1), (256, 512, 32, 40, 3, 1), (128, 256, 64, 80, 3, 1), (32, 64, 256, 320, 3, 1),
        (512, 256, 64, 80, 3, 1), (1024, 512, 32, 40, 3, 1), (2048, 1024, 16, 20, 3, 1),
        (64, 128, 256, 320, 3, 2), (256, 512, 64, 80, 3, 2),
        (128, 128, 64, 80, 1, 1), (256, 256, 32, 40, 1, 1), (512, 256, 32, 40, 1, 1), (1024, 512, 16, 20, 1, 1),
        (64, 64, 256, 320, 1, 1),
    ]
    rows = []
    for dt in (torch.bfloat16, torch.float32):
        for (ci, co, H, W, k, s) in shapes:
            if dt == torch.float32 and H * W > 64 * 80:
                continue
            B = 16
            x = torch.randn(B, H, W, ci, device="cuda").to(dt)
            w = torch.randn(co, ci, k, k, device="cuda") * 0.05
            wp = ops.pack_weight(w, dt)
            Ho, Wo = ops.conv_out_size(H, k, s, k // 2), ops.conv_out_size(W, k, s, k // 2)
            out = torch.empty(B, Ho, Wo, co, device="cuda", dtype=dt)
            for _ in range(3):
                ops.conv2d_fwd(x, wp, k, s, k // 2, co, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                ops.conv2d_fwd(x, wp, k, s, k // 2, co, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            fl = 2.0 * B * Ho * Wo * co * ci * k * k
            by = (x.numel() + out.numel() + wp.numel()) * x.element_size()
            rows.append(dict(dtype=str(dt), cin=ci, cout=co, H=H, W=W, k=k, s=s, ms=ms, tflops=fl / ms / 1e9,
                             gbps=by / ms / 1e6))
            print(rows[-1], flush=True)
    res["convbench"] = rows

if "ablate" in what:
    import ctypes as C
    from dyk import lib as L
    rows = []
    shapes = [(128, 128, 64, 80, 1), (64, 64, 256, 320, 1), (256, 256, 32, 40, 1), (128, 128, 64, 80, 3)]
    if os.environ.get("PROBE_EARLY"):
        shapes = [(32, 64, 256, 320, 3), (64, 32, 256, 320, 3), (64, 64, 256, 320, 1), (64, 32, 256, 320, 1)]
    if os.environ.get("PROBE_MID"):
        shapes = [(256, 256, 32, 40, 3), (128, 128, 64, 80, 3), (512, 512, 16, 20, 3), (256, 256, 32, 40, 1)]
    for (ci, co, H, W, k) in shapes:
        B, dt = 16, torch.bfloat16
        x = torch.randn(B, H, W, ci, device="cuda").to(dt)
        w = torch.randn(co, ci, k, k, device="cuda") * 0.05
        wp = ops.pack_weight(w, dt)
        out = torch.empty(B, H, W, co, device="cuda", dtype=dt)
        stats = torch.zeros(64 * co, dtype=torch.float64, device="cuda")
        for name, tune, st in [("t160", 64 | (2 << 8) | (2 << 12), None), ("t160 stats", 64 | (2 << 8) | (2 << 12), stats), ("t160 noloop", 64 | (2 << 8) | (2 << 12) | (1 << 17), None), ("t160 tables", 64 | (2 << 8) | (2 << 12) | (1 << 19), None), ("t160 noepi", 64 | (2 << 8) | (2 << 12) | (1 << 20), None),("default", 0, None), ("stats", 0, stats), ("nostore", 1 << 16, None), ("noloop", 1 << 17, None),
                               ("noloop+nostore", 3 << 16, None), ("bkb128", 128 | (2 << 8), None), ("bkb64p3", 64 | (3 << 8), None),
                               ("empty", 1 << 18, None), ("tables", 1 << 19, None), ("noepi", 1 << 20, None), ("noloop+noepi", (1 << 20) | (1 << 17), None)]:
            d = ops.make_conv_desc(x, wp, out, Hi=H, Wi=W, Cin=ci, Cout=co, Hg=H, Wg=W, Ho=H, Wo=W, taps=ops.fwd_taps(k, k // 2), stats=st)
            d.tune = tune
            d.stats_slots = 32 if st is not None else 0
            fn = L.load().dyk_conv_igemm
            for _ in range(3):
                fn(C.byref(d), None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn(C.byref(d), None)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            by = (x.numel() + out.numel()) * 2
            rows.append((ci, co, H, k, name, round(ms * 1e3, 1), round(by / ms / 1e6)))
            print(rows[-1], flush=True)
    res["ablate"] = rows

if "clsablate" in what:
    # stride-2 data gradient with the four output-parity classes in one launch: where does the time go?
    import ctypes as C
    from dyk import lib as L
    fn = L.load().dyk_conv_igemm
    for (cf_in, cf_out, Hi, Wi) in [(32, 64, 512, 640), (64, 128, 256, 320), (128, 256, 128, 160)]:
        B, dt = 16, torch.bfloat16
        Ho, Wo = Hi // 2, Wi // 2
        dy = torch.randn(B, Ho, Wo, cf_out, device="cuda").to(dt)
        w = torch.randn(cf_out, cf_in, 3, 3, device="cuda") * 0.05
        wpt = ops.pack_weight(w, dt, transposed=True)
        out = torch.empty(B, Hi, Wi, cf_in, device="cuda", dtype=dt)
        classes = ops.dgrad_classes(3, 1, 2, Hi, Wi)
        d = ops.make_conv_desc(dy, wpt, out, Hi=Ho, Wi=Wo, Cin=cf_out, Cout=cf_in, Hg=classes[0][2], Wg=classes[0][3], Ho=Hi, Wo=Wi,
                               taps=[t for c in classes for t in c[4]], osy=2, osx=2)
        d.ncls, q0 = len(classes), 0
        for c, (py, px, _, _, taps) in enumerate(classes):
            d.cls_first[c], d.cls_ntaps[c], d.cls_ooy[c], d.cls_oox[c] = q0, len(taps), py, px
            q0 += len(taps)
        line = []
        for base in (0x2340, 0x2240, 0x2280, 0x240):
            for name, bits in [("full", 0), ("nostore", 1 << 16), ("noloop", 1 << 17), ("empty", 1 << 18), ("tables", 1 << 19), ("noepi", 1 << 20), ("noloop+noepi", (1 << 20) | (1 << 17))]:
                d.tune = base | bits
                for _ in range(2):
                    fn(C.byref(d), None)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn(C.byref(d), None)
                e1.record(); torch.cuda.synchronize()
                line.append("%s=%.1f" % (name, e0.elapsed_time(e1) / 10 * 1e3))
            print((cf_in, cf_out, Hi), hex(base), " ".join(line), flush=True)
            line = []

if "nobar" in what:
    # upper bound of a barrier-free K loop: the autotuned best configuration with / without its per-step s_barrier
    import ctypes as C
    from dyk import lib as L
    from dyk.plan import _conv_candidates
    for (ci, co, H, W, k) in [(128, 128, 64, 80, 3), (256, 256, 32, 40, 3), (512, 512, 16, 20, 3), (128, 128, 64, 80, 1), (512, 256, 32, 40, 1), (64, 64, 256, 320, 1)]:
        B, dt = 16, torch.bfloat16
        x = torch.randn(B, H, W, ci, device="cuda").to(dt)
        w = torch.randn(co, ci, k, k, device="cuda") * 0.05
        wp = ops.pack_weight(w, dt)
        out = torch.empty(B, H, W, co, device="cuda", dtype=dt)
        stats = torch.zeros(64 * co, dtype=torch.float64, device="cuda")
        d = ops.make_conv_desc(x, wp, out, Hi=H, Wi=W, Cin=ci, Cout=co, Hg=H, Wg=W, Ho=H, Wo=W, taps=ops.fwd_taps(k, k // 2), stats=stats)
        d.stats_slots = 32
        fn = L.load().dyk_conv_igemm
        def t(tune, n=10):
            d.tune = tune
            for _ in range(3):
                fn(C.byref(d), None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn(C.byref(d), None)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        cands = [c for c in _conv_candidates(d) if ((c >> 12) & 0xf) < 3 and not (c >> 28)]
        best = min(cands, key=t)
        fl = 2.0 * B * H * W * ci * co * k * k
        tb, tn, te, tl = t(best), t(best | (1 << 22)), t(best | (1 << 20)), t(best | (1 << 17))
        print((ci, co, H, k), "best tune %x: %.1f us (%.0f TF) | no barrier %.1f us (%.0f TF) | no epilogue %.1f us | no K loop %.1f us" % (
            best, tb, fl / tb / 1e6, tn, fl / tn / 1e6, te, tl), flush=True)

if "tunes" in what:
    import ctypes as C
    from dyk import lib as L
    from dyk.plan import _conv_candidates
    rows = []
    for (ci, co, H, W, k) in [(512, 512, 16, 20, 3), (256, 256, 32, 40, 3), (128, 128, 64, 80, 3), (1024, 512, 16, 20, 1), (256, 256, 32, 40, 1)]:
        B, dt = 16, torch.bfloat16
        x = torch.randn(B, H, W, ci, device="cuda").to(dt)
        w = torch.randn(co, ci, k, k, device="cuda") * 0.05
        wp = ops.pack_weight(w, dt)
        out = torch.empty(B, H, W, co, device="cuda", dtype=dt)
        stats = torch.zeros(64 * co, dtype=torch.float64, device="cuda")
        d = ops.make_conv_desc(x, wp, out, Hi=H, Wi=W, Cin=ci, Cout=co, Hg=H, Wg=W, Ho=H, Wo=W, taps=ops.fwd_taps(k, k // 2), stats=stats)
        d.stats_slots = 32
        fn = L.load().dyk_conv_igemm
        res_ = []
        for tune in _conv_candidates(d):
            d.tune = tune
            for _ in range(3):
                fn(C.byref(d), None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn(C.byref(d), None)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            res_.append((round(ms * 1e3, 1), "bkb%d p%d tile%d bm%d" % (tune & 0xff, (tune >> 8) & 0xf, (tune >> 12) & 0xf, (tune >> 24) & 0xf)))
        res_.sort()
        fl = 2.0 * B * H * W * ci * co * k * k
        print((ci, co, H, k), "best", res_[:6], "worst", res_[-2:], "best TF %.0f" % (fl / res_[0][0] / 1e6), flush=True)
        rows.append(((ci, co, H, k), res_))
    res["tunes"] = rows

if "bnbench" in what:
    import ctypes as C
    from dyk import lib as L
    rows = []
    for (H, W, c) in [(64, 80, 128), (32, 40, 256), (128, 160, 64), (16, 20, 1024), (256, 320, 64)]:
        B, dt = 16, torch.bfloat16
        y = torch.randn(B, H, W, c, device="cuda").to(dt)
        dz = torch.randn(B, H, W, c, device="cuda").to(dt)
        z = torch.empty_like(y)
        vec = [torch.rand(c, device="cuda") + 0.5 for _ in range(4)]
        red = torch.zeros(32 * 2 * c, dtype=torch.float64, device="cuda")
        out = []
        for name, fn, d in [("fwd", "dyk_bn_act_fwd", ops.ew_desc(a=y, out=z, act="mish", p0=vec[0], p1=vec[1])),
                            ("reduce", "dyk_bn_act_bwd_reduce", ops.ew_desc(a=dz, b=y, act="mish", p0=vec[0], p1=vec[1], p2=vec[2], p3=vec[3], red=red)),
                            ("apply", "dyk_bn_act_bwd_apply", ops.ew_desc(a=dz, b=y, out=z, act="mish", p0=vec[0], p1=vec[1], p2=vec[2], p3=vec[3], red=red))]:
            d.slots = 32
            f = getattr(L.load(), fn)
            for _ in range(3):
                f(C.byref(d), None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f(C.byref(d), None)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            nb = y.numel() * 2 * (2 if name != "apply" else 3)
            out.append("%s %.1fus %.0fGB/s" % (name, ms * 1e3, nb / ms / 1e6))
        print((H, W, c), "  ".join(out), flush=True)
    res["bnbench"] = rows

if "bnfold" in what:
    # where does a launch-bound BatchNorm pass spend its time?  fused finalize + normalise with 1 / 8 / 32 statistics replicas,
    # the plain normalise pass (no fold) and an empty launch, on the small tensors of the deep stages
    import ctypes as C
    from dyk import lib as L
    lib = L.load()
    def timed(f, n=40):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for (H, W, c) in [(32, 40, 256), (64, 80, 128), (16, 20, 512), (128, 160, 64)]:
        B, dt = 16, torch.bfloat16
        y = torch.randn(B, H, W, c, device="cuda").to(dt)
        dz = torch.randn(B, H, W, c, device="cuda").to(dt)
        z = torch.empty_like(y)
        vec = [torch.rand(c, device="cuda") + 0.5 for _ in range(8)]
        out = []
        d0 = ops.ew_desc(a=y, out=z, act="leaky", p0=vec[0], p1=vec[1])
        out.append("plain %.1f" % timed(lambda: lib.dyk_bn_act_fwd(C.byref(d0), None)))
        for slots in (1, 8, 32):
            stats = torch.rand(slots * 2 * c, dtype=torch.float64, device="cuda") * 1000 + 2000
            f = L.DykBnFinalizeDesc()
            f.stats, f.gamma, f.beta = stats.data_ptr(), vec[2].data_ptr(), vec[3].data_ptr()
            f.running_mean, f.running_var = vec[4].data_ptr(), vec[5].data_ptr()
            f.scale, f.shift, f.save_mean, f.save_rstd = vec[0].data_ptr(), vec[1].data_ptr(), vec[6].data_ptr(), vec[7].data_ptr()
            f.C, f.count, f.momentum, f.eps, f.slots = c, B * H * W, 0.03, 1e-4, slots
            d = ops.ew_desc(a=y, out=z, act="leaky", p0=vec[0], p1=vec[1])
            out.append("fused/%d %.1f" % (slots, timed(lambda: lib.dyk_bn_finalize_act_fwd(C.byref(f), C.byref(d), None))))
        for slots in (1, 16):
            red = torch.zeros(32 * 2 * c, dtype=torch.float64, device="cuda")
            da = ops.ew_desc(a=dz, b=y, out=z, act="leaky", p0=vec[0], p1=vec[1], p2=vec[2], p3=vec[3], red=red)
            da.slots = slots
            out.append("apply/%d %.1f" % (slots, timed(lambda: lib.dyk_bn_act_bwd_apply(C.byref(da), None))))
        print("bnfold", (H, W, c), " ".join(out), "us", flush=True)

if "bnbwdablate" in what:
    # data gradient with the fused BN-backward-reduce epilogue against the same GEMM with the plain epilogue, per activation
    import ctypes as C
    from dyk import lib as L
    from dyk.plan import _conv_candidates
    for (ci, co, H, W, k) in [(64, 32, 256, 320, 3), (128, 64, 128, 160, 3), (128, 128, 64, 80, 3), (256, 256, 32, 40, 3), (128, 128, 64, 80, 1)]:
        B, dt = 16, torch.bfloat16
        x = torch.randn(B, H, W, ci, device="cuda").to(dt)
        w = torch.randn(co, ci, k, k, device="cuda") * 0.05
        wp = ops.pack_weight(w, dt)
        out = torch.empty(B, H, W, co, device="cuda", dtype=dt)
        yprev = torch.randn(B, H, W, co, device="cuda").to(dt)
        slots = 32
        red = torch.zeros(slots * 2 * co, dtype=torch.float64, device="cuda")
        vec = [torch.rand(co, device="cuda") + 0.5 for _ in range(4)]
        fn = L.load().dyk_conv_igemm
        def mk(flags, act):
            d = ops.make_conv_desc(x, wp, out, Hi=H, Wi=W, Cin=ci, Cout=co, Hg=H, Wg=W, Ho=H, Wo=W, taps=ops.fwd_taps(k, k // 2), act=act)
            d.flags = flags
            if flags:
                d.res, d.ldr = yprev.data_ptr(), co
                d.scale, d.shift, d.aux0, d.aux1 = (t.data_ptr() for t in vec)
                d.stats, d.stats_slots = red.data_ptr(), slots
            return d
        def t(d, tune, n=10):
            d.tune = tune
            for _ in range(2):
                fn(C.byref(d), None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn(C.byref(d), None)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        line = []
        for name, flags, act in [("plain", 0, "linear"), ("bnbwd linear", L.EPI_BNBWD, "linear"), ("bnbwd leaky", L.EPI_BNBWD, "leaky"), ("bnbwd mish", L.EPI_BNBWD, "mish")]:
            d = mk(flags, act)
            cands = _conv_candidates(d)
            ts = sorted((t(d, c, 5), c) for c in cands)
            line.append("%s %.1f us (tune %x; 2nd %.1f %x)" % (name, ts[0][0], ts[0][1], ts[1][0], ts[1][1]))
        print((ci, co, H, k), " | ".join(line), flush=True)

if "wgablate" in what:
    import ctypes as C
    from dyk import lib as L
    rows = []
    for (ci, co, H, W, k) in [(128, 128, 64, 80, 3), (256, 256, 32, 40, 3), (128, 128, 64, 80, 1), (512, 512, 16, 20, 3), (64, 64, 256, 320, 1)]:
        B, dt = 16, torch.bfloat16
        x = torch.randn(B, H, W, ci, device="cuda").to(dt)
        dy = torch.randn(B, H, W, co, device="cuda").to(dt)
        dw = torch.zeros(k * k, co, ci, device="cuda")
        plane_sp = {(128, 128, 64, 3): 28, (256, 256, 32, 3): 6, (512, 512, 16, 3): 6, (128, 128, 64, 1): 80, (64, 64, 256, 1): 512}[(ci, co, H, k)]
        part = torch.zeros(plane_sp * k * k * co * ci, device="cuda")
        for name, tune, splits in [("plane", 0, -10), ("plane kg2", 2 | (2 << 8), -10), ("plane noloop", 1 << 17, -10), ("plane noatomic", 1 << 16, -10), ("default", 0, 0), ("pipe3", 3, 0), ("kg2", 2 | (2 << 8), 0), ("kg2s.5", 2 | (2 << 8), -2), ("kg2s2", 2 | (2 << 8), -3), ("noatomic", 1 << 16, 0), ("noloop", 1 << 17, 0),
                                   ("noloop+noatomic", 3 << 16, 0), ("splits16", 0, 16), ("splits32", 0, 32)]:
            d = L.DykWgradDesc()
            d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
            d.dtype = L.DYK_BF16
            d.ldx, d.lddy = ci, co
            d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, ci, H, W, co
            d.isy = d.isx = 1
            taps = ops.fwd_taps(k, k // 2)
            d.ntaps = len(taps)
            for i, (ty, tx, wt) in enumerate(taps):
                d.tdy[i], d.tdx[i], d.twt[i] = ty, tx, wt
            if splits == -10:
                d.part, d.part_stride = part.data_ptr(), k * k * co * ci
                splits = plane_sp
            elif splits < 0:       # relative to the K-grouped default: -2 = half, -3 = double
                tiles = ((co + 127) // 128) * ((ci + 127) // 128) * k * k
                base = max(1, -(-256 // tiles))
                splits = max(1, base // 2) if splits == -2 else base * 2
            d.splits, d.lddw, d.tune = splits, 0, tune
            fn = L.load().dyk_conv_wgrad
            for _ in range(3):
                fn(C.byref(d), None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn(C.byref(d), None)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            fl = 2.0 * B * H * W * ci * co * k * k
            rows.append((ci, co, H, k, name, round(ms * 1e3, 1), round(fl / ms / 1e9)))
            print(rows[-1], flush=True)
    res["wgablate"] = rows

with open(os.path.join(OUT, "probe.json"), "w") as f:
    json.dump(res, f, indent=1)
print("wrote probe.json")
