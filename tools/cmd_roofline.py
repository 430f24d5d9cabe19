"""Per-command roofline table from `bench.py --dump-cmds FILE`: every command of one forward + backward pass timed ALONE on
the chip (HIP events), its algorithmic bytes / flops, and the time above max(HBM, MFMA) floor -- where a step's time goes
relative to what the hardware allows.  Usage: python tools/cmd_roofline.py gpurun_out/cmds.json [HBM_TBps] [PFLOPs]"""
import collections
import ctypes
import json
import sys


def _es(dtype_code):
    return 2 if dtype_code == 1 else 4


def cmd_model(L, op, d, plan=None):
    """(label, algorithmic HBM bytes, flops) of one command: each operand tensor once"""
    n = {getattr(L, k): k[3:] for k in dir(L) if k.startswith("OP_")}[op]
    if op == L.OP_CONV:
        es = _es(d.dtype)
        px_o = d.B * d.Hg * d.Wg
        px_i = d.B * d.Hi * d.Wi
        by = px_i * d.Cin * es + px_o * d.Cout * es + d.ntaps * d.Cin * d.Cout * es
        if d.res: by += px_o * d.Cout * es
        if d.add: by += px_o * d.Cout * es
        if d.aux0: by += px_o * d.Cout * es          # fused BN-backward epilogue reads the raw conv output of the producer
        if d.flags & 128: by += px_o * d.Cout * es   # DYK_EPI_BNFWD: the normalised tensor is written by the same launch
        fl = 2.0 * px_o * d.Cin * d.Cout * d.ntaps
        return "%s %dx%d c%d>%d t%d s%d f%x" % (n, d.Hg, d.Wg, d.Cin, d.Cout, d.ntaps, d.isy, d.flags), by, fl
    if op == L.OP_WGRAD:
        es = _es(d.dtype)
        by = d.B * d.Hi * d.Wi * d.Cin * es + d.B * d.Ho * d.Wo * d.Cout * es
        fl = 2.0 * d.B * d.Ho * d.Wo * d.Cin * d.Cout * d.ntaps
        g = max(d.group_n, 1)              # grouped launch (DykWgradDesc.group): g problems of this geometry
        return "%s %dx%d c%d>%d t%d sp%d%s" % (n, d.Ho, d.Wo, d.Cin, d.Cout, d.ntaps, d.splits, " g%d" % g if g > 1 else ""), by * g, fl * g
    if op in (L.OP_DW_FWD, L.OP_DW_DGRAD, L.OP_DW_WGRAD):
        es = _es(d.dtype)
        by = (d.B * d.Hi * d.Wi + d.B * d.Ho * d.Wo) * d.C * es
        return "%s %dx%d c%d k%d s%d" % (n, d.Hi, d.Wi, d.C, d.k, d.stride), by, 2.0 * d.B * d.Ho * d.Wo * d.C * d.k * d.k
    if op in (L.OP_BN_ACT_FWD, L.OP_BN_BWD_REDUCE, L.OP_BN_BWD_APPLY, L.OP_AXPBY, L.OP_DOT, L.OP_SE_POOL, L.OP_SE_SCALE):
        es = _es(d.dtype)
        k = sum(1 for p in (d.a, d.b, d.out) if p)
        if op == L.OP_BN_BWD_APPLY and d.p3: k += 1
        return "%s n%d c%d" % (n, d.npix, d.C), float(d.npix) * d.C * es * k, 0.0
    if op in (L.OP_UPSAMPLE_FWD, L.OP_UPSAMPLE_BWD, L.OP_MAXPOOL_FWD, L.OP_MAXPOOL_BWD):
        es = _es(d.dtype)
        return "%s %dx%d c%d" % (n, d.H, d.W, d.C), float(d.B) * d.H * d.W * d.C * es * 2, 0.0
    if op == L.OP_BN_FWD_FUSED and plan is not None:
        a = plan._desc_at[d.p[1]]
        return "%s n%d c%d" % (n, a.npix, a.C), float(a.npix) * a.C * _es(a.dtype) * (3 if a.b else 2), 0.0
    if op in (L.OP_STEM_FWD, L.OP_STEM_WGRAD):
        es = _es(d.dtype)
        return "%s %dx%d c%d" % (n, d.Ho, d.Wo, d.Cout), float(d.B) * (3 * d.H * d.W * (1 if d.in_u8 else 4) + d.Ho * d.Wo * d.Cout * es), \
            2.0 * d.B * d.Ho * d.Wo * 27 * d.Cout
    return n, 0.0, 0.0


def main():
    rows = json.load(open(sys.argv[1]))
    hbm = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0         # TB/s
    pf = float(sys.argv[3]) if len(sys.argv) > 3 else 2.5          # PFLOP/s dense bf16
    tot = sum(r["us"] for r in rows)
    fam = collections.OrderedDict()
    for r in rows:
        r["floor"] = max(r["bytes"] / (hbm * 1e6), r["flops"] / (pf * 1e9))
        k = r["label"].split()[0] + ("/" + r["pass"])
        a = fam.setdefault(k, [0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += r["us"]; a[2] += r["floor"]; a[3] += r["bytes"]; a[4] += r["flops"]
    print("isolated sum %.1f us over %d commands; floor sum %.1f us (HBM %.1f TB/s, MFMA %.2f PF)" % (
        tot, len(rows), sum(r["floor"] for r in rows), hbm, pf))
    if rows and "deps" in rows[0]:
        # longest dependent chain with the isolated durations (+ GAP us of launch / event latency per link): what the
        # step would take on infinitely many streams; the share of each family ON that chain
        for gap in (0.0, 3.0):
            onpath = collections.Counter()
            total = 0.0
            for which in ("fwd", "bwd"):
                rs = [r for r in rows if r["pass"] == which]
                fin, prev = [0.0] * len(rs), [-1] * len(rs)
                for i, r in enumerate(rs):
                    st, pv = 0.0, -1
                    for j in r["deps"]:
                        if fin[j] > st:
                            st, pv = fin[j], j
                    fin[i], prev[i] = st + r["us"] + gap, pv
                i = max(range(len(rs)), key=lambda q: fin[q])
                total += fin[i]
                while i >= 0:
                    onpath[rs[i]["label"].split()[0] + "/" + which] += rs[i]["us"]
                    i = prev[i]
            print("critical path (gap %.0f us): %.1f us;  on it: %s" % (gap, total, ", ".join(
                "%s %.0f" % kv for kv in onpath.most_common(8))))
    print("%-22s %5s %10s %10s %9s %8s %8s" % ("family", "n", "us", "floor us", "excess", "TB/s", "TF/s"))
    for k, a in sorted(fam.items(), key=lambda kv: -(kv[1][1] - kv[1][2])):
        print("%-22s %5d %10.1f %10.1f %9.1f %8.2f %8.1f" % (k, a[0], a[1], a[2], a[1] - a[2], a[3] / max(a[1], 1e-9) / 1e6,
                                                          a[4] / max(a[1], 1e-9) / 1e6))
    top = int(__import__("os").environ.get("DYK_ROOFLINE_TOP", "40"))      # (all commands: DYK_ROOFLINE_TOP=1000)
    print("\ntop %d commands by time above floor" % top)
    agg = collections.OrderedDict()
    for r in rows:
        a = agg.setdefault((r["pass"], r["label"]), [0, 0.0, 0.0, r])
        a[0] += 1; a[1] += r["us"]; a[2] += r["floor"]
    for (p, lab), a in sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:top]:
        r = a[3]
        print("%-3s %-46s n=%3d us %8.1f floor %7.1f  (%.2f TB/s, %.0f TF/s)" % (
            p, lab, a[0], a[1], a[2], r["bytes"] / max(r["us"], 1e-9) / 1e6, r["flops"] / max(r["us"], 1e-9) / 1e6))


if __name__ == "__main__":
    main()
