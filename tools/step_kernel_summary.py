"""Per-kernel-family durations of ONE train step from a rocprofv3 kernel trace (csv): the launches between the last
two optimizer launches (plan-compiler autotune trials and warm-up steps are excluded).  Prints and optionally writes
a json {family: {n, total_us, avg_us}} - the table `roofline.frac` can be recomputed from.
    python tools/step_kernel_summary.py gpurun_out/prof/r2_kernel_trace.csv [profiles/r02_step_kernels.json]"""
import collections
import csv
import json
import re
import sys


def family(name):
    name = re.sub(r"^void\s+", "", name).replace("(anonymous namespace)::", "")
    name = re.sub(r"[<(].*$", "", name)
    return name


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) for r in rows)
    opt = [i for i, e in enumerate(ev) if "adam_kernel" in e[2] or "sgd_kernel" in e[2]]
    # round 4: two optimizer launches per step (deep layers early on a side stream, the rest behind the backward): the step
    # ends with the SMALLER one
    grids = sorted({ev[i][4] for i in opt})
    if len(grids) > 1:
        opt = [i for i in opt if ev[i][4] == grids[0]]
    ev = [e[:4] for e in ev]
    a0, a1 = opt[-2], opt[-1]
    seg = ev[a0 + 1:a1 + 1]
    T = (ev[a1][1] - ev[a0][1]) / 1e3
    fam = collections.OrderedDict()
    for s, e, n, st in seg:
        d = fam.setdefault(family(n), {"n": 0, "total_us": 0.0})
        d["n"] += 1
        d["total_us"] += (e - s) / 1e3
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from code_sha import code_sha
    out = {"code_sha": code_sha(), "step_us": round(T, 1), "launches": len(seg), "sum_kernel_us": round(sum(d["total_us"] for d in fam.values()), 1),
           "families": {}}
    print("one step: %.1f us wall, %d launches, sum of kernel durations %.1f us" % (T, len(seg), out["sum_kernel_us"]))
    for k, d in sorted(fam.items(), key=lambda kv: -kv[1]["total_us"]):
        d["total_us"] = round(d["total_us"], 1)
        d["avg_us"] = round(d["total_us"] / d["n"], 2)
        out["families"][k] = d
        print("%-44s n=%5d total %9.1f us  avg %8.2f us  %5.1f%%" % (k[:44], d["n"], d["total_us"], d["avg_us"],
                                                                     100 * d["total_us"] / out["sum_kernel_us"]))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
