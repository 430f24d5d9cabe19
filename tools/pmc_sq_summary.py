"""Per-kernel-family summary of SQ counters from `rocprofv3 --pmc ... --kernel-trace` passes over bench.py (every dispatch
runs alone under counter collection, so these are ISOLATED-kernel figures): MFMA utilisation, LDS bank conflicts, where the
waves wait.  One step is isolated as in tools/pmc_summary.py (the dispatches between the last two optimizer launches).

    python tools/pmc_sq_summary.py <dir> [<dir> ...] <out.json>

Derived columns (MI355X_MICROARCH.md "rocprofv3 PMC slots"; 256 CUs x 4 SIMDs):
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)      share of SIMD-cycles with the matrix pipe busy
  mfma_util   = MFMA flops implied by SQ_INSTS_VALU_MFMA_MOPS_BF16 (x512 flop) / (peak flop/clk/CU * SQ_BUSY_CU_CYCLES)
  lds_conf    = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                 share of LDS-array cycles lost to bank conflicts
  wait / issue_stall / active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
"""
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)


def family(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("<")[0].split("(")[0].strip()


def collect(d):
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Dispatch_Id"]), family(r["Kernel_Name"]), r["Counter_Name"], float(r["Counter_Value"]), int(r.get("Grid_Size", 0) or 0)))
    rows.sort()
    # round 4: two optimizer launches per step, enqueued back to back: a step ends with the SMALLER one
    grids = sorted({r[4] for r in rows if r[1] in ("adam_kernel", "sgd_kernel")})
    ids = sorted({r[0] for r in rows if r[1] in ("adam_kernel", "sgd_kernel") and (len(grids) < 2 or r[4] == grids[0])})
    rows = [r[:4] for r in rows]
    lo, hi = (ids[-2], ids[-1]) if len(ids) >= 2 else (-1, 1 << 62)
    out = {}
    for did, fam, cn, val in rows:
        if not (lo < did <= hi):
            continue
        a = out.setdefault(fam, {})
        a[cn] = a.get(cn, 0.0) + val
        a.setdefault("_ids_" + cn, set()).add(did)
    for fam, a in out.items():
        n = max((len(v) for k, v in a.items() if k.startswith("_ids_")), default=0)
        for k in [k for k in a if k.startswith("_ids_")]:
            del a[k]
        a["dispatches"] = n
    return out


def main():
    dirs, outp = sys.argv[1:-1], sys.argv[-1]
    fams = {}
    for d in dirs:
        for fam, a in collect(d).items():
            fams.setdefault(fam, {}).update(a)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from code_sha import code_sha
    res = {"code_sha": code_sha(), "families": {}}
    print("%-30s %5s %9s %9s %9s %7s %7s %7s" % ("kernel family", "n", "mfma_busy", "mfma_util", "lds_conf", "wait", "stall", "active"))

    def ratio(a, num, den, scale=1.0):
        return scale * a[num] / a[den] if num in a and a.get(den) else None

    def fmt(v):
        return "   -   " if v is None else "%7.3f" % v
    order = sorted(fams.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", kv[1].get("SQ_WAVE_CYCLES", 0.0)))
    for fam, a in order:
        d = dict(a)
        d["mfma_busy"] = ratio(a, "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", 0.25)
        mops = a.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) + a.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
        # one MOPS unit = 512 flop (rocprof derived-metric convention); peak 4096 bf16 flop/clk/CU (2.5 PF / 256 CUs / 2.4 GHz)
        d["mfma_util"] = (mops * 512.0 / (4096.0 * a["SQ_BUSY_CU_CYCLES"])) if a.get("SQ_BUSY_CU_CYCLES") and mops else None
        d["lds_conf"] = ratio(a, "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
        d["wait"] = ratio(a, "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
        d["issue_stall"] = ratio(a, "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
        d["active"] = ratio(a, "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")
        d["lds_issue_stall"] = ratio(a, "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES")
        res["families"][fam] = d
        if d["dispatches"] and (d["mfma_busy"] is not None or d["wait"] is not None):
            print("%-30s %5d %9s %9s %9s %s %s %s" % (fam[:30], d["dispatches"], fmt(d["mfma_busy"]), fmt(d["mfma_util"]), fmt(d["lds_conf"]),
                                                     fmt(d["wait"]), fmt(d["issue_stall"]), fmt(d["active"])))
    with open(outp, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
