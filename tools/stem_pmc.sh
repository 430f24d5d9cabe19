#!/bin/bash
# SQ counters + durations of the stem kernels alone at the BASELINE size (tools/stem_probe.py under rocprofv3)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/stempmc && mkdir -p gpurun_out/stempmc
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stempmc -o t -- python tools/stem_probe.py > gpurun_out/stempmc/probe.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/stempmc/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "stem" in r["Name"]: print("%-60s calls %s avg %.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/stempmc -o a -- python tools/stem_probe.py >> gpurun_out/stempmc/probe.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/stempmc -o b -- python tools/stem_probe.py >> gpurun_out/stempmc/probe.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ("a", "b"):
    fs = glob.glob("gpurun_out/stempmc/**/%s_counter_collection.csv" % tag, recursive=True)
    if not fs: print("no counters", tag); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, c in agg.items():
        if "stem" not in k: continue
        n = max(cnt[(k, list(c)[0])], 1)
        print(k[:50], {q: round(v / n / 1e6, 3) for q, v in c.items()}, "(millions per launch)")
PY
tail -3 gpurun_out/stempmc/probe.log
rm -rf gpurun_out/stempmc
