#!/bin/bash
# SQ counters of the depthwise kernels on the MobileNetV3 cfg's shapes (tools/dw_probe.py under rocprofv3 --pmc)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/dwpmc && mkdir -p gpurun_out/dwpmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/dwpmc -o a -- python tools/dw_probe.py > gpurun_out/dwpmc/probe.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/dwpmc/**/a_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, c in agg.items():
    if "dwconv" not in k: continue
    w = c.get("SQ_WAVE_CYCLES", 1)
    print("%-44s n=%4d wait %.2f stall %.2f active %.2f | valu %.2f lds %.2f vmem %.2f" % (k[:44], cnt[(k, "SQ_WAVE_CYCLES")], c["SQ_WAIT_ANY"] / w, c["SQ_WAIT_INST_ANY"] / w,
          c["SQ_ACTIVE_INST_ANY"] / w, c["SQ_ACTIVE_INST_VALU"] / w, c["SQ_ACTIVE_INST_LDS"] / w, c["SQ_ACTIVE_INST_VMEM"] / w))
PY
rm -rf gpurun_out/dwpmc
