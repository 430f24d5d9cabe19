#!/bin/bash
# LDS counters of the row-block weight gradient per layer shape (tools/rb_probe.py under rocprofv3 --pmc): bank-conflict share
# of the LDS cycles, LDS share of the wave cycles.  Output: gpurun_out/rb_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/rb_pmc && mkdir -p gpurun_out/rb_pmc
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/rb_pmc -o r -- python tools/rb_probe.py ${1:-7} > gpurun_out/rb_pmc.log 2>&1
python - <<'PY' > gpurun_out/rb_pmc.txt
import csv, glob, collections
f = glob.glob("gpurun_out/rb_pmc/**/r_counter_collection.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    if "wgrad" not in k: continue
    key = (k, r["Grid_Size"], r.get("LDS_Block_Size", ""))
    a = agg.setdefault(key, collections.Counter())
    a[r["Counter_Name"]] += float(r["Counter_Value"]); a["n_" + r["Counter_Name"]] += 1
for (k, g, l), a in agg.items():
    conf, act = a["SQ_LDS_BANK_CONFLICT"], a["SQ_LDS_IDX_ACTIVE"]
    print("%-58s grid %-8s lds %-7s launches %3d  conflict/active %.3f  lds-active/busy-cu %.3f" % (k, g, l, a["n_SQ_LDS_IDX_ACTIVE"], conf / max(act, 1), act / max(a["SQ_BUSY_CU_CYCLES"], 1)))
PY
cat gpurun_out/rb_pmc.txt
find gpurun_out/rb_pmc -name "*.csv" -size +5M -delete
