#!/bin/bash
# SQ counter passes over bench.py (MFMA busy / ops, LDS bank conflicts, wave wait states), kernel trace only (gpurun refuses
# --pmc together with other trace domains).  Counter names are taken from `rocprofv3 -L` on the box: wished-for names that
# this rocprofv3 does not list are dropped.  Output: gpurun_out/pmc_sq{1,2}/ + gpurun_out/pmc_sq_summary.{json,txt}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/counters_avail.txt 2>&1
pick() { python - "$@" <<'PY'
import re, sys
avail = set(re.findall(r"\b(SQ_[A-Z0-9_]+|GRBM_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TCC_[A-Z0-9_]+)\b", open("gpurun_out/counters_avail.txt").read()))
print(" ".join([c for c in sys.argv[1:] if c in avail][:8]))
PY
}
P1=$(pick SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32)
P2=$(pick SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM)
echo "pass 1: $P1"; echo "pass 2: $P2"
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d gpurun_out/pmc_sq1 -o s1 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_sq1.log 2>&1
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -o s2 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_sq2.log 2>&1
python tools/pmc_sq_summary.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq_summary.json > gpurun_out/pmc_sq_summary.txt 2>&1
cat gpurun_out/pmc_sq_summary.txt
tail -3 gpurun_out/pmc_sq1.log | cut -c1-200
find gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 -name "*counter_collection.csv" -size +30M -delete
