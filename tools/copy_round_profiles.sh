#!/bin/bash
# gpurun_out/ (scratch, written by tools/run_gpu_round.sh) -> profiles/<tag>_* (tracked): the files the judge reads
T=${1:-r06}
cd "$(dirname "$0")/.."
cp gpurun_out/bench.json profiles/${T}_bench.json
cp gpurun_out/bench_b1.json profiles/${T}_bench_b1.json
cp gpurun_out/bench_c5.json profiles/${T}_bench_mobilenetv3_b32.json
cp gpurun_out/bench_eval_c3.json profiles/${T}_bench_eval_target_bf16.json
cp gpurun_out/bench_eval_c2.json profiles/${T}_bench_eval_dyolov3_add_sl_fp32.json
cp gpurun_out/bench_eval_c3_fp32.json profiles/${T}_bench_eval_target_fp32.json
cp gpurun_out/ap_64pair.json profiles/${T}_ap_64pair.json 2>/dev/null
cp gpurun_out/serial_kernels.txt profiles/${T}_serial_kernels.txt 2>/dev/null; cp gpurun_out/serial_kernels.json profiles/${T}_serial_kernels.json 2>/dev/null
for c in c3 c5 b1 c3_full; do cp gpurun_out/cmd_roofline_$c.txt profiles/${T}_cmd_roofline_$c.txt; done
for f in pmc_summary.txt pmc_summary.json pmc_sq_summary.txt pmc_sq_summary.json step_kernels.txt step_kernels.json step_timeline.txt; do cp gpurun_out/$f profiles/${T}_$f; done
cp gpurun_out/prof/r3_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv 2>/dev/null
python - <<PY
import json
for f in ("bench", "bench_mobilenetv3_b32", "bench_b1", "bench_eval_target_bf16"):
    d = json.loads(open("profiles/${T}_%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 2), "ms", round(d["value"], 1), d.get("roofline", {}).get("frac_is", "")[:60])
print(open("profiles/${T}_pmc_summary.txt").readline().strip())
print(open("profiles/${T}_step_kernels.txt").readline().strip())
print("code hash", json.load(open("profiles/${T}_step_kernels.json"))["code_sha"], json.load(open("profiles/${T}_pmc_summary.json"))["code_sha"])
PY
python tools/code_sha.py
