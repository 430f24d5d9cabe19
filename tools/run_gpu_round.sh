#!/bin/bash
# One GPU-box session: full GPU test suite, smoke, bench, rocprofv3 kernel-trace summary, PMC passes (HBM traffic).
# Outputs under gpurun_out/ (copy what should be judged into profiles/).
set -x
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -f gpurun_out/ap_64pair.json
DYK_AP_JSON=$PWD/gpurun_out/ap_64pair.json python -m pytest tests -m gpu -q 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | tail -60 > gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
python bench.py --steps 20 --warmup 5 --dump-cmds gpurun_out/cmds_c3.json > gpurun_out/bench.json 2> gpurun_out/bench.err
python tools/cmd_roofline.py gpurun_out/cmds_c3.json > gpurun_out/cmd_roofline_c3.txt 2>&1
DYK_ROOFLINE_TOP=1000 python tools/cmd_roofline.py gpurun_out/cmds_c3.json > gpurun_out/cmd_roofline_c3_full.txt 2>&1
tail -c 3000 gpurun_out/bench.json
# the other BASELINE configs (parity-test cases; kept beside the bench line for reference)
python bench.py --mode eval --cfg kaist_dyolov3_add_sl --dtype fp32 --steps 10 --warmup 3 > gpurun_out/bench_eval_c2.json 2>/dev/null
python bench.py --mode eval --steps 10 --warmup 3 > gpurun_out/bench_eval_c3.json 2>/dev/null
# the declared AP-parity path (fp32 evaluation of the target cfg) gets its throughput number too
python bench.py --mode eval --dtype fp32 --steps 10 --warmup 3 > gpurun_out/bench_eval_c3_fp32.json 2>/dev/null
python bench.py --cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --dump-cmds gpurun_out/cmds_c5.json > gpurun_out/bench_c5.json 2>/dev/null
python tools/cmd_roofline.py gpurun_out/cmds_c5.json > gpurun_out/cmd_roofline_c5.txt 2>&1
python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --dump-cmds gpurun_out/cmds_b1.json > gpurun_out/bench_b1.json 2>/dev/null
python tools/cmd_roofline.py gpurun_out/cmds_b1.json > gpurun_out/cmd_roofline_b1.txt 2>&1
rm -f gpurun_out/cmds_c3.json gpurun_out/cmds_c5.json gpurun_out/cmds_b1.json
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof/bench_under_prof.json 2> gpurun_out/prof/err.log
# the same plan on ONE stream: every kernel alone on the chip, the rocprofv3 counterpart of bench.py's live isolated figure
rm -rf gpurun_out/prof_serial && mkdir -p gpurun_out/prof_serial
DYK_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_serial -o s1 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof_serial/bench_under_prof.json 2> gpurun_out/prof_serial/err.log
python tools/step_kernel_summary.py $(ls gpurun_out/prof_serial/s1_kernel_trace.csv gpurun_out/prof_serial/*/s1_kernel_trace.csv 2>/dev/null | head -1) gpurun_out/serial_kernels.json > gpurun_out/serial_kernels.txt 2>&1
find gpurun_out/prof_serial -name "*kernel_trace.csv" -delete
# counters in their own runs, one pass per counter (TCC slots), kernel trace only
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_write.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_summary.json > gpurun_out/pmc_summary.txt 2>&1
bash tools/run_pmc_sq.sh > gpurun_out/pmc_sq_run.log 2>&1
cat gpurun_out/pmc_summary.txt
# one step's kernels by family (durations as they ran in the step, autotune trials excluded) and per-stream occupancy
python tools/step_kernel_summary.py gpurun_out/prof/*/r3_kernel_trace.csv gpurun_out/step_kernels.json > gpurun_out/step_kernels.txt 2>&1 || python tools/step_kernel_summary.py gpurun_out/prof/r3_kernel_trace.csv gpurun_out/step_kernels.json > gpurun_out/step_kernels.txt 2>&1
head -20 gpurun_out/step_kernels.txt
python tools/trace_timeline.py $(ls gpurun_out/prof/r3_kernel_trace.csv gpurun_out/prof/*/r3_kernel_trace.csv 2>/dev/null | head -1) > gpurun_out/step_timeline.txt 2>&1
# keep the merge small: drop the per-dispatch traces
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*counter_collection.csv" -size +20M -delete
ls -R gpurun_out/prof | head -30
