#!/bin/bash
# rocprofv3 kernel trace of the bench command -> per-family durations of one step + per-stream timeline
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out; rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
python -m pytest tests/test_eval_ap.py tests/test_gpu_ddp.py -m gpu -q 2>&1 | tail -120 > gpurun_out/pytest_sel.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r2 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof/bench_under_prof.json 2> gpurun_out/prof/err.log
T=$(ls gpurun_out/prof/r2_kernel_trace.csv gpurun_out/prof/*/r2_kernel_trace.csv 2>/dev/null | head -1)
python tools/step_kernel_summary.py $T gpurun_out/step_kernels.json > gpurun_out/step_kernels.txt 2>&1
head -30 gpurun_out/step_kernels.txt
python tools/trace_timeline.py $T > gpurun_out/step_timeline.txt 2>&1
tail -3 gpurun_out/step_timeline.txt
cp $T gpurun_out/prof/r2_kernel_trace.csv 2>/dev/null
find gpurun_out/prof -name "*.db" -delete
