#!/bin/bash
# one rocprofv3 kernel trace of the C3 step: in-step kernel table + per-stream timeline
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1     # (fills the tune cache: the traced run does not tune)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof/bench_under_prof.json 2> gpurun_out/prof/err.log
T=$(ls gpurun_out/prof/r3_kernel_trace.csv gpurun_out/prof/*/r3_kernel_trace.csv 2>/dev/null | head -1)
python tools/step_kernel_summary.py $T gpurun_out/step_kernels.json > gpurun_out/step_kernels.txt 2>&1
python tools/trace_timeline.py $T > gpurun_out/step_timeline.txt 2>&1
gzip -9 -c $T > gpurun_out/c3_trace.csv.gz; rm -f $T
head -40 gpurun_out/step_timeline.txt; head -24 gpurun_out/step_kernels.txt
