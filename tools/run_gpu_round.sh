#!/bin/bash
# One GPU-box session: full GPU test suite, bench, rocprofv3 kernel-trace summary.  Outputs under gpurun_out/.
set -x
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 3000 gpurun_out/bench.json
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof/bench_under_prof.json 2> gpurun_out/prof/err.log
ls -R gpurun_out/prof | head -30
