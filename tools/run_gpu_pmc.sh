#!/bin/bash
# HBM-traffic and SQ counter passes only (the PMC part of tools/run_gpu_round.sh)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_write.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_summary.json > gpurun_out/pmc_summary.txt 2>&1
bash tools/run_pmc_sq.sh > gpurun_out/pmc_sq_run.log 2>&1
head -12 gpurun_out/pmc_summary.txt
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*counter_collection.csv" -size +20M -delete
