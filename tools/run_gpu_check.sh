#!/bin/bash
# Quick GPU-box session: GPU test suite, smoke, the bench line, and the same step through the RCCL exchange path on one rank.
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 1500 gpurun_out/bench.json; tail -3 gpurun_out/bench.err
DYK_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/bench_ddp1.json 2> gpurun_out/bench_ddp1.err; tail -c 600 gpurun_out/bench_ddp1.json; tail -3 gpurun_out/bench_ddp1.err
python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/bench_gpus2.log 2>&1; tail -2 gpurun_out/bench_gpus2.log
