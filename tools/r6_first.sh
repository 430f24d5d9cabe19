#!/bin/bash
# round 6, first GPU call: the re-pinned trajectory / AP tests (values printed), then the whole suite, then the bench line
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_model.py tests/test_eval_ap.py -m gpu -q -x -s -k "three_adam or three_sgd or 64_pair" 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | grep -E "passed|failed|Error|assert|three|lr |64-pair|oracle:|deviation" | cut -c1-900 > gpurun_out/r6_pinned_tests.log
python -m pytest tests -m gpu -q 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | tail -30 > gpurun_out/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -5 gpurun_out/r6_pinned_tests.log; tail -5 gpurun_out/pytest_gpu.log; tail -c 1500 gpurun_out/bench.json
