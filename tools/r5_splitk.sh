#!/bin/bash
# round 5: split-K across workgroups -- kernel tests, then what the tuner picks and what the step does
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "split_k or every_tile or large_tile or baseline_size" 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | tail -25 > gpurun_out/r5_splitk_pytest.log
cat gpurun_out/r5_splitk_pytest.log
DYK_TUNE_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^tune \('c'|ms_per_step" > gpurun_out/r5_splitk_tune.log
grep -c "^tune" gpurun_out/r5_splitk_tune.log; tail -1 gpurun_out/r5_splitk_tune.log | cut -c1-300
bash tools/ab.sh "DYK_CONV_SPLITK=1" "DYK_CONV_SPLITK=0" 2>&1 | tee gpurun_out/r5_ab_splitk.log
