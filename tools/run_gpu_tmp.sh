#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/ab.sh "A=1" "DYK_LATE_FUSE=0" > gpurun_out/ab_late.log 2>&1
cat gpurun_out/ab_late.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_ddp.py -q 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids" | tail -12 > gpurun_out/pytest_sel.log
cat gpurun_out/pytest_sel.log
