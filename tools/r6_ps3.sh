#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q -x 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | tail -15 > gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
bash tools/ab.sh "DYK_WGRAD_FOLD_W=2" "DYK_WGRAD_FOLD_W=3" "DYK_WGRAD_FOLD_W=4" "DYK_WGRAD_FOLD_W=1" 2>&1 | tee gpurun_out/r6_ab_fold_w.log
