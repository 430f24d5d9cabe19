#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/ab_tree.sh 2>&1 | tee gpurun_out/r6_ab_tree_group_shape.log
python -m pytest tests -m gpu -q -x 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | tail -25 > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
