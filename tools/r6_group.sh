#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "grouped or pixel_streaming or row_block" 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | tail -25 > gpurun_out/r6_group_tests.log
tail -3 gpurun_out/r6_group_tests.log
bash tools/ab.sh "DYK_WGRAD_GROUP=0" "DYK_WGRAD_GROUP=8" "DYK_WGRAD_GROUP=4" "DYK_WGRAD_GROUP=16" 2>&1 | tee gpurun_out/r6_ab_wgrad_group_c3.log
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "DYK_WGRAD_GROUP=0" "DYK_WGRAD_GROUP=8" 2>&1 | tee gpurun_out/r6_ab_wgrad_group_c5.log
AB_ARGS="--batch 1 --steps 30" bash tools/ab.sh "DYK_WGRAD_GROUP=0" "DYK_WGRAD_GROUP=8" 2>&1 | tee gpurun_out/r6_ab_wgrad_group_b1.log
python -m pytest tests -m gpu -q -x 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | tail -25 > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
