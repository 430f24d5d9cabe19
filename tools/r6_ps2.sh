#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "pixel_streaming or test_conv_wgrad or plane_mode" 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | tail -30 > gpurun_out/r6_ps_tests.log
tail -3 gpurun_out/r6_ps_tests.log
bash tools/ab.sh "DYK_WGRAD_PS=0" "DYK_WGRAD_PS=1" 2>&1 | tee gpurun_out/r6_ab_wgrad_ps_c3.log
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "DYK_WGRAD_PS=0" "DYK_WGRAD_PS=1" 2>&1 | tee gpurun_out/r6_ab_wgrad_ps_c5.log
