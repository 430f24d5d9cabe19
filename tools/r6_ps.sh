#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "pixel_streaming or test_conv_wgrad or plane_mode" 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | tail -30 > gpurun_out/r6_ps_tests.log
tail -5 gpurun_out/r6_ps_tests.log
timeout 1200 python tools/wgps_probe.py > gpurun_out/r6_wgps_probe.log 2>&1
tail -120 gpurun_out/r6_wgps_probe.log
