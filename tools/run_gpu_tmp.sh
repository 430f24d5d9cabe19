#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_model.py tests/test_gpu_ddp.py tests/test_gpu_conv.py -q 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids" | tail -6 > gpurun_out/pytest_sel.log
cat gpurun_out/pytest_sel.log
python bench.py --cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | head -c 220; echo
