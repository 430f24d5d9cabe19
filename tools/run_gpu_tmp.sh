#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_model.py tests/test_all_cfgs.py tests/test_gpu_layers.py -m gpu -q -k "not baseline_size" 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids" | tail -30 > gpurun_out/pytest_sel.log
cat gpurun_out/pytest_sel.log
