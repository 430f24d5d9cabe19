#!/bin/bash
# round 5, first GPU call: the new parity tests + optimizer ordering A/B
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_loss_nms.py tests/test_gpu_repro.py tests/test_gpu_harness.py tests/test_abi.py -m gpu -q -x 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | tail -15 > gpurun_out/r5_first_pytest.log
cat gpurun_out/r5_first_pytest.log
bash tools/ab.sh "DYK_OPT_OVERLAP=1" "DYK_OPT_OVERLAP=early" "DYK_OPT_OVERLAP=0" 2>&1 | tee gpurun_out/r5_ab_optimizer_order.log
