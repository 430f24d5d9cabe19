#!/bin/bash
# round 3, session 3: replica folds of the BatchNorm passes in one round trip; focal loss parity
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
set -x
timeout 900 python -m pytest tests/test_gpu_loss_nms.py tests/test_gpu_conv.py tests/test_gpu_elementwise.py -q -x 2>&1 | tail -5 > gpurun_out/r3j_pytest.log
timeout 300 python tools/gpu_probe.py bnfold > gpurun_out/r3j_bnfold.log 2>&1
B=double-yolo-kaist_amd/csrc/libdyk_base.so
AB_TIMEOUT=300 bash tools/ab.sh "DYK_LIB=$B" "A=1" "DYK_FWD_SLOT_WG=128" "DYK_FWD_SLOT_WG=128 DYK_STAT_SLOTS=8" > gpurun_out/r3j_ab.log 2>&1
cat gpurun_out/r3j_pytest.log gpurun_out/r3j_bnfold.log gpurun_out/r3j_ab.log
