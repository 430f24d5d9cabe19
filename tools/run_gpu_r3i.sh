#!/bin/bash
# round 3, session 3: (1) where a launch-bound BatchNorm pass spends its time, (2) 4-stage weight-gradient ring
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
set -x
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "wgrad" 2>&1 | tail -5 > gpurun_out/r3i_pytest.log
timeout 300 python tools/gpu_probe.py bnfold > gpurun_out/r3i_bnfold.log 2>&1
DYK_TUNE_VERBOSE=1 timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r3i_tune_verbose.log 2>&1
AB_TIMEOUT=300 bash tools/ab.sh "A=1" "DYK_WGRAD_CANDS=2,3,0x202,0x1000002,0x1000003,0x1000202,0x10000002" > gpurun_out/r3i_ab.log 2>&1
cat gpurun_out/r3i_pytest.log gpurun_out/r3i_bnfold.log gpurun_out/r3i_ab.log
grep "^tune ('w'" gpurun_out/r3i_tune_verbose.log | grep ", 1, 1)" | head -30
