#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
T="tests/test_gpu_model.py::test_baseline_size_train_step_is_bit_reproducible"
for v in 1 2 3 4; do
  python -m pytest "$T" -q 2>&1 | grep -E "passed|failed|^FAILED" | tail -3
done > gpurun_out/bisect.log 2>&1
python -m pytest tests/test_gpu_ddp.py tests/test_gpu_elementwise.py -q 2>&1 | grep -E "passed|failed|^FAILED" | tail -5 >> gpurun_out/bisect.log
cat gpurun_out/bisect.log
