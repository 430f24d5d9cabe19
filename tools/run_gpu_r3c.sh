#!/bin/bash
# round-3 call C: new parity tests (bf16 backward per section, SGD trajectory, ADVICE fixes), SQ counter passes, bench with the per-command table
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bwd_bf16.py tests/test_gpu_harness.py tests/test_boxes.py "tests/test_gpu_model.py::test_three_sgd_steps_match_reference" "tests/test_gpu_model.py::test_three_adam_steps_match_reference_losses" -m gpu -q -s 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids|Model Summary" | cut -c1-400 | tail -60 > gpurun_out/r3c_pytest.log
tail -30 gpurun_out/r3c_pytest.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-cmds gpurun_out/cmds_c3.json > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench.err
python tools/cmd_roofline.py gpurun_out/cmds_c3.json > gpurun_out/r3c_cmd_roofline_c3.txt 2>&1
head -12 gpurun_out/r3c_cmd_roofline_c3.txt; grep -E "256x320 c64>32|128x160 c128>64" gpurun_out/r3c_cmd_roofline_c3.txt
bash tools/run_pmc_sq.sh
