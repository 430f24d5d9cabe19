#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bwd_bf16.py tests/test_gpu_layers.py "tests/test_gpu_model.py::test_three_sgd_steps_match_reference" -m gpu -x -q -s 2>&1 | grep -E "passed|failed|Error|bf16 backward|three SGD|parameter-delta" | cut -c1-1500 | tail -8
B=double-yolo-kaist_amd/csrc/libdyk_var_base.so
for rep in 1 2; do
DYK_LIB=$B python bench.py --steps 8 --warmup 3 --no-cpu-baseline --dump-cmds gpurun_out/cmds_base$rep.json 2>/dev/null | tail -1 | cut -c1-120
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --dump-cmds gpurun_out/cmds_new$rep.json 2>/dev/null | tail -1 | cut -c1-120
done
python tools/cmd_compare.py gpurun_out/cmds_base1.json gpurun_out/cmds_new1.json > gpurun_out/r3g_cmp_base_new.txt; cat gpurun_out/r3g_cmp_base_new.txt
python tools/cmd_compare.py gpurun_out/cmds_base2.json gpurun_out/cmds_new2.json | head -14
bash tools/ab.sh "DYK_LIB=$B" "A=1" 2>&1 | tee gpurun_out/r3g_ab_c3.log
