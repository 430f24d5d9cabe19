#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bwd_bf16.py -m gpu -x -q -s 2>&1 | grep -E "passed|failed|Error|bf16 backward" | cut -c1-600 | tail -6
P=double-yolo-kaist_amd/csrc/libdyk_var_prev.so
for rep in 1 2; do
DYK_LIB=$P python bench.py --steps 8 --warmup 3 --no-cpu-baseline --dump-cmds gpurun_out/cmds_prev$rep.json 2>/dev/null | tail -1 | cut -c1-120
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --dump-cmds gpurun_out/cmds_new$rep.json 2>/dev/null | tail -1 | cut -c1-120
done
python tools/cmd_compare.py gpurun_out/cmds_prev1.json gpurun_out/cmds_new1.json > gpurun_out/r3h_cmp.txt; cat gpurun_out/r3h_cmp.txt
python tools/cmd_compare.py gpurun_out/cmds_prev2.json gpurun_out/cmds_new2.json | head -12
bash tools/ab.sh "DYK_LIB=$P" "A=1" 2>&1 | tee gpurun_out/r3h_ab_c3.log
