#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
python tools/debug_late_fuse.py keep 2>&1 | tail -1 | cut -c1-600
B=double-yolo-kaist_amd/csrc/libdyk_var_base.so
for rep in 1 2; do
DYK_LIB=$B python bench.py --steps 6 --warmup 2 --no-cpu-baseline --dump-cmds gpurun_out/cmds_base$rep.json 2>/dev/null | tail -1 | cut -c1-120
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --dump-cmds gpurun_out/cmds_new$rep.json 2>/dev/null | tail -1 | cut -c1-120
DYK_EPI_OLD=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --dump-cmds gpurun_out/cmds_old$rep.json 2>/dev/null | tail -1 | cut -c1-120
done
python tools/cmd_compare.py gpurun_out/cmds_base1.json gpurun_out/cmds_base2.json > gpurun_out/r3e_cmp_noise.txt; head -12 gpurun_out/r3e_cmp_noise.txt
python tools/cmd_compare.py gpurun_out/cmds_base1.json gpurun_out/cmds_new1.json > gpurun_out/r3e_cmp_base_new.txt; cat gpurun_out/r3e_cmp_base_new.txt
python tools/cmd_compare.py gpurun_out/cmds_old2.json gpurun_out/cmds_new2.json > gpurun_out/r3e_cmp_old_new.txt; cat gpurun_out/r3e_cmp_old_new.txt
timeout 600 python -m pytest "tests/test_gpu_model.py::test_three_sgd_steps_match_reference" -m gpu -q -s 2>&1 | grep -E "three SGD|parameter-delta|passed|failed|Error" | cut -c1-900
