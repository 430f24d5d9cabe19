#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_stem.py -q 2>&1 | tail -12 > gpurun_out/pytest_sel.log; cat gpurun_out/pytest_sel.log
bash tools/ab.sh "A=1" "DYK_STEM_FWD_U8=0" > gpurun_out/ab_stemw.log 2>&1; cat gpurun_out/ab_stemw.log
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "A=1" "DYK_STEM_FWD_U8=0" > gpurun_out/ab_stemw5.log 2>&1; cat gpurun_out/ab_stemw5.log
python bench.py --steps 4 --warmup 4 --no-cpu-baseline --dump-cmds gpurun_out/cmds_c3.json 2>/dev/null >/dev/null
python tools/cmd_roofline.py gpurun_out/cmds_c3.json | grep -E "STEM" | head -4
