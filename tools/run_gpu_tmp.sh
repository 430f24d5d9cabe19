#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_stem.py -q 2>&1 | tail -3 > gpurun_out/pytest_sel.log
cat gpurun_out/pytest_sel.log
python bench.py --cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --dump-cmds gpurun_out/cmds_c5.json > gpurun_out/b_c5.json 2> gpurun_out/b_c5.err
python tools/cmd_roofline.py gpurun_out/cmds_c5.json > gpurun_out/cmds_c5.txt
head -c 200 gpurun_out/b_c5.json; echo; grep -E "isolated|STEM" gpurun_out/cmds_c5.txt | head
