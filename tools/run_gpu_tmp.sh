#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_conv.py -q -k "depthwise" 2>&1 | tail -5 > gpurun_out/pytest_sel.log
cat gpurun_out/pytest_sel.log
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "A=1" "DYK_DW_WGRAD_TILE=0" > gpurun_out/ab_dwwg.log 2>&1
cat gpurun_out/ab_dwwg.log
python bench.py --cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32 --steps 5 --warmup 3 --no-cpu-baseline --dump-cmds gpurun_out/cmds_c5.json > gpurun_out/b_c5.json 2> gpurun_out/b_c5.err
python tools/cmd_roofline.py gpurun_out/cmds_c5.json > gpurun_out/cmds_c5.txt
grep -E "isolated|DW_" gpurun_out/cmds_c5.txt | head -12
python -m pytest tests/test_gpu_model.py -q -k "mobilenet" 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids" | tail -5
