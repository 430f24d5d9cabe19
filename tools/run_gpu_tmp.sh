#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_conv.py -q -k "depthwise" 2>&1 | tail -3
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "A=1" "DYK_DW_TILE_WGS=0" "DYK_DW_TILE_WGS=3072" "DYK_DW_TILE_WGS=768" > gpurun_out/ab_dwp.log 2>&1; cat gpurun_out/ab_dwp.log
