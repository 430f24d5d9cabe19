#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/ab.sh "A=1" "DYK_TUNE_COLD=1" > gpurun_out/ab_cold.log 2>&1; cat gpurun_out/ab_cold.log
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "A=1" "DYK_TUNE_COLD=1" >> gpurun_out/ab_cold.log 2>&1; tail -4 gpurun_out/ab_cold.log
