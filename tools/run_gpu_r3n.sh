#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32 --steps 20 --warmup 5" AB_TIMEOUT=300 bash tools/ab.sh "DYK_FWD_SLOT_WG=32" "A=1" "DYK_DW_SLOTS=8" "DYK_DW_SLOTS=16" "DYK_FWD_SLOT_WG=512 DYK_DW_SLOTS=8" > gpurun_out/r3n_ab_c5.log 2>&1
cat gpurun_out/r3n_ab_c5.log
