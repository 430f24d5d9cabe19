#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/ab.sh "DYK_FWD_SLOTS_CAP=256" "DYK_FWD_SLOTS_CAP=32" "DYK_FWD_SLOTS_CAP=64" > gpurun_out/ab_slots3.log 2>&1; cat gpurun_out/ab_slots3.log
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "DYK_FWD_SLOTS_CAP=256" "DYK_FWD_SLOTS_CAP=32" > gpurun_out/ab_slots2.log 2>&1; cat gpurun_out/ab_slots2.log
