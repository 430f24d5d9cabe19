#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/ab.sh "DYK_OPT_OVERLAP=1" "DYK_OPT_OVERLAP=0" "DYK_OPT_OVERLAP=1 DYK_STREAMS=5" "DYK_OPT_OVERLAP=1 DYK_STREAMS=3" 2>&1 | tee gpurun_out/r6_ab_opt_overlap_c3.log
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "DYK_OPT_OVERLAP=1" "DYK_OPT_OVERLAP=0" "DYK_OPT_OVERLAP=1 DYK_WGRAD_GROUP=0" "DYK_OPT_OVERLAP=0 DYK_WGRAD_GROUP=0" 2>&1 | tee gpurun_out/r6_ab_opt_overlap_c5.log
