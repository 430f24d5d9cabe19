#!/bin/bash
# round 5: persistent pointwise kernels -- what the tuner picks and what the step does (C3 and the MobileNetV3 cfg)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
DYK_TUNE_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^tune \('c', 1, 16, [0-9]+, [0-9]+, [0-9]+, [0-9]+, 1,|ms_per_step" | cut -c1-330 > gpurun_out/r5_pw_tune.log
grep -c "0x7" gpurun_out/r5_pw_tune.log
bash tools/ab.sh "DYK_CONV_PW=1" "DYK_CONV_PW=0" 2>&1 | tee gpurun_out/r5_ab_pw.log
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "DYK_CONV_PW=1" "DYK_CONV_PW=0" 2>&1 | tee gpurun_out/r5_ab_pw_c5.log
