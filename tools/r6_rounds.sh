#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/ab.sh "A=1" "DYK_TMP_ROUNDS=2" "DYK_TMP_ROUNDS=4" "DYK_TMP_ROUNDS=8" 2>&1 | tee gpurun_out/r6_ab_filler_rounds.log
