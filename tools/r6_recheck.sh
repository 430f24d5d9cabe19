#!/bin/bash
# re-check of two kept-but-off experiments on the round-6 code (grouped weight gradients change what runs beside them)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/ab.sh "A=1" "DYK_BNFWD=1" "DYK_PAIR=1 DYK_PAIR_OPS=ew" 2>&1 | tee gpurun_out/r6_ab_recheck_bnfwd_pair.log
AB_ARGS="--batch 1 --steps 30" bash tools/ab.sh "A=1" "DYK_BNFWD=1" 2>&1 | tee -a gpurun_out/r6_ab_recheck_bnfwd_pair.log
