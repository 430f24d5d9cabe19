#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/ab.sh "A=1" "DYK_TMP_PS_NS=2" "DYK_TMP_PS_NS=3" "DYK_TMP_PS_NS=4" 2>&1 | tee gpurun_out/r6_ab_ps_ring_in_step.log
