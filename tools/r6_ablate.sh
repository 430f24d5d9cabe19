#!/bin/bash
# What the C3 step would cost if a kernel family were free (DYK_SKIP_OPS: timing only, the results are garbage): the upper
# bound of anything an optimisation of that family can buy in the step.  Op codes: include/dyk_hip.h.
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
AB_TIMEOUT=300 bash tools/ab.sh "DYK_SKIP_OPS=0" "DYK_SKIP_OPS=2,31" "DYK_SKIP_OPS=6" "DYK_SKIP_OPS=30" "DYK_SKIP_OPS=6,30,5" "DYK_SKIP_OPS=2,31,6,30,5" "DYK_SKIP_OPS=1" 2>&1 | tee gpurun_out/r6_ablate_c3.log
