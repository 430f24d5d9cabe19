#!/bin/bash
# forward statistics replicas: conv workgroups per replica (fold cost of the normalise pass is L2-bandwidth bound, ~0.1 us per replica)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
AB_ARGS="--steps 30 --warmup 5" AB_TIMEOUT=300 bash tools/ab.sh "A=1" "DYK_FWD_SLOT_WG=64" "DYK_FWD_SLOT_WG=128" "DYK_FWD_SLOT_WG=256" "DYK_FWD_SLOT_WG=1024" "DYK_FWD_SLOT_WG=128 DYK_STAT_SLOTS=8" > gpurun_out/r3k_ab.log 2>&1
cat gpurun_out/r3k_ab.log
