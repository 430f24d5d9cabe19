#!/bin/bash
# twin pairing restricted to the forward pass (one chain of two-problem launches instead of two half-speed streams)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
AB_ARGS="--steps 30 --warmup 5" AB_TIMEOUT=300 bash tools/ab.sh "A=1" "DYK_PAIR=1 DYK_PAIR_OPS=all DYK_PAIR_WHICH=fwd" "DYK_PAIR=1 DYK_PAIR_OPS=ew DYK_PAIR_WHICH=fwd" "DYK_PAIR=1 DYK_PAIR_OPS=all DYK_PAIR_WHICH=both" "DYK_FWD_SLOT_WG=32" > gpurun_out/r3l_ab.log 2>&1
cat gpurun_out/r3l_ab.log
