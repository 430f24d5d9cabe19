#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/ab.sh "A=1" "DYK_STREAMS_FWD=2" "DYK_STREAMS_FWD=3" "DYK_SCHED_POLICY=hlfet" "DYK_STREAMS_BWD=3" "DYK_STREAMS_BWD=5" 2>&1 | tee gpurun_out/r6_ab_streams_policy.log
