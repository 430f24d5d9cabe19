#!/bin/bash
# In-call A/B of bench.py variants on ONE GPU box (boxes differ by a few tenths of a ms: never compare across calls).
#   gpurun -- 'bash tools/ab.sh "A=1" "DYK_SCHED=lanes" "DYK_STREAMS=6" "DYK_LIB=double-yolo-kaist_amd/csrc/libdyk_var_x.so"'
# Each argument is an environment assignment list for one variant; every variant runs twice, interleaved, each run under
# its own timeout (a runtime flag that hangs the process must not eat the call: ROC_SYSTEM_SCOPE_SIGNAL=0 did, round 3).
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { timeout ${AB_TIMEOUT:-240} env $1 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline $AB_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), 'ms', round(d['value'],1), 'pairs/s')"; }
[ $# -eq 0 ] && set -- "A=1"
for rep in 1 2; do for v in "$@"; do run "$v"; done; done
