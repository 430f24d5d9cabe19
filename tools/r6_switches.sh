#!/bin/bash
# every documented switch once through three train steps of the target cfg (no crash, finite loss); timing is not the point
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { timeout 400 env $1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline $2 2>gpurun_out/sw_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ok  ', '$1', round(d['ms_per_step'],2), 'ms loss', round(d['final_loss'],4))
except Exception as e:
    print('FAIL', '$1', e); print(open('gpurun_out/sw_err.log').read()[-600:])"; }
for v in "A=1" "DYK_SCHED=lanes" "DYK_OVERLAP=0" "DYK_GRAPH=1" "DYK_BNBWD_FUSE=0" "DYK_DEBUG_PLAN=1" "DYK_CONV_LT=0 DYK_CONV_KG=0 DYK_CONV_PW=0 DYK_CONV_SC=0 DYK_CONV_SPLITK=0" "DYK_OPT_OVERLAP=0" "DYK_OPT_OVERLAP=early" "DYK_ISSUE_THREADS=0" "DYK_STREAMS=2" "DYK_SCHED_POLICY=typed" "DYK_WGRAD_PARTIALS=0" "DYK_AUTOTUNE=0" "DYK_KEEP_DZ=1" "DYK_WGRAD_GROUP=2"; do run "$v"; done 2>&1 | tee gpurun_out/r6_switch_smoke.log
run "DYK_DW_PRE=1" "--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" | tee -a gpurun_out/r6_switch_smoke.log
