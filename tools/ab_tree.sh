#!/bin/bash
# In-call A/B of TWO TREES on one GPU box (boxes differ by +-1.5 %: numbers of different calls do not compare): the current tree
# against a worktree of an earlier commit built beside it --
#   git worktree add -f _old <commit> && make -C _old/double-yolo-kaist_amd/csrc -j8 all
#   gpurun -- 'bash tools/ab_tree.sh'        (then: git worktree remove --force _old)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { (cd $1 && timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline $AB_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), 'ms', round(d['value'],1), 'pairs/s')"); }
for rep in 1 2 3; do run _old; run .; done
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32"
for rep in 1 2; do run _old; run .; done
AB_ARGS="--batch 1 --steps 30"
for rep in 1 2; do run _old; run .; done
