#!/bin/bash
# one rank through the data-parallel exchange path (five segments, RCCL world 1): grouped weight gradients on / off, bucket counts
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { env $1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$2 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 DYK_FORCE_DDP=1 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ddp1 $1', round(d['ms_per_step'],2), 'ms', d.get('rccl_ranks'))"; }
p=29700
for rep in 1 2; do
  for v in "DYK_WGRAD_GROUP=16" "DYK_WGRAD_GROUP=0" "DYK_WGRAD_GROUP=4" "DYK_WGRAD_GROUP=16 DYK_DDP_BUCKETS=2" "DYK_WGRAD_GROUP=0 DYK_DDP_BUCKETS=2"; do p=$((p+1)); run "$v" $p; done
done 2>&1 | tee gpurun_out/r6_ab_ddp1_group.log
bash tools/ab.sh "A=1" 2>&1 | tee -a gpurun_out/r6_ab_ddp1_group.log
