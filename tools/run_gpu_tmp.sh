#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
p() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2))"; }
B="bench.py --gpus 1 --steps 12 --warmup 6 --no-cpu-baseline --no-roofline"
python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | p "no-ddp"
DYK_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $B 2>/dev/null | tail -1 | p "ddp-geometric"
DYK_FORCE_DDP=1 DYK_DDP_BUCKETS=8 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $B 2>/dev/null | tail -1 | p "ddp-8buckets"
python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | p "no-ddp"
python -m pytest tests/test_gpu_ddp.py -q 2>&1 | grep -E "passed|failed" | tail -2
