# the data-parallel path (bucketed exchange forced on one rank) under the two list-scheduling policies
run() { env $1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$2 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 DYK_FORCE_DDP=1 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), 'ms', d.get('rccl_ranks'), 'loss', round(d['final_loss'],4))"; }
run "DYK_SCHED_POLICY=hlfet" 29601; run "DYK_SCHED_POLICY=event" 29602; run "DYK_SCHED_POLICY=hlfet" 29603; run "DYK_SCHED_POLICY=event" 29604
