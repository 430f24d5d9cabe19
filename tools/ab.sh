run() { env "$@" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],2), round(d['value'],1))"; }
python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -3
run A=1
run DYK_WGRAD_CANDS=2,3,0x202,0x1000002,0x1000003,0x1000202
run A=1
run DYK_WGRAD_CANDS=2,3,0x202,0x1000002,0x1000003,0x1000202
