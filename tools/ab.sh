run() { env "$@" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],2), round(d['value'],1))"; }
python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_layers.py -x -q 2>&1 | tail -3
PROBE_EARLY=1 python tools/gpu_probe.py ablate 2>&1 | grep "^(" | grep -v "t160\|bkb\|nostore" 
run A=1
run A=2
