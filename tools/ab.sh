run() { env "$@" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],2), round(d['value'],1))"; }
python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_layers.py -x -q 2>&1 | tail -3
run A=1
run A=2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-layers gpurun_out/layers.json > /dev/null 2>&1; python tools/summarize_layers.py gpurun_out/layers.json | grep wgrad | head -12
