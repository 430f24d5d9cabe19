run() { env "$@" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],2), round(d['value'],1))"; }
D=double-yolo-kaist_amd/csrc
python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -2
for i in 1 2; do
run DYK_LIB=$D/libdyk_var_head.so
run DYK_LIB=$D/libdyk_var_inloop.so
done
