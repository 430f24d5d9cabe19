for cfg in "0 0" "64 3" "64 2" "128 2"; do set -- $cfg; echo "== BKB=$1 PIPE=$2"; DYK_CONV_FORCE_BKB=$1 DYK_CONV_FORCE_PIPE=$2 python tools/gpu_probe.py convbench 2>&1 | grep bfloat | python -c "
import sys,ast
for l in sys.stdin:
    d=ast.literal_eval(l); print(d['cin'],d['cout'],d['H'],d['k'],d['s'],round(d['ms'],3),round(d['tflops'],1), round(d['gbps']))
"; done
