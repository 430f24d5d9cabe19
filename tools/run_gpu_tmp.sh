cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
DYK_GRAPH=1 DYK_GRAPH_DEBUG=1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/graph_dbg.log 2>&1
echo "exit $?"; grep "dyk graph" gpurun_out/graph_dbg.log | head -4; tail -1 gpurun_out/graph_dbg.log | cut -c1-200
bash tools/ab.sh "DYK_GRAPH=0" "DYK_GRAPH=1"
export AB_ARGS="--batch 1"
bash tools/ab.sh "DYK_GRAPH=0" "DYK_GRAPH=1"
