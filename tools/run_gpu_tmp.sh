cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
python -m pytest tests/test_gpu_stem.py tests/test_eval_ap.py tests/test_gpu_ddp.py -m gpu -q 2>&1 | tail -150 > gpurun_out/pytest_sel.log
grep -E "AP 0|passed|failed|FAILED|Error" gpurun_out/pytest_sel.log | head -30
bash tools/ab.sh "DYK_SCHED=lanes DYK_STEM_DIRECT=0 DYK_BENCH_FLOAT_INPUT=1" "DYK_SCHED=dag DYK_STEM_DIRECT=0 DYK_BENCH_FLOAT_INPUT=1" "DYK_SCHED=dag" "DYK_SCHED=dag DYK_BENCH_FLOAT_INPUT=1" 2>&1 | tee gpurun_out/ab_stem.log
