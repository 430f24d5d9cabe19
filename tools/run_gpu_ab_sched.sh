cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
bash tools/ab.sh "DYK_SCHED=lanes DYK_STEM_DIRECT=0 DYK_BENCH_FLOAT_INPUT=1" "DYK_SCHED=lanes" "DYK_SCHED=dag" "DYK_SCHED=dag DYK_STREAMS=3" "DYK_SCHED=dag DYK_STREAMS=6" "DYK_SCHED=dag DYK_SCHED_FILLER=1" 2>&1 | tee gpurun_out/ab_sched.log
