cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
grep -E "AP 0|bf16 per-layer|three Adam|passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | cut -c1-600 | head -30
export AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32"
bash tools/ab.sh "DYK_SCHED=lanes" "DYK_SCHED=dag" "DYK_SCHED=dag DYK_STREAMS=6" 2>&1 | tee gpurun_out/ab_c5.log
export AB_ARGS="--batch 1"
bash tools/ab.sh "DYK_SCHED=lanes" "DYK_SCHED=dag" "DYK_SCHED=dag DYK_STREAMS=6" 2>&1 | tee gpurun_out/ab_b1.log
