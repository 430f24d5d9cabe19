cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
python -m pytest tests/test_eval_ap.py tests/test_gpu_model.py -m gpu -q -s -k "eval_chain or layer_by_layer or three_adam" 2>&1 | grep -E "AP 0|bf16|three Adam|passed|failed|FAILED|Error|section" | cut -c1-900 > gpurun_out/pytest_sel.log
cat gpurun_out/pytest_sel.log
