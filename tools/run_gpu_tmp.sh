cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/ab.sh "DYK_EPI_OLD=1" "A=1"
python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -q 2>&1 | tail -3
