cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python tools/dw_probe.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_conv.py tests/test_gpu_layers.py -m gpu -q 2>&1 | tail -2
export AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32"
bash tools/ab.sh "A=1"
