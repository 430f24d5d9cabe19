cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_ddp.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu" | tail -4
bash tools/ab.sh "DYK_ISSUE_THREADS=0" "DYK_ISSUE_THREADS=1"
export AB_ARGS="--batch 1"
bash tools/ab.sh "DYK_ISSUE_THREADS=0" "DYK_ISSUE_THREADS=1"
export AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32"
bash tools/ab.sh "DYK_ISSUE_THREADS=0" "DYK_ISSUE_THREADS=1"
