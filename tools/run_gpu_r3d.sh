#!/bin/bash
# round-3 call D: LDS-DMA BatchNorm-backward epilogue, BN prefetch, 32-channel halo tiles: kernel tests, A/B against the
# round-3 baseline library (libdyk_var_base.so = HEAD before these kernels), per-command table
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_elementwise.py tests/test_gpu_layers.py -m gpu -x -q 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids|Model Summary" | cut -c1-300 | tail -25 > gpurun_out/r3d_pytest1.log
tail -6 gpurun_out/r3d_pytest1.log
python tools/debug_late_fuse.py 2>&1 | tail -3 > gpurun_out/r3d_latefuse.log; cat gpurun_out/r3d_latefuse.log
timeout 1200 python -m pytest tests/test_gpu_bwd_bf16.py "tests/test_gpu_model.py::test_three_sgd_steps_match_reference" tests/test_gpu_harness.py -m gpu -q -s 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids|Model Summary" | cut -c1-600 | tail -40 > gpurun_out/r3d_pytest2.log
tail -25 gpurun_out/r3d_pytest2.log
B=double-yolo-kaist_amd/csrc/libdyk_var_base.so
bash tools/ab.sh "DYK_LIB=$B" "A=1" "DYK_EPI_OLD=1" > gpurun_out/r3d_ab_c3.log 2>&1; cat gpurun_out/r3d_ab_c3.log
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "DYK_LIB=$B" "A=1" > gpurun_out/r3d_ab_c5.log 2>&1; cat gpurun_out/r3d_ab_c5.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-cmds gpurun_out/cmds_c3.json > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err
python tools/cmd_roofline.py gpurun_out/cmds_c3.json > gpurun_out/r3d_cmd_roofline_c3.txt 2>&1
head -60 gpurun_out/r3d_cmd_roofline_c3.txt
