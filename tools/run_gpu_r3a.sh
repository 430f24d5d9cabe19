#!/bin/bash
# round-3 call A: two-problem launches -- bitwise tests, then in-call A/B (DYK_PAIR=0 | 1) for C3, batch 1 and C5
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_conv.py tests/test_gpu_elementwise.py tests/test_gpu_layers.py -m gpu -x -q 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | tail -25 > gpurun_out/r3a_pytest.log
tail -5 gpurun_out/r3a_pytest.log
bash tools/ab.sh "DYK_PAIR=0" "DYK_PAIR=1" > gpurun_out/r3a_ab_c3.log 2>&1; cat gpurun_out/r3a_ab_c3.log
AB_ARGS="--batch 1 --steps 30" bash tools/ab.sh "DYK_PAIR=0" "DYK_PAIR=1" > gpurun_out/r3a_ab_b1.log 2>&1; cat gpurun_out/r3a_ab_b1.log
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "DYK_PAIR=0" "DYK_PAIR=1" > gpurun_out/r3a_ab_c5.log 2>&1; cat gpurun_out/r3a_ab_c5.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-cmds gpurun_out/cmds_c3.json > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
python tools/cmd_roofline.py gpurun_out/cmds_c3.json > gpurun_out/r3a_cmd_roofline_c3.txt 2>&1
head -40 gpurun_out/r3a_cmd_roofline_c3.txt
