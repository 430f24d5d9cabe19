#!/bin/bash
# round-3 call B: scheduling policy A/B (resource-typed streams, selective pairing), tests touched by the ADVICE fixes
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
bash tools/ab.sh "DYK_SCHED_POLICY=hlfet DYK_PAIR=0" "DYK_SCHED_POLICY=typed DYK_PAIR=0" "DYK_SCHED_POLICY=typed DYK_PAIR_OPS=ew" "DYK_SCHED_POLICY=hlfet DYK_PAIR_OPS=ew" "DYK_SCHED_POLICY=typed DYK_PAIR=0 DYK_STREAMS=3" > gpurun_out/r3b_ab_c3.log 2>&1; cat gpurun_out/r3b_ab_c3.log
AB_ARGS="--batch 1 --steps 30" bash tools/ab.sh "DYK_SCHED_POLICY=hlfet DYK_PAIR=0" "DYK_SCHED_POLICY=typed DYK_PAIR_OPS=ew" > gpurun_out/r3b_ab_b1.log 2>&1; cat gpurun_out/r3b_ab_b1.log
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "DYK_SCHED_POLICY=hlfet DYK_PAIR=0" "DYK_SCHED_POLICY=typed DYK_PAIR=0" "DYK_SCHED_POLICY=typed DYK_PAIR_OPS=ew" "DYK_SCHED_POLICY=hlfet DYK_PAIR_OPS=ew" > gpurun_out/r3b_ab_c5.log 2>&1; cat gpurun_out/r3b_ab_c5.log
timeout 900 python -m pytest tests/test_gpu_harness.py tests/test_boxes.py tests/test_gpu_ddp.py -m gpu -x -q 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path|amdgpu.ids" | tail -15 > gpurun_out/r3b_pytest.log
tail -5 gpurun_out/r3b_pytest.log
