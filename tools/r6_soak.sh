#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/soak300.json
cut -c1-500 gpurun_out/soak300.json
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32 2>/dev/null | tail -1 > gpurun_out/soak100_c5.json
cut -c1-400 gpurun_out/soak100_c5.json
