#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dump-cmds gpurun_out/cmds_c3.json > gpurun_out/r3m_bench.json 2> gpurun_out/r3m_bench.err
DYK_ROOFLINE_TOP=1000 python tools/cmd_roofline.py gpurun_out/cmds_c3.json > gpurun_out/r3m_cmd_roofline_c3_full.txt 2>&1
rm -f gpurun_out/cmds_c3.json
tail -c 1500 gpurun_out/r3m_bench.json
