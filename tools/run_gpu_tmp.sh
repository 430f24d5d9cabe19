#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["ms_per_step"], d["value"], r["frac"], r["traffic"], (r["in_step"] or {}).get("tflops"), d["cpu_baseline"]["value"], d["cpu_baseline"]["c1"]["value"])
P
