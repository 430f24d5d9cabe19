#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 python -m pytest tests/test_gpu_stem.py -m gpu -q 2>&1 | tail -3
for v in 1 2; do
  echo "== DYK_STEM_FWD_U8=$v"
  rm -rf gpurun_out/stemt; mkdir -p gpurun_out/stemt
  DYK_STEM_FWD_U8=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stemt -o t -- python tools/stem_probe.py > gpurun_out/stemt/probe.log 2>&1
  python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/stemt/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "stem_fwd" in r["Name"]: print("%-60s calls %s avg %.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
rm -rf gpurun_out/stemt
bash tools/ab.sh "DYK_STEM_FWD_U8=1" "DYK_STEM_FWD_U8=2"
