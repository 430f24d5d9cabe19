#!/bin/bash
# round 5: tiled persistent depthwise weight gradient -- parity, then the MobileNetV3 cfg's shapes with the row kernel / the tiled one
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "depthwise" 2>&1 | tail -3
for v in "DYK_DW_WGRAD_TILE=0" "DYK_DW_WGRAD_TILE=512"; do
  echo "== $v"
  env $v timeout 300 python tools/dw_probe.py 2>&1 | grep -E "s1:" | sed -E 's/fwd .* dgrad [0-9.]+ us//'
done 2>&1 | tee gpurun_out/r5_dww.log
AB_ARGS="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32" bash tools/ab.sh "DYK_DW_WGRAD_TILE=0" "DYK_DW_WGRAD_TILE=512" 2>&1 | tee -a gpurun_out/r5_dww.log
