cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/sp; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp -o sp -- python tools/stem_probe.py > /dev/null 2>&1
cat $(ls gpurun_out/sp/sp_kernel_stats.csv gpurun_out/sp/*/sp_kernel_stats.csv 2>/dev/null | head -1) | cut -c1-160 | head -8
rm -rf gpurun_out/sp
