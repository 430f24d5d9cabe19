cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof5 && mkdir -p gpurun_out/prof5
C5="--cfg kaist_dyolov4_mobilenetv3_fshare_global_cse3 --batch 32"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5 -o c5 -- python bench.py $C5 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof5/bench.json 2> gpurun_out/prof5/err.log
T=$(ls gpurun_out/prof5/c5_kernel_trace.csv gpurun_out/prof5/*/c5_kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_timeline.py $T > gpurun_out/c5_timeline.txt 2>&1
python tools/step_kernel_summary.py $T gpurun_out/c5_step_kernels.json > gpurun_out/c5_step_kernels.txt 2>&1

gzip -9 -c $T > gpurun_out/c5_trace.csv.gz; rm -f $T
