#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 / "transient": nothing charged)
# usage: tools/gpurun_retry.sh TIMEOUT_S 'command'
T=$1; shift
for i in $(seq 1 40); do
    out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1); rc=$?
    if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
    echo "$out" | tail -60
    exit $rc
done
echo "gpurun_retry: no slot after 40 tries"; exit 3
