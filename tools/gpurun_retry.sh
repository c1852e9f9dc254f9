#!/bin/bash
# Build-container helper: submit a gpurun call, retrying while the pod's GPU slots are busy (exit code 3 = nothing charged).
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'
T=$1; shift
git -C /root/repo rev-parse --short HEAD > /root/repo/.tree_id 2>/dev/null
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout $T -- "$@"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 90
done
exit 3
