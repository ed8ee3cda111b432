#!/bin/bash
# developer helper (this container, not the GPU box): retry a gpurun call while the pod's GPU slots are busy (exit 3).
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
