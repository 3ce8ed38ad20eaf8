#!/bin/bash
# Dev aid: gpurun with retries while no GPU slot is free (exit code 3: nothing charged).  Usage: tools/gpu_retry.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
