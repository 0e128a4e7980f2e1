#!/bin/bash
# Developer aid (this container): gpurun with retries while every GPU slot of the pod is busy (exit code 3: nothing charged).
# Usage: tools/dev_gpurun.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
