#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged).  usage: gpurun_retry.sh <log> <timeout> <command>
log=$1; to=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
