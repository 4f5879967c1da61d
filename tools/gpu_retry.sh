#!/bin/bash
# usage: tools/gpu_retry.sh <logfile> <timeout> <command...>   -- retries while the pod's GPU slots are busy (exit code 3)
log=$1; shift; to=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > "$log" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
