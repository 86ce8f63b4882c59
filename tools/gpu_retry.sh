#!/bin/bash
# gpu_retry.sh <timeout-seconds> <command...>: gpurun with retries while no box / slot is free (exit code 3: nothing charged)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
