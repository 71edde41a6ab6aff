#!/bin/bash
# usage: tools/gpu_retry.sh <timeout_s> '<command>'   -- retries while no GPU slot is free (gpurun exit code 3)
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
