#!/bin/bash
# gpurun with retries while no box / slot is free (exit code 3: nothing charged): tools/gpurun_retry.sh <timeout s> '<command>'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
