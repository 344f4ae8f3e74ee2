#!/bin/bash
# usage: scripts/dev/gpu_retry.sh <timeout-seconds> '<command>'   - retries while no GPU slot is free (gpurun exit code 3)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/gpu_retry.$$.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then tail -4 /tmp/gpu_retry.$$.log; exit $rc; fi
  sleep 45
done
echo "no GPU slot after 40 tries"; exit 3
