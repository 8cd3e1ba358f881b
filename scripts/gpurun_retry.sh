#!/bin/bash
# usage (in the build container): scripts/gpurun_retry.sh <log> <timeout s> '<command>'  -- gpurun, retried every 2 minutes while no GPU slot is free
log=$1; to=$2; cmd=$3
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$cmd" > $log 2>&1
  grep -q "status=transient" $log || break
  sleep 120
done
