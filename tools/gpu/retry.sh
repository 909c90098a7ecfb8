#!/bin/bash
# usage: tools/gpu/retry.sh <timeout> <script> <out>: gpurun with retries while no slot is free (exit 3 = nothing charged)
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $1 -- "bash $2" > $3 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
