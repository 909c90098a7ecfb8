#!/bin/bash
# submit.sh <name> <timeout_s> <command...>: gpurun with retries while the pool is busy (exit code 3: nothing charged); log in gpurun_out/<name>.out
name=$1; to=$2; shift 2
mkdir -p gpurun_out
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > gpurun_out/$name.out 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "submit rc=$rc" >> gpurun_out/$name.out; exit $rc; fi
  sleep 45
done
echo "submit: gave up" >> gpurun_out/$name.out
