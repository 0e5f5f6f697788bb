#!/bin/bash
# dev: gpurun with retries while no box / slot is free (exit code 3). usage: tools/grun.sh TIMEOUT 'command'
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
