#!/bin/bash
# Developer helper (this container only): run a command on the GPU box through gpurun, waiting for a free slot.
#   tools/gpu.sh <tag> <timeout_s> '<command>'      -> gpurun_out/<tag>_call.log
# Only "no slot free" (exit 3: nothing ran, nothing charged) is retried; a command that ran is never repeated.
tag=$1; to=$2; shift 2
mkdir -p gpurun_out
for attempt in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > gpurun_out/${tag}_call.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
