#!/bin/bash
# Runs a list of GPU steps on the box, each under its own timeout; a step that is killed by its
# timeout ends the batch (no further GPU step is started), a step that merely fails does not.
# Usage: tools/gpu_batch.sh <tag> "<secs> <name> <command...>" ...
tag=$1; shift
mkdir -p gpurun_out/$tag
for spec in "$@"; do
  secs=${spec%% *}; rest=${spec#* }; name=${rest%% *}; cmd=${rest#* }
  echo "== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > gpurun_out/$tag/$name.out 2> gpurun_out/$tag/$name.err
  rc=$?
  echo "== $name rc=$rc"; tail -n 4 gpurun_out/$tag/$name.out | cut -c1-400
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name was killed by its limit: stopping the batch"; exit $rc; fi
done
exit 0
