#!/bin/bash
# One gpurun call = a list of steps; a step that times out or is killed ends the call (no further GPU step after a hang),
# an ordinary failure (assertion, non-zero exit) does not.  usage: gpu_run.sh "<timeout> <name> <command...>" ...
mkdir -p gpurun_out
for spec in "$@"; do
    set -- $spec
    t=$1; name=$2; shift 2
    echo "=== $name: $* (limit ${t}s)"
    timeout -k 10 "$t" bash -c "$*" > "gpurun_out/$name.log" 2> "gpurun_out/$name.err"
    rc=$?
    echo "=== $name rc=$rc"; tail -n 6 "gpurun_out/$name.log"; tail -n 4 "gpurun_out/$name.err"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name timed out: stopping"; exit $rc; fi
done
exit 0
