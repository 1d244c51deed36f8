#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cmd="python3 bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-build --no-extras"
timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_kt -- $cmd > gpurun_out/tl_kt.log 2>&1 || { tail -5 gpurun_out/tl_kt.log; exit 1; }
python3 scripts/timeline.py gpurun_out/tl_kt 3 > gpurun_out/timeline_c3.txt 2>&1
rm -rf gpurun_out/tl_kt
head -150 gpurun_out/timeline_c3.txt
