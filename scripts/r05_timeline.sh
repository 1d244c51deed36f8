#!/bin/bash
# timelines (scripts/timeline.py) of one cold k-truss step and one k-core pass at C2 and C3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for cfg in c2 c3; do
  cmd="python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-build --no-extras"
  timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_kt -- $cmd > gpurun_out/tl_kt.log 2>&1 || { tail -5 gpurun_out/tl_kt.log; exit 1; }
  python3 scripts/timeline.py gpurun_out/tl_kt 3 truss > gpurun_out/timeline_${cfg}_truss.txt 2>&1
  python3 scripts/timeline.py gpurun_out/tl_kt 3 core > gpurun_out/timeline_${cfg}_core.txt 2>&1
  rm -rf gpurun_out/tl_kt
  head -3 gpurun_out/timeline_${cfg}_truss.txt; head -3 gpurun_out/timeline_${cfg}_core.txt
done
