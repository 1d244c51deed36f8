#!/bin/bash
# round 5: the four rocprofv3 passes of the bench command for C3 and C2 (scripts/gpu_profile.sh), then the default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash scripts/gpu_profile.sh r05_c3 c3 || exit 1
bash scripts/gpu_profile.sh r05_c2 c2 || exit 1
timeout -k 10 600 python bench.py --no-build > gpurun_out/r05_bench_default_line.json 2> gpurun_out/r05_bench_default.err || exit 1
cut -c1-300 gpurun_out/r05_bench_default_line.json
