#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 scripts/tail_sweep.py 10000000 27500000 2.2 > gpurun_out/tail_sweep_a22.txt 2>&1 || { tail -5 gpurun_out/tail_sweep_a22.txt; exit 1; }
cat gpurun_out/tail_sweep_a22.txt | cut -c1-250
