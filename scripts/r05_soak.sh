#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1100 python3 tests/manual/soak.py 2000 5077 > gpurun_out/r05_soak_2000_graphs.log 2>&1; echo "soak rc=$?"; tail -4 gpurun_out/r05_soak_2000_graphs.log
