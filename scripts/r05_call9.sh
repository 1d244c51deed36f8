#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_all.txt 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu_all.txt
