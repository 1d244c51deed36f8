#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 200 python3 scripts/prep_probe.py c3 3 > gpurun_out/prep_probe_c3.txt 2>&1 || { tail -5 gpurun_out/prep_probe_c3.txt; exit 1; }
cat gpurun_out/prep_probe_c3.txt | cut -c1-220
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_all.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu_all.txt
