#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "preparation_long_rows or cliques_and_hubs or induced" > gpurun_out/pytest_prep.txt 2>&1; rc=$?; tail -5 gpurun_out/pytest_prep.txt; [ $rc = 0 ] || exit 1
timeout -k 10 200 python3 scripts/prep_probe.py c3 3 > gpurun_out/prep_probe_c3.txt 2>&1 || { tail -5 gpurun_out/prep_probe_c3.txt; exit 1; }
cat gpurun_out/prep_probe_c3.txt | cut -c1-220
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_komb2.py -x -q -m gpu -k "not full_size_c3" > gpurun_out/pytest_parity.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_parity.txt
