#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python3 scripts/prep_probe.py c2 2 > gpurun_out/prep_probe_c2.txt 2>&1 || { tail -5 gpurun_out/prep_probe_c2.txt; exit 1; }
cat gpurun_out/prep_probe_c2.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" > gpurun_out/pytest_parity.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_parity.txt
timeout -k 10 300 python3 scripts/prep_probe.py c3 3 > gpurun_out/prep_probe_c3.txt 2>&1 || { tail -5 gpurun_out/prep_probe_c3.txt; exit 1; }
cat gpurun_out/prep_probe_c3.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prep_kt -- python3 scripts/prep_probe.py c3 3 > gpurun_out/prep_kt.log 2>&1 || exit 1
f=$(ls gpurun_out/prep_kt/*/*kernel_stats.csv | head -1); cp $f gpurun_out/prep_kernel_stats.csv; rm -rf gpurun_out/prep_kt
grep -E "k_prep|k_task|k_induce|radix|scan" gpurun_out/prep_kernel_stats.csv | cut -c1-80,200-400 | head -5
