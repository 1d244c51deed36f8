#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "retire" > gpurun_out/pytest_retire.txt 2>&1; echo "pytest retire rc=$?"; tail -3 gpurun_out/pytest_retire.txt
timeout -k 10 300 bash tests/manual/retire_rule_negative_control.sh 2>&1 | tail -4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prep_kt -- python3 scripts/prep_probe.py c3 3 > gpurun_out/prep_kt.log 2>&1 || exit 1
f=$(ls gpurun_out/prep_kt/*/*kernel_stats.csv | head -1); cp $f gpurun_out/prep_kernel_stats.csv; rm -rf gpurun_out/prep_kt
( time timeout -k 10 600 python bench.py --no-build > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | tail -3; echo "bench rc=$?"; tail -3 gpurun_out/bench_default.err; cut -c1-600 gpurun_out/bench_default.json
