#!/bin/bash
# final verification of the round: the whole GPU suite, the default bench line, the N = 2 one-device rehearsal
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_all.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu_all.txt; [ $rc = 0 ] || exit 1
timeout -k 10 600 python bench.py --no-build > gpurun_out/r05_bench_default_line.json 2> gpurun_out/r05_bench_default.err || { tail -5 gpurun_out/r05_bench_default.err; exit 1; }
cut -c1-260 gpurun_out/r05_bench_default_line.json
KOMB_BENCH_ONE_DEVICE=1 timeout -k 10 500 python bench.py --gpus 2 --steps 2 --warmup 1 --no-build --no-cpu-baseline > gpurun_out/r05_bench_gpus2_one_device_rehearsal.json 2> gpurun_out/bench_gpus2.err || { tail -5 gpurun_out/bench_gpus2.err; exit 1; }
python3 - <<'P'
import json
d=json.load(open("gpurun_out/r05_bench_gpus2_one_device_rehearsal.json"))
print(d["ms_per_step"], d["ms_per_step_resident"], d["alternatives"], d["phases_ms"]["ms_exchange"], d["phases_ms"]["ms_allreduce"])
P
