#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_komb2.py -x -q -m gpu -k "not full_size_c3" > gpurun_out/pytest_parity.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_parity.txt
timeout -k 10 600 python bench.py --no-build --no-cpu-baseline > gpurun_out/bench_nocpu.json 2> gpurun_out/bench_nocpu.err || { tail -5 gpurun_out/bench_nocpu.err; exit 1; }
python3 - <<'P'
import json
d=json.load(open("gpurun_out/bench_nocpu.json"))
print("ms", round(d["ms_per_step"],2), "res", round(d["ms_per_step_resident"],2), "faithful", d["runtruss_faithful"]["ms"], "c2 kcore", d["c2"]["kcore"]["ms"], "corea", d["corea"]["ms_call_wall"], d["corea"]["ms_device_rank_kernels"], "kcore", d["kcore"]["ms"])
print("first", d["first_call"])
print("traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"])
P
