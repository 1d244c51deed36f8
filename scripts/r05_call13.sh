#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python3 scripts/shard_probe.py c3 > gpurun_out/shard_probe_c3.txt 2>&1; cat gpurun_out/shard_probe_c3.txt | tail -5
KOMB_BENCH_ONE_DEVICE=1 timeout -k 10 500 python bench.py --gpus 2 --steps 2 --warmup 1 --no-build --no-cpu-baseline > gpurun_out/r05_bench_gpus2_one_device_rehearsal.json 2> gpurun_out/bench_gpus2.err || { tail -5 gpurun_out/bench_gpus2.err; exit 1; }
python3 - <<'P'
import json
d=json.load(open("gpurun_out/r05_bench_gpus2_one_device_rehearsal.json"))
print(d["ms_per_step"], d["ms_per_step_resident"], d["alternatives"], d["config"]["shard_peel"], d["phases_ms"]["ms_exchange"], d["phases_ms"]["ms_allreduce"])
P
