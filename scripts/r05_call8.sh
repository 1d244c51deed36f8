#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2; do
timeout -k 10 400 python bench.py --no-build --no-cpu-baseline > gpurun_out/bench_nocpu_$i.json 2> gpurun_out/bench_nocpu_$i.err || exit 1
python3 - gpurun_out/bench_nocpu_$i.json <<'P'
import json,sys
d=json.load(open(sys.argv[1]))
print("ms", round(d["ms_per_step"],2), "res", round(d["ms_per_step_resident"],2), "faithful", d["runtruss_faithful"]["ms"], "c2 kcore", d["c2"]["kcore"]["ms"], d["c2"]["kcore"]["first_call_ms"], "corea", d["corea"]["ms_call_wall"], d["corea"]["ms_device_rank_kernels"], "kcore", d["kcore"]["ms"])
P
done
