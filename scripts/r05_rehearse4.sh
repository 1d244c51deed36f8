#!/bin/bash
# rehearsal of the N = 4 default flow with all ranks on ONE GPU (gloo through host memory): the code path, not its speed
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export KOMB_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 4 --config c2 --steps 2 --warmup 1 --no-build --no-cpu-baseline > gpurun_out/bench_gpus4_c2_one_device.json 2> gpurun_out/bench_gpus4.err || { tail -8 gpurun_out/bench_gpus4.err; exit 1; }
python3 - <<'P'
import json
d=json.load(open("gpurun_out/bench_gpus4_c2_one_device.json"))
print(d["n_gpus"], d["ms_per_step"], d["ms_per_step_resident"], d["alternatives"], d["config"].get("parallelism"), d["phases_ms"]["ms_exchange"], d["phases_ms"]["ms_allreduce"], d["kcore"]["ms"])
P
