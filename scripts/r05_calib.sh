#!/bin/bash
# round 5: calibration + per-step traces of round 4's engines (debug build), the data the round's plan is made from
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 120 scripts/calib/bin/atomic_rate > gpurun_out/atomic_rate.txt 2>&1 || exit 1
tail -n 3 gpurun_out/atomic_rate.txt
export KOMB_ACCEL_LIB=komb_amd/libv/dbg/libkomb_accel.so
for cfg in c3 c2; do
  KOMB_FINISH=local timeout -k 10 200 python3 scripts/peel_trace.py $cfg > gpurun_out/trace_$cfg.log 2>&1 || exit 1
  mv gpurun_out/peel_trace_$cfg.txt gpurun_out/peel_trace_local_$cfg.txt
  python3 scripts/peel_trace_report.py gpurun_out/peel_trace_local_$cfg.txt > gpurun_out/peel_trace_local_${cfg}_report.txt 2>&1
done
unset KOMB_ACCEL_LIB
for cfg in c3 c2; do
  KOMB_LOCAL_DEBUG=2 timeout -k 10 200 python3 - $cfg > gpurun_out/local_sweeps_$cfg.txt 2>&1 <<'P' || exit 1
import os, sys
sys.path.insert(0, os.getcwd())
import bench, komb_amd
nv, ncl, alpha, seed = bench.CONFIGS[sys.argv[1]][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
a = komb_amd.KombAccel(); a.from_edges(nv, uv); del uv
for rep in range(2):
    print("--- truss", file=sys.stderr, flush=True); a.truss_run(); s = a.stats()
    print({k: s[k] for k in s if k.startswith("ms_") or "launch" in k or "round" in k or "level" in k or "scan" in k}, file=sys.stderr, flush=True)
    print("--- core", file=sys.stderr, flush=True); a.core_run(); s = a.stats()
    print({k: s[k] for k in s if "core" in k}, file=sys.stderr, flush=True)
P
done
echo done
