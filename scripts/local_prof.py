#!/usr/bin/env python3
"""Profiling target: the local finish of the k-truss (and k-core) peel of one config, a few repetitions.
usage: local_prof.py <config> [reps]   (run under rocprofv3; nothing is spawned)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, komb_amd
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nv, ncl, alpha, seed = bench.CONFIGS[cfg][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
acc = komb_amd.KombAccel(); acc.from_edges(nv, uv); del uv
for _ in range(reps):
    acc.truss_run()
    acc.core_run()
st = acc.stats()
print({k: st[k] for k in ("ms_peel", "ms_truss_local", "truss_local_units", "truss_local_sweeps", "ms_core", "ms_core_local", "core_local_units", "core_local_sweeps")})
