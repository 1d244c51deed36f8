#!/usr/bin/env python3
"""k-core time of a config against the hand-over threshold of the local finish (KOMB_LOCAL_LIMIT, in units).
usage: core_limits.py <config> <limit> [<limit> ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, komb_amd
cfg = sys.argv[1]
nv, ncl, alpha, seed = bench.CONFIGS[cfg][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
acc = komb_amd.KombAccel(); acc.from_edges(nv, uv); del uv
for lim in sys.argv[2:]:
    os.environ["KOMB_LOCAL_LIMIT"] = lim
    best = None
    for _ in range(4):
        acc.core_run(); st = acc.stats()
        if best is None or st["ms_core"] < best["ms_core"]: best = st
    print(f"{cfg} limit {lim}: core {best['ms_core']:.2f} ms (local {best['ms_core_local']:.2f}, {best['core_local_units']} units, {best['core_local_items']} items, {best['core_local_sweeps']} sweeps)", flush=True)
