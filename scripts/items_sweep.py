#!/usr/bin/env python3
"""How the item limit of the local finish (KOMB_LOCAL_ITEMS) moves the k-core and k-truss times of a shape.
usage: items_sweep.py <config> | <nv> <n_cliques> <alpha>   (GPU box; one line per limit)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench, komb_amd
if len(sys.argv) > 3:
    name = " ".join(sys.argv[1:4]); nv, ncl, alpha, seed = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), 5
else:
    name = sys.argv[1]; nv, ncl, alpha, seed = bench.CONFIGS[name][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
acc = komb_amd.KombAccel(); acc.from_edges(nv, uv); del uv
ref = None
for lim in (0, 4 << 20, 16 << 20, 32 << 20, 64 << 20, 128 << 20, 512 << 20, 4 << 30):
    os.environ["KOMB_LOCAL_ITEMS"] = str(lim)
    best_c = best_t = None
    for _ in range(2):
        acc.core_run(); sc = acc.stats()
        acc.truss_run(); st = acc.stats()
        if best_c is None or sc["ms_core"] < best_c["ms_core"]: best_c = sc
        if best_t is None or st["ms_peel"] < best_t["ms_peel"]: best_t = st
    tr = acc.truss_fetch()[2]; core = acc.core_fetch()[1]
    if ref is None: ref = (tr, core)
    same = bool(np.array_equal(tr, ref[0]) and np.array_equal(core, ref[1]))
    print(f"{name}: items <= {lim >> 20:5d}M  core {best_c['ms_core']:8.2f} ms (local {best_c['ms_core_local']:7.2f}, {best_c['core_local_units']:8d} units {best_c['core_local_items']:11d} items {best_c['core_local_sweeps']:3d} sweeps)"
          f"  truss peel {best_t['ms_peel']:8.2f} ms (local {best_t['ms_truss_local']:7.2f}, {best_t['truss_local_units']:8d} units {best_t['truss_local_items']:11d} items {best_t['truss_local_sweeps']:3d} sweeps)  same={same}", flush=True)
