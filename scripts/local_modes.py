#!/usr/bin/env python3
"""Debug build only (KOMB_LOCAL_MODE): k-core / k-truss peel times of a config with the fixed point's list (1) and
notification kernel (2) switched on and off.  usage: local_modes.py <config>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, komb_amd
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
nv, ncl, alpha, seed = bench.CONFIGS[cfg][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
acc = komb_amd.KombAccel(); acc.from_edges(nv, uv); del uv
for mode in ("0", "1", "2", "3"):
    os.environ["KOMB_LOCAL_MODE"] = mode
    best = [1e9, 1e9, 1e9, 1e9]
    for _ in range(4):
        acc.core_run(); sc = acc.stats()
        acc.truss_run(); st = acc.stats()
        best = [min(best[0], sc["ms_core"]), min(best[1], sc["ms_core_local"]), min(best[2], st["ms_peel"]), min(best[3], st["ms_truss_local"])]
    print(f"{cfg} mode {mode}: core {best[0]:.2f} (local {best[1]:.2f})  truss peel {best[2]:.2f} (local {best[3]:.2f})", flush=True)
