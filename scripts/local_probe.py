#!/usr/bin/env python3
"""Per-sweep times of the local finish at a few hand-over points (KOMB_LOCAL_DEBUG=2).
usage: local_probe.py <config> | <nv> <n_cliques> <alpha>   [PROBE_WHAT=core|truss|both] [PROBE_DIVS=16,64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, komb_amd
if len(sys.argv) > 3:
    cfg = "custom"; nv, ncl, alpha, seed = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), 42
else:
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    nv, ncl, alpha, seed = bench.CONFIGS[cfg][:4]
what_all = {"core": ("core",), "truss": ("truss",), "both": ("core", "truss")}[os.environ.get("PROBE_WHAT", "both")]
divs = [int(x) for x in os.environ.get("PROBE_DIVS", "16,64").split(",")]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
acc = komb_amd.KombAccel(); acc.from_edges(nv, uv); del uv
os.environ["KOMB_FINISH"] = "local"
for div in divs:
    for what in what_all:
        units = nv if what == "core" else acc.ne
        os.environ["KOMB_LOCAL_LIMIT"] = str(units // div)
        os.environ["KOMB_LOCAL_DEBUG"] = "0"
        (acc.core_run if what == "core" else acc.truss_run)()
        os.environ["KOMB_LOCAL_DEBUG"] = "2"
        print(f"--- {cfg} {what} limit=units/{div}", file=sys.stderr, flush=True)
        (acc.core_run if what == "core" else acc.truss_run)()
