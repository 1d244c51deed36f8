#!/usr/bin/env python3
"""Dump the simple CSR of a bench.py config to a binary file for scripts/sim/*.c (CPU-side algorithm studies).
usage: dump_graph.py <config|nv,ncl,alpha,seed> <out.bin>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import komb_amd
from oracle import oracle as O
import bench
spec = sys.argv[1]
if spec in bench.CONFIGS:
    nv, ncl, alpha, seed = bench.CONFIGS[spec][:4]
else:
    a = spec.split(","); nv, ncl, alpha, seed = int(a[0]), int(a[1]), float(a[2]), int(a[3])
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
rowptr, col = O.simplify(nv, uv)
with open(sys.argv[2], "wb") as f:
    np.array([nv, len(col)], dtype=np.int64).tofile(f)
    rowptr.astype(np.int64).tofile(f)
    col.astype(np.int32).tofile(f)
print("nv", nv, "ne", len(col) // 2)
