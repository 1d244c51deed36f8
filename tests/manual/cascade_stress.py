"""One-off: long peel cascades (one unit per sub-round): path for k-core, triangle strip for k-truss."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import komb_amd
from oracle import oracle as O
n = 300_000
path = np.stack([np.arange(n - 1), np.arange(1, n)], axis=1)
strip = np.concatenate([path, np.stack([np.arange(n - 2), np.arange(2, n)], axis=1)])
# strip with a dense head so that peeling must walk the whole strip from one end at level 2
for name, uv in (("path", path), ("strip", strip)):
    uv = uv.astype(np.int64)
    rowptr, col = O.simplify(n, uv)
    with komb_amd.KombAccel() as a:
        a.from_edges(n, uv)
        t = time.time(); deg, core = a.run_core(); tc = time.time() - t
        t = time.time(); eu, ev, tr = a.run_truss(); tt = time.time() - t
        st = a.stats()
        print(f"{name}: core ok={bool(np.array_equal(core, O.coreness(rowptr, col)))} {tc*1e3:.1f} ms (levels {st['core_levels']}, rounds {st['core_subrounds']}, launches {st['core_launches']}); "
              f"truss ok={bool(np.array_equal(tr, O.trussness(rowptr, col)))} {tt*1e3:.1f} ms (rounds {st['truss_subrounds']}, launches {st['truss_launches']})", flush=True)
