"""First k-truss / k-core call on fresh contexts: wall time, phase times, and what the context's pool spent in hipMalloc."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, komb_amd
os.environ["KOMB_POOL_DEBUG"] = "1"
nv, ncl = 10_000_000, 24_250_000
uv = komb_amd.gen_hug_edges(nv, ncl, 2.6, 42)
for rep in range(3):
    a = komb_amd.KombAccel()
    t0 = time.perf_counter(); a.from_edges(nv, uv); tb = time.perf_counter() - t0
    t0 = time.perf_counter(); a.truss_run(); t1 = time.perf_counter() - t0
    s1 = a.stats()
    t0 = time.perf_counter(); a.truss_run(); t2 = time.perf_counter() - t0
    s2 = a.stats()
    keys = ("ms_orient", "ms_tri_fill", "ms_sort", "ms_compact", "ms_peel", "ms_gather")
    print(f"rep {rep}: build {tb*1e3:.1f} ms, first truss {t1*1e3:.1f} ms {[round(s1[k],1) for k in keys]}, second {t2*1e3:.1f} ms {[round(s2[k],1) for k in keys]}", flush=True)
    a.close()
