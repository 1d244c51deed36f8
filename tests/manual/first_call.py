"""Wall time of the FIRST komb_truss_run / komb_core_run on a fresh context (pool empty: every buffer is a hipMalloc) against
the steady state.  usage: first_call.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (one HIP runtime for both)
import komb_amd, bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
nv, ncl, alpha, seed = bench.CONFIGS[cfg][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
with komb_amd.KombAccel() as a:
    a.from_edges(nv, uv)
    for name, fn in (("k-core", a.core_run), ("k-truss", a.truss_run)):
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{cfg} {name}: first call {ts[0]:.1f} ms, then {ts[1]:.1f} {ts[2]:.1f} {ts[3]:.1f} ms (wall, incl. host)")
