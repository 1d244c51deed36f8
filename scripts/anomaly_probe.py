"""Round 5 debugging aid: where the wall time of a vmask k-truss run and of repeated C2 k-core runs goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import komb_amd, bench
def wall(f, n=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
nv, ncl, alpha, seed = bench.CONFIGS["c2"][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
a = komb_amd.KombAccel(); a.from_edges(nv, uv)
print("c2 k-core first wall", wall(a.core_run), "event", a.stats()["ms_core"])
for i in range(3): print("c2 k-core x10 wall", wall(a.core_run, 10), "event", a.stats()["ms_core"], "launches", a.stats()["core_launches"], flush=True)
deg, core = a.core_fetch()
mask = (core == core.max()).astype(np.uint8)
for i in range(4):
    w = wall(lambda: a.truss_run(mask)); s = a.stats()
    print("c2 faithful wall", round(w, 2), {k: round(v, 3) for k, v in s.items() if k.startswith("ms_") and v}, flush=True)
a.set_option("POOL_DEBUG", "1")
a.truss_run(mask); a.truss_run(mask)
a.set_option("POOL_DEBUG", None)
b = komb_amd.KombAccel(); b.from_edges(nv, uv)
print("second context: c2 k-core first wall", wall(b.core_run), "then x10", wall(b.core_run, 10), wall(b.core_run, 10), flush=True)
