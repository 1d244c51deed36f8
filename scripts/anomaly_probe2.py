"""Round 5 debugging aid: bench.py's sequence of extras at C3, every call timed by wall clock and by the library's events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import komb_amd, bench
def wall(f, n=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / n * 1e3, 2)
nv, ncl, alpha, seed = bench.CONFIGS["c3"][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
acc = komb_amd.KombAccel(); acc.from_edges(nv, uv)
for i in range(2):
    acc.truss_unprepare(); print("cold step", wall(acc.truss_run), flush=True)
print("k-core", wall(acc.core_run), acc.stats()["ms_core"], flush=True)
deg, core = acc.core_fetch()
t0 = time.perf_counter(); acc.get_anomaly_score(deg, core); print("corea wall", round((time.perf_counter() - t0) * 1e3, 1), "device", acc.stats()["ms_corea"], flush=True)
t0 = time.perf_counter(); acc.get_anomaly_score(deg, core); print("corea wall", round((time.perf_counter() - t0) * 1e3, 1), "device", acc.stats()["ms_corea"], flush=True)
mask = (core == core.max()).astype(np.uint8)
acc.set_option("POOL_DEBUG", "1")
for i in range(4):
    w = wall(lambda: acc.truss_run(mask)); s = acc.stats()
    print("faithful wall", w, {k: round(v, 3) for k, v in s.items() if k.startswith("ms_") and v and k not in ("ms_build", "ms_build_h2d", "ms_core", "ms_core_local", "ms_corea")}, flush=True)
acc.set_option("POOL_DEBUG", None)
print("k-core again", wall(acc.core_run), wall(acc.core_run, 5), flush=True)
fresh = komb_amd.KombAccel(); print("fresh build", wall(lambda: fresh.from_edges(nv, uv)), "core", wall(fresh.core_run), "truss", wall(fresh.truss_run), flush=True)
fresh.close(); del uv
nv2, ncl2 = bench.CONFIGS["c2"][:2]
uv2 = komb_amd.gen_hug_edges(nv2, ncl2, 2.6, 42)
a2 = komb_amd.KombAccel(); a2.from_edges(nv2, uv2)
print("c2 k-core first", wall(a2.core_run))
for i in range(4): print("c2 k-core x10", wall(a2.core_run, 10), "event", round(a2.stats()["ms_core"], 2), flush=True)
