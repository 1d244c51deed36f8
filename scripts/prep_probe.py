"""Times of the k-truss preparation (truss_prep.hip) beside the step it belongs to.  usage: prep_probe.py [config] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import komb_amd, bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nv, ncl, alpha, seed = bench.CONFIGS[cfg][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
a = komb_amd.KombAccel()
t0 = time.perf_counter(); a.from_edges(nv, uv); t1 = time.perf_counter()
s = a.stats()
print(f"{cfg}: from_edges {1e3*(t1-t0):.1f} ms (ms_build {s['ms_build']:.1f}, h2d {s['ms_build_h2d']:.1f})", flush=True)
del uv
for i in range(reps):
    a.truss_unprepare()
    t0 = time.perf_counter(); a.truss_run(); t1 = time.perf_counter()
    s = a.stats()
    print(f"cold step {i}: wall {1e3*(t1-t0):.2f} ms; prepare {s['ms_prepare']:.2f} (vertex {s['ms_prep_vertex']:.2f} edges {s['ms_prep_edges']:.2f} rows {s['ms_prep_rows']:.2f}) enum {s['ms_tri_fill']:.2f} sort {s['ms_sort']:.2f} finish {s['ms_compact']:.2f} "
          f"peel {s['ms_peel']:.2f} gather {s['ms_gather']:.2f}; T={s['triangles']} tmax={s['max_trussness']}", flush=True)
for i in range(reps):
    t0 = time.perf_counter(); a.truss_run(); t1 = time.perf_counter()
    s = a.stats()
    print(f"resident step {i}: wall {1e3*(t1-t0):.2f} ms; prepare {s['ms_prepare']:.2f} (prepared {s['truss_prepared']})", flush=True)
a.core_run(); print("k-core ms", a.stats()["ms_core"], flush=True)
import hashlib
eu, ev, tr, sup = a.truss_fetch(with_support=True)
print("sha256 trussness", hashlib.sha256(tr.tobytes()).hexdigest()[:16], "support", hashlib.sha256(sup.tobytes()).hexdigest()[:16], flush=True)
