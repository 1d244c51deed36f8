import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, komb_amd
nv, ncl = 10_000_000, 24_250_000
uv = komb_amd.gen_hug_edges(nv, ncl, 2.6, 42)
a = komb_amd.KombAccel()
for i in range(3):
    t0=time.perf_counter(); a.from_edges(nv, uv); t=time.perf_counter()-t0
    st=a.stats()
    print(f"from_edges wall {t*1e3:.1f} ms  ms_build {st['ms_build']:.1f} h2d {st['ms_build_h2d']:.1f} relabel {st['ms_build_relabel']:.1f}")
t0=time.perf_counter(); a.truss_run(); print("first truss", (time.perf_counter()-t0)*1e3)
t0=time.perf_counter(); a.truss_run(); print("second truss", (time.perf_counter()-t0)*1e3)
