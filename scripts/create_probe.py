"""komb_create / first graph build / first calls of a fresh process: where the one-time costs fall."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import komb_amd
nv, ncl = 10_000_000, 24_250_000
uv = komb_amd.gen_hug_edges(nv, ncl, 2.6, 42)
t0 = time.perf_counter(); a = komb_amd.KombAccel(); t_create = time.perf_counter() - t0
t0 = time.perf_counter(); a.from_edges(nv, uv); t_build = time.perf_counter() - t0
t0 = time.perf_counter(); a.core_run(); t_core = time.perf_counter() - t0
t0 = time.perf_counter(); a.truss_run(); t_truss = time.perf_counter() - t0
t0 = time.perf_counter(); a.truss_run(); t_truss2 = time.perf_counter() - t0
print(f"komb_create {t_create*1e3:.1f} ms, first build {t_build*1e3:.1f} ms, first k-core {t_core*1e3:.1f} ms, first k-truss {t_truss*1e3:.1f} ms, second {t_truss2*1e3:.1f} ms")
