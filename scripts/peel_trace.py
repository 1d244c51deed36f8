"""Debug (needs a build with EXTRA=-DKOMB_DEBUG_SWITCHES): one k-truss and one k-core pass with KOMB_PEEL_TRACE,
general engine only (KOMB_FINISH=none), one line per step in gpurun_out/peel_trace_<config>.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import komb_amd
import komb_amd.api; komb_amd.api.FORWARD_ENV_OPTIONS = True   # KOMB_* switches reach the library as per-context options
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
if len(sys.argv) > 3:                  # a shape instead of a configuration: nv n_cliques alpha
    nv, ncl, alpha, seed = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), 42
    name = f"{nv}_{ncl}_{alpha}"
else:
    nv, ncl, alpha, seed = bench.CONFIGS[name][:4]
out = f"gpurun_out/peel_trace_{name}.txt"
os.makedirs("gpurun_out", exist_ok=True)
if os.path.exists(out): os.remove(out)
os.environ["KOMB_FINISH"] = os.environ.get("KOMB_FINISH", "none")
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
a = komb_amd.KombAccel(); a.from_edges(nv, uv); del uv
a.truss_run()                      # warm the pool
a.core_run()
a.truss_run()
st = a.stats()
print("untraced:", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items() if k in ("ms_peel", "ms_truss_local", "truss_local_items", "truss_local_units", "truss_local_sweeps", "truss_subrounds", "truss_levels", "truss_launches", "truss_scans", "max_trussness", "ms_core", "ms_core_local")}, flush=True)
os.environ["KOMB_PEEL_TRACE"] = out
a.truss_run()
a.core_run()
st = a.stats()
print("traced:", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items() if k in ("ms_peel", "ms_truss_local", "truss_local_items", "truss_local_units", "truss_local_sweeps", "truss_subrounds", "truss_levels", "truss_launches", "truss_scans", "max_trussness", "ms_core", "ms_core_local")}, flush=True)
