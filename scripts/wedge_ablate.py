"""Debug (library built with -DKOMB_DEBUG_SWITCHES, KOMB_ACCEL_LIB): time the wedge enumeration with parts switched off
(KOMB_TRI_ABLATE: 1 = survivors dropped, 2 = no candidate tests, 4 = no line loads; results are wrong on purpose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import komb_amd
import komb_amd.api; komb_amd.api.FORWARD_ENV_OPTIONS = True   # KOMB_* switches reach the library as per-context options
nv, ncl = 10_000_000, 24_250_000
uv = komb_amd.gen_hug_edges(nv, ncl, 2.6, 42)
a = komb_amd.KombAccel(); a.from_edges(nv, uv); del uv
a.truss_run()
for ab in (0, 1, 8, 2, 4, 0):
    os.environ["KOMB_TRI_ABLATE"] = str(ab)
    try:
        a.truss_run()
    except Exception as e:
        print(ab, "error", str(e)[:60])
    s = a.stats()
    print(f"ablate={ab}: lines {s['ms_orient']:.2f} fill {s['ms_tri_fill']:.2f} sort {s['ms_sort']:.2f} compact {s['ms_compact']:.2f} records {s['tri_records']}", flush=True)
