"""Debug build (EXTRA=-DKOMB_DEBUG_SWITCHES): time of the truss peel's FIRST giant step at C3 under KOMB_PEEL_ABLATE
(1 no decrements, 2 no state gathers, 4 no index loads; results are wrong on purpose -- only the first step is compared,
whose input does not depend on what earlier steps did).  KOMB_ACCEL_LIB=komb_amd/libv/dbg/libkomb_accel.so python scripts/peel_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import komb_amd, bench
import komb_amd.api; komb_amd.api.FORWARD_ENV_OPTIONS = True   # KOMB_* switches reach the library as per-context options
nv, ncl, alpha, seed = bench.CONFIGS["c3"][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
a = komb_amd.KombAccel(); a.from_edges(nv, uv); del uv
os.environ["KOMB_FINISH"] = "none"      # (no hand-over: a finish would be given inconsistent supports)
a.truss_run()
out = "gpurun_out/peel_ablate_trace.txt"
for ab in (0, 1, 2, 3, 4, 6, 7, 0):
    os.environ["KOMB_PEEL_ABLATE"] = str(ab)
    for rep in range(2):
        if os.path.exists(out): os.remove(out)
        os.environ["KOMB_PEEL_TRACE"] = out
        try: a.truss_run()
        except Exception as e: pass
        os.environ.pop("KOMB_PEEL_TRACE")
        rows = [l.split() for l in open(out) if not l.startswith('#')]
        first = [r for r in rows if r[0] == '1'][0]
        print(f"ablate {ab}: first step light={first[3]} heavy={first[4]} {float(first[8]):.0f} us", flush=True)
