"""Sweep of the hand-over rule of the k-truss local finish (options LOCAL_DENSITY / LOCAL_ITEMS / LOCAL_LIMIT / FINISH) on one
shape: `tail_sweep.py nv n_cliques alpha`.  Prints the peel time and what the finish took for every setting; the trussness
vector of every setting is compared with the first one's."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import komb_amd

nv, ncl, alpha = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
SETTINGS = [s for s in (sys.argv[4] if len(sys.argv) > 4 else "").split(";") if s] or [
    "", "FINISH=none", "LOCAL_DENSITY=0", "LOCAL_DENSITY=0,LOCAL_ITEMS=67108864", "LOCAL_DENSITY=0,LOCAL_ITEMS=134217728",
    "LOCAL_DENSITY=0,LOCAL_ITEMS=268435456", "LOCAL_DENSITY=400", "LOCAL_DENSITY=400,LOCAL_ITEMS=134217728"]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, 42)
ref = None
with komb_amd.KombAccel() as a:
    a.from_edges(nv, uv); del uv
    a.truss_run()
    for setting in SETTINGS:
        kv = dict(x.split("=") for x in setting.split(",") if "=" in x)
        for k, v in kv.items(): a.set_option(k, v)
        best = None
        for _ in range(3):
            a.truss_run()
            st = a.stats()
            if best is None or st["ms_peel"] < best["ms_peel"]: best = st
        tr = a.truss_fetch()[2] if hasattr(a, "truss_fetch") else a.run_truss()[2]
        h = hashlib.sha256(np.ascontiguousarray(tr)).hexdigest()[:12]
        if ref is None: ref = h
        print(f"{setting or 'default':50s} peel {best['ms_peel']:7.2f} ms (local {best['ms_truss_local']:6.2f} ms: {best['truss_local_units']} units, {best['truss_local_items']} items, "
              f"{best['truss_local_sweeps']} sweeps); {best['truss_subrounds']} sub-rounds; {'same' if h == ref else 'DIFFERENT'} trussness", flush=True)
        for k in kv: a.set_option(k, None)
