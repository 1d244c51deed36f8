"""Quick check of the three index layouts (and the first-frontier queue) against the oracle on one 40k-vertex graph; prints the
layout that ran, parity, and the peel's statistics.  usage: python scripts/dbg_layouts.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, komb_amd
from oracle import oracle as O
uv = komb_amd.gen_hug_edges(40000, 110000, 2.3, 21)
with komb_amd.KombAccel() as a:
    a.from_edges(40000, uv)
    rowptr, col = a.get_csr()
    otr = O.trussness(rowptr, col); osup = O.support(rowptr, col)[0]
    for env in ({}, {"KOMB_TWO_PASS": "1"}, {"KOMB_INDEX": "slices"}, {}, {"KOMB_TWO_PASS": "1"}, {"KOMB_NO_FIRST_QUEUE": "1"}):
        for k in ("KOMB_TWO_PASS", "KOMB_INDEX", "KOMB_NO_FIRST_QUEUE"): os.environ.pop(k, None)
        os.environ.update(env)
        eu, ev, tr, sup = a.run_truss(with_support=True)
        st = a.stats()
        print(env, "layout", st["index_layout"], "tr ok", np.array_equal(tr, otr), "sup ok", np.array_equal(sup, osup), "bad tr", int((tr != otr).sum()), "levels", st["truss_levels"], "scans", st["truss_scans"], "local units", st["truss_local_units"], flush=True)
    os.environ["KOMB_TWO_PASS"] = "1"
    eu, ev, tr, sup = a.run_truss(with_support=True)
    print("two-pass tr unique", np.unique(tr)[:10], "stats", {k: v for k, v in a.stats().items() if k.startswith("truss") or k in ("max_trussness",)})
