#!/usr/bin/env python3
"""How the peel's finish (KOMB_FINISH / KOMB_LOCAL_LIMIT) moves the k-core and k-truss times of a bench config.
usage: finish_sweep.py <config> [reps]   (GPU box; prints one line per setting)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import komb_amd  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nv, ncl, alpha, seed = bench.CONFIGS[cfg][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
acc = komb_amd.KombAccel()
acc.from_edges(nv, uv)
del uv
m = acc.ne
settings = [("none", None), ("lds", None), ("local", None)]
for div in (2, 4, 8, 16, 32, 64, 128, 512):
    settings.append(("local", div))
ref_core = ref_tr = None
for fin, div in settings:
    os.environ["KOMB_FINISH"] = fin
    for what in ("core", "truss"):
        units = nv if what == "core" else m
        if div is None:
            os.environ.pop("KOMB_LOCAL_LIMIT", None)
        else:
            os.environ["KOMB_LOCAL_LIMIT"] = str(units // div)
        best = None
        for _ in range(reps):
            if what == "core":
                acc.core_run()
            else:
                acc.truss_run()
            st = acc.stats()
            t = st["ms_core"] if what == "core" else st["ms_peel"]
            if best is None or t < best[0]:
                best = (t, st)
        t, st = best
        if what == "core":
            core = acc.core_fetch()[1]
            if ref_core is None:
                ref_core = core
            same = bool(np.array_equal(core, ref_core))
            print(f"{cfg} core  finish={fin:5s} limit=units/{div}: {t:8.3f} ms  launches {st['core_launches']:4d} local: units {st['core_local_units']:8d} "
                  f"items {st['core_local_items']:10d} sweeps {st['core_local_sweeps']:3d} ms {st['ms_core_local']:7.3f}  same={same}", flush=True)
        else:
            tr = acc.truss_fetch()[2]
            if ref_tr is None:
                ref_tr = tr
            same = bool(np.array_equal(tr, ref_tr))
            print(f"{cfg} truss finish={fin:5s} limit=units/{div}: peel {t:8.3f} ms launches {st['truss_launches']:4d} local: units {st['truss_local_units']:8d} "
                  f"items {st['truss_local_items']:10d} sweeps {st['truss_local_sweeps']:3d} ms {st['ms_truss_local']:7.3f} tail {st['ms_tail']:.3f}  same={same}", flush=True)
