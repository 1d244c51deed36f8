#!/usr/bin/env python3
"""A/B timing of komb_core_run under environment switches.  usage: core_ab.py <config> [ENV=VAL ...] (each ENV=VAL is one variant)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import komb_amd
import bench

cfg = sys.argv[1]
variants = [""] + sys.argv[2:]
nv, ncl, alpha, seed = bench.CONFIGS[cfg][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
with komb_amd.KombAccel() as a:
    a.from_edges(nv, uv)
    ref = None
    for rep in range(2):
        for v in variants:
            kv = dict(x.split("=", 1) for x in v.split(",") if x)
            for k, val in kv.items():
                os.environ[k] = val
            ts = []
            for _ in range(6):
                a.core_run()
                ts.append(a.stats()["ms_core"])
            deg, core = a.core_fetch()
            if ref is None:
                ref = core
            assert np.array_equal(core, ref)
            st = a.stats()
            print(f"{cfg} [{v or 'default'}]: ms_core min {min(ts):.3f} med {sorted(ts)[len(ts)//2]:.3f}  local {st['ms_core_local']:.3f} launches {st['core_launches']}", flush=True)
            for k in kv:
                del os.environ[k]
