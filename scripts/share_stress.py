#!/usr/bin/env python3
"""Two (or more) processes on ONE GPU at the same time, each decomposing its own graph over and over and comparing every
result with its first one (and the coreness with the oracle): shakes out anything that depends on timing or on which CUs a
launch gets.  usage: share_stress.py <nproc> <iterations> [finish]   (spawns its workers itself)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def worker(idx, iters):
    import numpy as np
    import komb_amd
    import komb_amd.api
    komb_amd.api.FORWARD_ENV_OPTIONS = True      # KOMB_FINISH (argument 3) reaches the library as a per-context option
    from oracle import oracle as O
    nv = 300000
    uv = komb_amd.gen_hug_edges(nv, int(2.45 * nv), 2.6, 100 + idx)
    a = komb_amd.KombAccel(); a.from_edges(nv, uv)
    rowptr, col = a.get_csr()
    want_core = O.coreness(rowptr, col)
    ref = None
    bad = 0
    for it in range(iters):
        try:
            core = a.run_core()[1]
            tr = a.run_truss()[2]
        except Exception as e:      # noqa: BLE001
            print(f"worker {idx} iter {it}: ERROR {e}", flush=True); bad += 1; continue
        if not np.array_equal(core, want_core):
            print(f"worker {idx} iter {it}: coreness differs from the oracle in {int((core != want_core).sum())} places", flush=True); bad += 1
        if ref is None:
            ref = tr
        elif not np.array_equal(tr, ref):
            print(f"worker {idx} iter {it}: trussness differs from iteration 0 in {int((tr != ref).sum())} places", flush=True); bad += 1
    print(f"worker {idx}: {iters} iterations, {bad} bad", flush=True)
    return bad

if __name__ == "__main__":
    if sys.argv[1] == "worker":
        sys.exit(1 if worker(int(sys.argv[2]), int(sys.argv[3])) else 0)
    nproc, iters = int(sys.argv[1]), int(sys.argv[2])
    env = dict(os.environ)
    if len(sys.argv) > 3:
        env["KOMB_FINISH"] = sys.argv[3]
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(i), str(iters)], env=env) for i in range(nproc)]
    rc = [p.wait() for p in ps]
    print("share_stress", nproc, iters, env.get("KOMB_FINISH", "default"), "rc", rc, flush=True)
    sys.exit(1 if any(rc) else 0)
