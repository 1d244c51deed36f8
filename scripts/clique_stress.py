"""One-off stress: big cliques (non-staged triangle path, heavy units, the 32-bit incidence limit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import komb_amd
for n in (1200, 2100):
    iu = np.triu_indices(n, 1)
    uv = np.stack(iu, axis=1).astype(np.int64)
    with komb_amd.KombAccel() as a:
        a.from_edges(n, uv)
        t = time.time(); deg, core = a.run_core(); tc = time.time() - t
        print(f"K_{n}: ne={a.ne} core ok={bool(np.all(core == n - 1))} {tc*1e3:.1f} ms", flush=True)
        t = time.time()
        try:
            eu, ev, tr, sup = a.run_truss(with_support=True)
            st = a.stats()
            print(f"   truss ok={bool(np.all(tr == n) and np.all(sup == n - 2))} T={st['triangles']} expected {n*(n-1)*(n-2)//6} "
                  f"{(time.time()-t)*1e3:.0f} ms  phases tri {st['ms_tri_fill']:.1f} compact {st['ms_compact']:.1f} peel {st['ms_peel']:.1f}", flush=True)
        except komb_amd.KombError as e:
            print(f"   refused after {(time.time()-t)*1e3:.0f} ms: code {e.code}: {e}", flush=True)
