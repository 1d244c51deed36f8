"""One-off: moderately dense random graphs (16-vertex tasks exceed the LDS budget, single rows do not)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import komb_amd
from oracle import oracle as O
rng = np.random.default_rng(3)
for nv, ne in ((5000, 400_000), (20000, 3_000_000)):
    uv = rng.integers(0, nv, (ne, 2)).astype(np.int64)
    with komb_amd.KombAccel() as a:
        a.from_edges(nv, uv)
        a.truss_run()
        t = time.time(); eu, ev, tr, sup = a.run_truss(with_support=True); dt = time.time() - t
        st = a.stats()
        line = f"nv={nv} ne={a.ne} T={st['triangles']} tmax={tr.max()} truss {dt*1e3:.1f} ms (tri {st['ms_tri_fill']:.1f} compact {st['ms_compact']:.1f} peel {st['ms_peel']:.1f})"
        if a.ne <= 500_000:
            rowptr, col = a.get_csr()
            line += f"  parity={bool(np.array_equal(tr, O.trussness(rowptr, col)))}"
        print(line, flush=True)
