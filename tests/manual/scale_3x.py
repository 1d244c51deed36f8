"""Scale / stress run beyond the bench configuration.  Default: |V| = 30M, |E| ~ 300M (3x C3); arguments
`nv n_cliques alpha` select another shape (SURVEY 8(d)'s stress variant: 10000000 27500000 2.2).  Checks the
size-independent properties the suite checks at C2 (support >= trussness - 2 inside every k-truss, coreness bound)
and prints times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import komb_amd

nv, nc, alpha = 30_000_000, 72_750_000, 2.6
if len(sys.argv) > 3:
    nv, nc, alpha = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
t0 = time.time()
uv = komb_amd.gen_hug_edges(nv, nc, alpha, 42)
print(f"generated {len(uv) // 2} raw pairs in {time.time() - t0:.1f} s", flush=True)
with komb_amd.KombAccel() as a:
    t0 = time.time(); a.from_edges(nv, uv); del uv
    print(f"graph build {time.time() - t0:.2f} s", flush=True)
    deg, core = a.run_core()
    st = a.stats()
    print(f"k-core {st['ms_core']:.1f} ms (local finish {st['ms_core_local']:.1f} ms, {st['core_local_units']} units, {st['core_local_sweeps']} sweeps), max coreness {st['max_coreness']}, |E| = {st['ne']}", flush=True)
    assert core.max() <= deg.max() and np.all(core <= deg)
    for i in range(2):
        a.truss_run()
        st = a.stats()
        print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items() if k.startswith('ms_') or k.startswith('truss') or k in ('triangles', 'max_trussness', 'max_degree')}, flush=True)
    eu, ev, tr, sup = a.run_truss(with_support=True)
    assert sup.sum() == 3 * st["triangles"]
    assert np.all(tr >= 2) and np.all(tr <= sup + 2)
    assert np.all(tr[sup == 0] == 2)
    # an edge of trussness t has both endpoints of coreness >= t - 1
    assert np.all(np.minimum(core[eu], core[ev]) >= tr - 1)
    total = st['ms_orient'] + st['ms_support'] + st['ms_peel'] + st['ms_gather']
    print(f"k-truss {total:.1f} ms -> {st['ne'] / total / 1e6:.2f} G edges/s; properties hold", flush=True)
