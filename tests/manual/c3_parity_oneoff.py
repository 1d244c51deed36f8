"""One-off: full-size C3 (|V|=10M, |E|~100M) GPU results vs the CPU oracle, every value.
Too slow for the test suite (the single-thread oracle needs ~10 min); run by hand:
    python tests/manual/c3_parity_oneoff.py > gpurun_out/c3_parity.log
Another shape of the same generator: `c3_parity_oneoff.py nv n_cliques alpha [oracle threads]` (with threads > 1 the
trussness oracle is the OpenMP variant, which tests/test_oracle.py ties to the sequential one).
"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import komb_amd
from oracle import oracle as O

import threading
def _heartbeat(t0=time.time()):                      # a line a minute: the single-thread oracle passes are silent for longer than that
    while True:
        time.sleep(60)
        print(f"  ... {time.time() - t0:.0f} s", flush=True)
threading.Thread(target=_heartbeat, daemon=True).start()

nv, ncl, alpha, threads = 10_000_000, 24_250_000, 2.6, 1
if len(sys.argv) > 3:
    nv, ncl, alpha = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    threads = int(sys.argv[4]) if len(sys.argv) > 4 else 1
t = time.time(); uv = komb_amd.gen_hug_edges(nv, ncl, alpha, 42); print("gen", round(time.time() - t, 1), "s", flush=True)
a = komb_amd.KombAccel(); a.from_edges(nv, uv); del uv
rowptr, col = a.get_csr()
print("graph nv", a.nv, "ne", a.ne, flush=True)
deg, core = a.run_core()
eu, ev, tr, sup = a.run_truss(with_support=True)
st = a.stats()
print("gpu: sub-rounds", st["truss_subrounds"], "levels", st["truss_levels"], "index layout", st["index_layout"], "records", st["tri_records"], flush=True)
print("gpu: T", st["triangles"], "kmax", core.max(), "tmax", tr.max(), "sha256(truss)", hashlib.sha256(tr.tobytes()).hexdigest()[:16],
      "sha256(core)", hashlib.sha256(core.tobytes()).hexdigest()[:16], flush=True)
t = time.time(); ocore = O.coreness(rowptr, col); print("oracle coreness", round(time.time() - t, 1), "s  equal:", bool(np.array_equal(core, ocore)), flush=True)
t = time.time(); oeu, oev = O.edge_list(rowptr, col); print("edge order equal:", bool(np.array_equal(eu, oeu) and np.array_equal(ev, oev)), flush=True)
t = time.time(); osup, otri = O.support(rowptr, col); print("oracle support", round(time.time() - t, 1), "s  equal:", bool(np.array_equal(sup, osup)), "T equal:", otri == st["triangles"], flush=True)
t = time.time(); otr = O.trussness_native(rowptr, col, threads) if threads > 1 else O.trussness(rowptr, col); print("oracle trussness (%d thread%s)" % (threads, "s" if threads > 1 else ""), round(time.time() - t, 1), "s  equal:", bool(np.array_equal(tr, otr)), flush=True)
print("C3_PARITY", "OK" if (np.array_equal(core, ocore) and np.array_equal(sup, osup) and np.array_equal(tr, otr)) else "MISMATCH")
