"""Times of the k-truss / k-core path on graph shapes other than the benchmark's (heavier / lighter degree tails,
dense uniform random, one large clique inside a sparse graph): a guard against shape-specific cliffs, e.g. in the
single-workgroup tail kernels.  Results are checked against the oracle where the oracle is fast enough."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import komb_amd
import komb_amd.api; komb_amd.api.FORWARD_ENV_OPTIONS = True   # KOMB_* switches reach the library as per-context options
from oracle import oracle as O

rng = np.random.default_rng(3)
cases = []
for alpha in (2.1, 2.3, 3.0):
    cases.append((f"hug 1M alpha {alpha}", 1_000_000, np.asarray(komb_amd.gen_hug_edges(1_000_000, 2_450_000, alpha, 5)).reshape(-1, 2), False))
cases.append(("uniform random 200k x 8M", 200_000, rng.integers(0, 200_000, (8_000_000, 2)), False))
cases.append(("uniform random 20k x 4M (dense)", 20_000, rng.integers(0, 20_000, (4_000_000, 2)), False))
k = 250
iu = np.triu_indices(k, 1)
cases.append(("K_250 inside 1M sparse", 1_000_000, np.concatenate([np.stack(iu, axis=1) * 3999, rng.integers(0, 1_000_000, (3_000_000, 2))]), True))
kk = 180
iu = np.triu_indices(kk, 1)
cases.append(("40 x K_180 disjoint", 40 * kk, np.concatenate([np.stack(iu, axis=1) + i * kk for i in range(40)]), True))
modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["default"]     # KOMB_FINISH values to compare
with komb_amd.KombAccel() as a:
    for name, nv, uv, check in cases:
      uv = np.ascontiguousarray(uv, dtype=np.int64)
      a.from_edges(nv, uv)
      for mode in modes:
        if mode == "default": os.environ.pop("KOMB_FINISH", None)
        else: os.environ["KOMB_FINISH"] = mode
        a.truss_run(); a.core_run()                     # warm the pool
        a.truss_run(); st = a.stats()
        a.core_run(); sc = a.stats()
        tot = st["ms_orient"] + st["ms_support"] + st["ms_peel"] + st["ms_gather"]
        line = (f"{name:34s} [{mode:7s}] |E| {st['ne']:9d} T {st['triangles']:10d} tmax {st['max_trussness']:4d}  truss {tot:7.2f} ms "
                f"(tri {st['ms_tri_fill'] + st['ms_tri_count']:.2f} peel {st['ms_peel']:.2f} tail {st['ms_tail']:.2f} x{st['truss_tail_runs']} "
                f"local {st['ms_truss_local']:.2f}/{st['truss_local_sweeps']})"
                f"  core {sc['ms_core']:.2f} ms (local {sc['ms_core_local']:.2f}/{sc['core_local_sweeps']}) kmax {sc['max_coreness']}")
        if check:
            rowptr, col = a.get_csr()
            _, _, tr = a.truss_fetch()
            deg, core = a.core_fetch()
            ok = np.array_equal(tr, O.trussness(rowptr, col)) and np.array_equal(core, O.coreness(rowptr, col))
            line += f"  parity {ok}"
        print(line, flush=True)
