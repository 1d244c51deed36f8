"""Maximum-size run: more than 2^31 sort keys (1.15 G raw pairs -> 2.3 G directed keys), more than 2^31 CSR slots,
~1.1 G edges -- the 32-bit *unsigned* slot range and the 64-bit item counts of the sort / unique primitives used
naturally.  Uniform random pairs (few triangles: the bounded slices would need ~350 GB, so the index is built in two
passes).  Checks size-independent properties and prints times.  Needs ~40 GB of host and ~90 GB of device memory.
    python tests/manual/big_raw.py [n_raw] [nv]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import komb_amd
import komb_amd.api; komb_amd.api.FORWARD_ENV_OPTIONS = True   # KOMB_* switches reach the library as per-context options

n_raw = int(sys.argv[1]) if len(sys.argv) > 1 else 1_150_000_000
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
t0 = time.time()
rng = np.random.default_rng(7)
uv = np.empty((n_raw, 2), dtype=np.int64)
step = 50_000_000
for i in range(0, n_raw, step):                                  # in pieces: progress lines, bounded temporaries
    uv[i:i + step] = rng.integers(0, nv, (min(step, n_raw - i), 2), dtype=np.int64)
    if (i // step) % 5 == 4:
        print(f"generated {i + step:,} pairs, {time.time() - t0:.0f} s", flush=True)
with komb_amd.KombAccel() as a:
    t1 = time.time(); a.from_edges(nv, uv); del uv
    st = a.stats()
    print(f"graph build {time.time() - t1:.1f} s (device part {st['ms_build']:.0f} ms): |V| = {nv:,}, |E| = {st['ne']:,}, slots = {2 * st['ne']:,} "
          f"(2^31 = {2**31:,})", flush=True)
    assert 2 * st["ne"] > 2**31, "the graph was meant to have more than 2^31 slots"
    deg, core = a.run_core()
    st = a.stats()
    print(f"k-core {st['ms_core']:.1f} ms (local finish {st['ms_core_local']:.1f} ms, {st['core_local_units']:,} units), max coreness {st['max_coreness']}", flush=True)
    assert int(deg.sum(dtype=np.int64)) == 2 * st["ne"] and np.all(core <= deg) and core.max() == st["max_coreness"]
    os.environ["KOMB_FINISH"] = "none"
    core2 = a.run_core()[1]
    os.environ.pop("KOMB_FINISH")
    assert np.array_equal(core, core2), "general engine and local finish disagree"
    del core2
    eu, ev, tr, sup = a.run_truss(with_support=True)
    st = a.stats()
    print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in st.items() if k.startswith("ms_") or k in ("triangles", "max_trussness", "max_degree")}, flush=True)
    assert sup.sum(dtype=np.int64) == 3 * st["triangles"]
    assert np.all(tr >= 2) and np.all(tr <= sup + 2) and np.all(tr[sup == 0] == 2)
    assert np.all(eu < ev) and np.all(np.minimum(core[eu], core[ev]) >= tr - 1)
    total = st["ms_orient"] + st["ms_support"] + st["ms_peel"] + st["ms_gather"]
    print(f"k-truss {total:.1f} ms -> {st['ne'] / total / 1e6:.2f} G edges/s; properties hold; wall {time.time() - t0:.0f} s", flush=True)
