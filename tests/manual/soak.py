"""Randomised soak: many graphs of assorted shapes, k-core and k-truss (with support) against the CPU oracle,
each under a random choice of the finish (local fixed point / LDS tails / none), its hand-over thresholds and item
limit, and of the index layout (record stream / two pass, dense or block-less
own-role entries, a dense region or a record stream that runs out), the period of the engine's RETIRE step, the fixed point's
notification kernel, -- one graph in seven -- the sharded peel's engine with one rank (shard_dev.h), and -- one in five --
the sliced results of komb_truss_run_slice for a random rank of a random world size.
    python tests/manual/soak.py [n_graphs] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import komb_amd
import komb_amd.api
komb_amd.api.FORWARD_ENV_OPTIONS = True     # the switches below reach the library as per-context options
from oracle import oracle as O

n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
t0 = time.time()
bad = 0
with komb_amd.KombAccel() as a:
    for g in range(n_graphs):
        kind = rng.integers(0, 5)
        if kind == 0:                                   # sparse random
            nv = int(rng.integers(2, 4000)); ne = int(rng.integers(1, 6 * nv))
            uv = rng.integers(0, nv, (ne, 2))
        elif kind == 1:                                 # dense random
            nv = int(rng.integers(3, 400)); ne = int(rng.integers(nv, nv * nv // 3 + 2))
            uv = rng.integers(0, nv, (ne, 2))
        elif kind == 2:                                 # union of cliques + noise (unitig-graph like)
            nv = int(rng.integers(20, 3000)); parts = []
            for _ in range(int(rng.integers(1, 60))):
                k = int(rng.integers(2, 40)); vs = rng.choice(nv, size=min(k, nv), replace=False)
                iu = np.triu_indices(len(vs), 1); parts.append(np.stack([vs[iu[0]], vs[iu[1]]], axis=1))
            parts.append(rng.integers(0, nv, (int(rng.integers(0, 2 * nv)), 2)))
            uv = np.concatenate(parts)
        elif kind == 3:                                 # generator of the benchmark
            nv = int(rng.integers(100, 30000))
            uv = np.asarray(komb_amd.gen_hug_edges(nv, int(nv * rng.uniform(1.0, 4.0)), float(rng.uniform(2.1, 3.0)), int(rng.integers(1, 1 << 30)))).reshape(-1, 2)
        else:                                           # hub + core: a star over a dense core
            nv = int(rng.integers(50, 5000)); c = int(rng.integers(5, min(nv, 300)))
            iu = np.triu_indices(c, 1); core = np.stack(iu, axis=1)
            keep = rng.random(len(core)) < rng.uniform(0.2, 1.0)
            star = np.stack([np.zeros(nv - 1, np.int64), np.arange(1, nv)], axis=1)
            uv = np.concatenate([core[keep], star, rng.integers(0, nv, (nv, 2))])
        uv = np.ascontiguousarray(uv, dtype=np.int64)
        os.environ["KOMB_FINISH"] = str(rng.choice(["local", "lds", "none"]))
        os.environ["KOMB_LOCAL_LIMIT"] = str(rng.choice([0, 40, 900, 20000, 4000000000]))
        os.environ["KOMB_TAIL"] = str(rng.choice([0, 50, 700, 5000, 32768, 65534]))
        os.environ["KOMB_CORE_TAIL"] = str(rng.choice([0, 9, 200, 1024]))
        dens = str(rng.choice(["", "0", "3", "40"]))                     # density rule of the k-truss local finish ("" = default 160, 0 = off)
        if dens: os.environ["KOMB_LOCAL_DENSITY"] = dens
        else: os.environ.pop("KOMB_LOCAL_DENSITY", None)
        items = str(rng.choice(["", "0", "60", "5000", "200000"]))       # item limit of the local finish ("" = the default): small ones exercise the refusal
        if items: os.environ["KOMB_LOCAL_ITEMS"] = items
        else: os.environ.pop("KOMB_LOCAL_ITEMS", None)
        for k, on in (("KOMB_TWO_PASS", rng.random() < 0.15), ("KOMB_NO_OWN_DENSE", rng.random() < 0.3),
                      ("KOMB_SHARD_PEEL", rng.random() < 0.15)):
            if on: os.environ[k] = "1"
            else: os.environ.pop(k, None)
        os.environ["KOMB_INDEX"] = str(rng.choice(["stream", "stream", "stream", "two_pass"]))
        for k, choices in (("KOMB_OWN_DENSE_CAP", ["", "", "0", "200", "5000"]), ("KOMB_REC_CAP", ["", "", "", "100", "3000"]), ("KOMB_RETIRE_EVERY", ["", "", "1", "3", "40"]),
                           ("KOMB_LOCAL_DEFER_CHUNKS", ["", "1"])):
            v = str(rng.choice(choices))
            if v: os.environ[k] = v
            else: os.environ.pop(k, None)
        a.from_edges(nv, uv)
        rowptr, col = a.get_csr()
        deg, core = a.run_core()
        eu, ev, tr, sup = a.run_truss(with_support=True)
        ok = (np.array_equal(core, O.coreness(rowptr, col)) and np.array_equal(sup, O.support(rowptr, col)[0])
              and np.array_equal(tr, O.trussness(rowptr, col)))
        if ok and rng.random() < 0.2 and len(tr):                  # the sliced results of komb_truss_run_slice (bench.py --gpus N)
            world = int(rng.integers(1, 9)); rank = int(rng.integers(0, world))
            a.truss_run_slice(rank, world)
            seu, sev, stra, ssup = a.truss_fetch(with_support=True)
            lo, hi = len(tr) * rank // world, len(tr) * (rank + 1) // world
            ok = (np.array_equal(seu, eu) and np.array_equal(sev, ev) and np.array_equal(stra[lo:hi], tr[lo:hi]) and np.array_equal(ssup[lo:hi], sup[lo:hi])
                  and not stra[:lo].any() and not stra[hi:].any() and not ssup[:lo].any() and not ssup[hi:].any())
        if not ok:
            bad += 1
            np.save(f"gpurun_out/soak_fail_{g}.npy", uv)
            print(f"MISMATCH graph {g} kind {kind} nv {nv} env FINISH={os.environ['KOMB_FINISH']} LOCAL_LIMIT={os.environ['KOMB_LOCAL_LIMIT']} "
                  f"TAIL={os.environ['KOMB_TAIL']} CORE_TAIL={os.environ['KOMB_CORE_TAIL']} "
                  f"TWO_PASS={os.environ.get('KOMB_TWO_PASS')} NO_OWN_DENSE={os.environ.get('KOMB_NO_OWN_DENSE')} LOCAL_ITEMS={os.environ.get('KOMB_LOCAL_ITEMS')} "
                  f"INDEX={os.environ.get('KOMB_INDEX')} OWN_DENSE_CAP={os.environ.get('KOMB_OWN_DENSE_CAP')} REC_CAP={os.environ.get('KOMB_REC_CAP')} "
                  f"DEFER={os.environ.get('KOMB_LOCAL_DEFER_CHUNKS')} SHARD_PEEL={os.environ.get('KOMB_SHARD_PEEL')} RETIRE={os.environ.get('KOMB_RETIRE_EVERY')}", flush=True)
        if g % 100 == 99:
            print(f"{g + 1} graphs, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n_graphs} graphs, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
