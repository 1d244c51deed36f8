"""GPU parity tests: the HIP path, called through the C ABI, against the golden
fixtures and against the oracle on the same seeded inputs.  Integer results are
compared bit for bit; CoreA scores (f64) are compared bit for bit as well since
ranks are exact and the logarithm is the host libm on both sides."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O(built):
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def K(built):
    import komb_amd
    return komb_amd


def _i64(x):
    return np.asarray(x, dtype=np.int64)


def test_native_library_is_loaded(K):
    lib = K._lib.load()
    assert lib.komb_abi_version() == 7
    with open("/proc/self/maps") as f:
        assert "libkomb_accel.so" in f.read()


def test_golden_from_edges(K, golden):
    for g in golden:
        with K.KombAccel() as a:
            a.from_edges(g["nv"], _i64(g["raw"]).reshape(-1, 2))
            assert (a.nv, a.ne) == (g["nv"], len(g["eu"])), g["name"]
            rowptr, col = a.get_csr()
            assert rowptr.tolist() == g["rowptr"] and col.tolist() == g["col"], g["name"]
            deg, core = a.run_core()
            assert deg.tolist() == g["degree"], g["name"]
            assert core.tolist() == g["coreness"], g["name"]
            eu, ev, tr, sup = a.run_truss(with_support=True)
            assert eu.tolist() == g["eu"] and ev.tolist() == g["ev"], g["name"]
            assert sup.tolist() == g["support"], g["name"]
            assert tr.tolist() == g["trussness"], g["name"]
            assert a.stats()["triangles"] == g["triangles"], g["name"]


def test_golden_from_csr_and_maxcore_subgraph(K, golden):
    """runTruss's composition: trussness of the max-core induced subgraph."""
    for g in golden:
        with K.KombAccel() as a:
            a.from_csr(_i64(g["rowptr"]), np.asarray(g["col"], dtype=np.int32))
            deg, core = a.run_core()
            assert core.tolist() == g["coreness"], g["name"]
            mask = (core == core.max()).astype(np.uint8) if g["nv"] else np.zeros(0, np.uint8)
            assert mask.tolist() == g["maxcore_mask"], g["name"]
            eu, ev, tr = a.run_truss(mask)
            assert eu.tolist() == g["sub_eu"] and ev.tolist() == g["sub_ev"], g["name"]
            assert tr.tolist() == g["sub_trussness"], g["name"]


def test_golden_corea_reference_header(K, golden):
    for g in golden:
        if not g["nv"]:
            continue
        want = np.array([float.fromhex(h) for h in g["ref_corea_hex"]])
        with K.KombAccel() as a:
            got = a.get_anomaly_score(g["degree"], g["coreness"])
        assert np.array_equal(got, want), g["name"]


@pytest.mark.parametrize("nv,fac,alpha,seed", [
    (64, 2.0, 2.6, 1), (1000, 2.5, 2.6, 2), (1000, 6.0, 2.2, 3), (20000, 2.5, 2.6, 4),
    (50000, 3.3, 2.2, 5), (200000, 2.45, 2.6, 42),
])
def test_generated_graphs_vs_oracle(K, O, nv, fac, alpha, seed):
    uv = K.gen_hug_edges(nv, int(fac * nv), alpha, seed)
    o_rowptr, o_col = O.simplify(nv, uv)
    with K.KombAccel() as a:
        a.from_edges(nv, uv)
        rowptr, col = a.get_csr()
        assert np.array_equal(rowptr, o_rowptr) and np.array_equal(col, o_col)
        deg, core = a.run_core()
        assert np.array_equal(deg, O.degree(o_rowptr))
        o_core = O.coreness(o_rowptr, o_col)
        assert np.array_equal(core, o_core)
        eu, ev, tr, sup = a.run_truss(with_support=True)
        oeu, oev = O.edge_list(o_rowptr, o_col)
        assert np.array_equal(eu, oeu) and np.array_equal(ev, oev)
        osup, otri = O.support(o_rowptr, o_col)
        assert np.array_equal(sup, osup) and a.stats()["triangles"] == otri
        assert np.array_equal(tr, O.trussness(o_rowptr, o_col))
        # runTruss-faithful variant: max-core induced subgraph
        mask = (core == core.max()).astype(np.uint8)
        seu, sev, stra = a.run_truss(mask)
        weu, wev, wtr = O.trussness_induced(o_rowptr, o_col, mask)
        assert np.array_equal(seu, weu) and np.array_equal(sev, wev) and np.array_equal(stra, wtr)
        # CoreA
        score = a.get_anomaly_score(deg, core)
        assert np.array_equal(score, O.corea_scores(deg, core))
        rd, rk = a.fractional_ranks(deg, core)
        assert np.array_equal(rd, O.fractional_rank_fast(deg.astype(np.int64)))
        assert np.array_equal(rk, O.fractional_rank_fast(core.astype(np.int64) * nv + deg))


def test_random_dense_graphs_vs_oracle(K, O):
    """Erdos-Renyi-like inputs with loops and duplicates: long oriented rows,
    many triangles per edge, deep peel cascades."""
    rng = np.random.default_rng(17)
    for nv, ne in ((40, 600), (300, 9000), (2000, 60000)):
        uv = rng.integers(0, nv, (ne, 2)).astype(np.int64)
        o_rowptr, o_col = O.simplify(nv, uv)
        with K.KombAccel() as a:
            a.from_edges(nv, uv)
            deg, core = a.run_core()
            assert np.array_equal(core, O.coreness(o_rowptr, o_col))
            eu, ev, tr, sup = a.run_truss(with_support=True)
            osup, _ = O.support(o_rowptr, o_col)
            assert np.array_equal(sup, osup)
            assert np.array_equal(tr, O.trussness(o_rowptr, o_col))


def test_structured_cascades(K, O):
    """Long paths / ladders force one vertex or edge per sub-round."""
    n = 5000
    path = np.stack([np.arange(n - 1), np.arange(1, n)], axis=1)
    # triangle strip: (i,i+1),(i,i+2): every inner edge in 2 triangles, peels from both ends
    strip = np.concatenate([path, np.stack([np.arange(n - 2), np.arange(2, n)], axis=1)])
    for uv in (path, strip):
        uv = uv.astype(np.int64)
        o_rowptr, o_col = O.simplify(n, uv)
        with K.KombAccel() as a:
            a.from_edges(n, uv)
            deg, core = a.run_core()
            assert np.array_equal(core, O.coreness(o_rowptr, o_col))
            _, _, tr = a.run_truss()
            assert np.array_equal(tr, O.trussness(o_rowptr, o_col))


def test_edge_cases_and_errors(K):
    with K.KombAccel() as a:
        a.from_edges(0, np.zeros((0, 2), np.int64))
        assert (a.nv, a.ne) == (0, 0)
        deg, core = a.run_core()
        assert len(deg) == 0 and len(core) == 0
        eu, ev, tr = a.run_truss()
        assert len(eu) == 0
        a.from_edges(4, np.array([[1, 1], [2, 2]], np.int64))           # only loops
        assert (a.nv, a.ne) == (4, 0)
        deg, core = a.run_core()
        assert deg.tolist() == [0, 0, 0, 0] and core.tolist() == [0, 0, 0, 0]
        assert len(a.run_truss()[0]) == 0
        with pytest.raises(K.KombError) as e:
            a.from_edges(3, np.array([[0, 3]], np.int64))
        assert e.value.code == K._lib.KOMB_ERR_ARG
        with pytest.raises(K.KombError):                                 # graph was dropped by the failed load
            a.core_run()
        with pytest.raises(K.KombError):                                 # asymmetric CSR
            a.from_csr(np.array([0, 1, 1], np.int64), np.array([1], np.int32))
        a.from_edges(3, np.array([[0, 1], [1, 2], [0, 2]], np.int64))
        with pytest.raises(K.KombError) as e:
            a.truss_fetch()                                              # fetch before run
        assert e.value.code == K._lib.KOMB_ERR_STATE
        eu, ev, tr = a.run_truss(np.array([1, 1, 0], np.uint8))          # mask leaves one edge
        assert (eu.tolist(), ev.tolist(), tr.tolist()) == ([0], [1], [2])
        eu, ev, tr = a.run_truss(np.array([1, 0, 0], np.uint8))          # mask leaves no edge
        assert len(eu) == 0
        assert a.get_anomaly_score([], []).tolist() == []


def test_repeat_runs_identical(K):
    """Atomics reorder freely between runs; integer results must not move."""
    uv = K.gen_hug_edges(30000, 80000, 2.4, 8)
    with K.KombAccel() as a:
        a.from_edges(30000, uv)
        c1 = a.run_core()[1]
        t1 = a.run_truss()[2]
        for _ in range(3):
            assert np.array_equal(a.run_core()[1], c1)
            assert np.array_equal(a.run_truss()[2], t1)


def _oracle_trussness_fast(O, rowptr, col):
    """Every trussness value from the oracle: the OpenMP variant of the native build when this box can build it
    (tests/test_oracle.py ties it to orc_trussness), else the single-thread restatement."""
    if O.native_lib() is not None:
        import os
        return O.trussness_native(rowptr, col, min(16, len(os.sched_getaffinity(0))))
    return O.trussness(rowptr, col)


def test_two_contexts_in_two_threads(K, O):
    """Reentrancy: two contexts of one process, each driven by its own thread at the same time (ctypes drops the GIL inside
    the calls), different graphs; every run of both must equal the oracle.  A context owns its stream, pool, pinned staging
    and control blocks; nothing in the library is process-global."""
    import threading
    graphs = [(50000, K.gen_hug_edges(50000, 130000, 2.4, 5)), (30000, K.gen_hug_edges(30000, 90000, 2.2, 8))]
    want = []
    for nv, uv in graphs:
        rowptr, col = O.simplify(nv, uv)
        want.append((O.coreness(rowptr, col), O.support(rowptr, col)[0], O.trussness(rowptr, col)))
    errors = []

    def work(i):
        try:
            nv, uv = graphs[i]
            with K.KombAccel() as a:
                a.from_edges(nv, uv)
                for _ in range(6):
                    deg, core = a.run_core()
                    eu, ev, tr, sup = a.run_truss(with_support=True)
                    if not (np.array_equal(core, want[i][0]) and np.array_equal(sup, want[i][1]) and np.array_equal(tr, want[i][2])):
                        errors.append(f"thread {i}: result differs from the oracle")
        except Exception as exc:  # noqa: BLE001
            errors.append(f"thread {i}: {exc!r}")

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errors, errors


def test_full_size_c2_properties(K, O):
    """BASELINE config C2 (|V|=1M, |E|~10M): k-core, supports and EVERY trussness value against the oracle,
    plus the size-independent properties."""
    nv = 1_000_000
    uv = K.gen_hug_edges(nv, 2_450_000, 2.6, 42)
    with K.KombAccel() as a:
        a.from_edges(nv, uv)
        rowptr, col = a.get_csr()
        deg, core = a.run_core()
        assert np.array_equal(deg, np.diff(rowptr).astype(np.int32))
        assert np.array_equal(core, O.coreness(rowptr, col))
        eu, ev, tr, sup = a.run_truss(with_support=True)
        st = a.stats()
        assert len(eu) == a.ne and np.all(eu < ev)
        key = eu.astype(np.int64) * nv + ev
        assert np.all(np.diff(key) > 0)                                  # canonical order, no duplicates
        assert int(sup.sum()) == 3 * st["triangles"]
        assert np.all(tr >= 2) and np.all(tr <= sup + 2)
        assert np.all(tr <= np.minimum(core[eu], core[ev]) + 1)
        assert np.all(tr[sup == 0] == 2)                                 # triangle-free edge -> 2
        assert st["max_trussness"] == tr.max() and st["max_coreness"] == core.max()
        osup, otri = O.support(rowptr, col)
        assert np.array_equal(sup, osup) and st["triangles"] == otri
        assert np.array_equal(tr, _oracle_trussness_fast(O, rowptr, col))      # full-value parity, every run
        # CoreA at this size (a9 + a10, src/CoreA.h:109-187): both rank vectors and every score against the oracle
        rd, rk = a.fractional_ranks(deg, core)
        assert np.array_equal(rd, O.fractional_rank_fast(deg.astype(np.int64)))
        assert np.array_equal(rk, O.fractional_rank_fast(core.astype(np.int64) * nv + deg))
        assert np.array_equal(a.get_anomaly_score(deg, core), O.corea_scores(deg, core))
        assert a.stats()["ms_corea"] > 0
        # the top truss class is closed: every edge of the max-truss subgraph has
        # >= tmax-2 triangles inside it (checked with the oracle on that small subgraph)
        top = tr == tr.max()
        vs = np.unique(np.concatenate([eu[top], ev[top]]))
        remap = -np.ones(nv, np.int64); remap[vs] = np.arange(len(vs))
        sub = np.stack([remap[eu[top]], remap[ev[top]]], axis=1)
        s_rowptr, s_col = O.simplify(len(vs), sub)
        s_sup, _ = O.support(s_rowptr, s_col)
        assert s_sup.min() >= tr.max() - 2


def test_index_layouts_agree(K, O, monkeypatch):
    """The two index builds -- over the two enumerations -- must give the same supports and trussness: record stream of the
    wedge enumeration (default: dense own-role blocks + sorted records; also with a dense region that runs out, with no dense
    region at all, and with a stream that runs out and falls back), exact two-pass over the probe enumeration of rounds 1-3."""
    env = ("KOMB_TWO_PASS", "KOMB_INDEX", "KOMB_OWN_DENSE_CAP", "KOMB_NO_OWN_DENSE", "KOMB_REC_CAP")
    uv = K.gen_hug_edges(40000, 110000, 2.3, 21)
    with K.KombAccel() as a:
        a.from_edges(40000, uv)

        def run(**kv):
            for k in env:
                monkeypatch.delenv(k, raising=False)
            for k, v in kv.items():
                monkeypatch.setenv(k, v)
            r = a.run_truss(with_support=True)
            st = a.stats()
            for k in env:
                monkeypatch.delenv(k, raising=False)
            return r, st

        r1, st = run()
        assert st["index_layout"] == 0 and st["ms_tri_count"] == 0 and st["ms_sort"] > 0 and st["tri_records"] >= st["triangles"]
        rowptr, col = a.get_csr()
        assert np.array_equal(r1[2], O.trussness(rowptr, col))
        assert np.array_equal(r1[3], O.support(rowptr, col)[0])
        variants = [
            (dict(KOMB_TWO_PASS="1"), 2), (dict(KOMB_INDEX="two_pass"), 2),
            # stream: a dense region that runs out half way (the tasks that find no room send their own-role entries to the
            # stream one by one), no dense region at all (three records per triangle), a stream that runs out (-> two-pass)
            (dict(KOMB_OWN_DENSE_CAP="70000"), 0), (dict(KOMB_NO_OWN_DENSE="1"), 0), (dict(KOMB_REC_CAP="50000"), 2),
        ]
        for kv, layout in variants:
            r, st = run(**kv)
            assert st["index_layout"] == layout, (kv, st["index_layout"])
            if layout == 2:
                assert st["ms_tri_count"] > 0
            if kv == dict(KOMB_NO_OWN_DENSE="1"):
                assert st["tri_records"] >= 3 * st["triangles"]
            for x, y in zip(r1, r):
                assert np.array_equal(x, y), kv


def test_sharded_peel_engine_single_rank(K, O, monkeypatch):
    """The sharded peel of shard_dev.h with ONE rank: the same host loop, kernels and range-filtered problem
    types as with N ranks, the exchange being the identity -- k-core and k-truss against the oracle on graphs with hub rows
    / hub edges (heavy units), cliques (one far level), cascades and the two index layouts.  (N ranks: tests/test_distributed.py.)"""
    rng = np.random.default_rng(17)
    iu = np.triu_indices(90, 1)
    clique = np.stack(iu, axis=1) + 50
    nvc = 4000
    star = np.stack([np.zeros(nvc - 1, np.int64), np.arange(1, nvc)], axis=1)
    path = np.stack([np.arange(1000, 3999), np.arange(1001, 4000)], axis=1)
    cases = [(40000, K.gen_hug_edges(40000, 110000, 2.3, 21)), (30000, K.gen_hug_edges(30000, 80000, 2.1, 9)),
             (nvc, np.concatenate([clique, star, path, rng.integers(0, nvc, (2500, 2))]).astype(np.int64)),
             (500, rng.integers(0, 500, (30000, 2)).astype(np.int64))]
    for i, (nv, uv) in enumerate(cases):
        with K.KombAccel() as a:
            a.from_edges(nv, uv)
            rowptr, col = a.get_csr()
            monkeypatch.setenv("KOMB_SHARD_PEEL", "1")
            ocore, osup, otr = O.coreness(rowptr, col), O.support(rowptr, col)[0], O.trussness(rowptr, col)
            # "none": the sharded sub-rounds to the very end; "local" (the default): the remainder is handed to the local
            # finish under the replicated peel's rule (tiny limits: many offers, some declined)
            for finish, limit in (("none", None), ("local", None), ("local", "300")):
                monkeypatch.setenv("KOMB_FINISH", finish)
                if limit: monkeypatch.setenv("KOMB_LOCAL_LIMIT", limit)
                deg, core = a.run_core()
                st = a.stats()
                assert np.array_equal(core, ocore), (i, finish)
                if finish == "none":                 # (with the local finish a small graph is handed over whole, before any exchange)
                    assert st["shard_exchanges"] > 0 and st["core_subrounds"] > 0 and st["core_local_units"] == 0
                for layout in ("stream", "two_pass"):
                    monkeypatch.setenv("KOMB_INDEX", layout)
                    eu, ev, tr, sup = a.run_truss(with_support=True)
                    st = a.stats()
                    if finish == "none": assert st["shard_exchanges"] > 0 and st["truss_local_units"] == 0
                    assert np.array_equal(sup, osup), (i, finish, layout)
                    assert np.array_equal(tr, otr), (i, finish, layout)
                monkeypatch.delenv("KOMB_INDEX", raising=False)
                monkeypatch.delenv("KOMB_LOCAL_LIMIT", raising=False)
            monkeypatch.delenv("KOMB_FINISH", raising=False)
            monkeypatch.delenv("KOMB_SHARD_PEEL", raising=False)
            a.truss_run()
            assert a.stats()["shard_exchanges"] == 0


def test_internal_renumbering_is_invisible(K, O):
    """The library works on (degree, id)-ranked internal vertex ids (graph_build.hip) and reports everything in the caller's:
    shapes whose degree order is far from their id order -- a hub at id 0, isolated vertices in the middle of the id range,
    a regular ring (every degree equal: the order is the id order), a dense block at the low ids -- through both
    constructors, with random induced subgraphs, must give the oracle's CSR, degrees, coreness, canonical edge order,
    supports and trussness."""
    rng = np.random.default_rng(31)
    ring = np.stack([np.arange(3000), (np.arange(3000) + 1) % 3000], axis=1)
    star = np.stack([np.zeros(500, np.int64), np.arange(1, 501)], axis=1)
    clique = np.array([(i, j) for i in range(40) for j in range(i + 1, 40)], dtype=np.int64)
    mixed = np.concatenate([star, clique + 600, np.stack([np.arange(700, 900), np.arange(701, 901)], axis=1),
                            rng.integers(1000, 1400, (3000, 2))]).astype(np.int64)
    cases = [(40000, K.gen_hug_edges(40000, 110000, 2.3, 21)), (3000, np.concatenate([ring, np.roll(ring, 1, axis=1)]).astype(np.int64)),
             (600, rng.integers(0, 600, (40000, 2)).astype(np.int64)), (1500, mixed), (70000, K.gen_hug_edges(70000, 180000, 2.1, 3))]
    for nv, uv in cases:
        o_rowptr, o_col = O.simplify(nv, uv)
        oeu, oev = O.edge_list(o_rowptr, o_col)
        osup, _ = O.support(o_rowptr, o_col)
        otr = O.trussness(o_rowptr, o_col)
        ocore = O.coreness(o_rowptr, o_col)
        for ctor in ("edges", "csr"):
            with K.KombAccel() as a:
                if ctor == "edges": a.from_edges(nv, uv)
                else: a.from_csr(o_rowptr, o_col)
                rowptr, col = a.get_csr()
                assert np.array_equal(rowptr, o_rowptr) and np.array_equal(col, o_col)
                deg, core = a.run_core()
                assert np.array_equal(deg, O.degree(o_rowptr)) and np.array_equal(core, ocore)
                eu, ev, tr, sup = a.run_truss(with_support=True)
                assert np.array_equal(eu, oeu) and np.array_equal(ev, oev)
                assert np.array_equal(sup, osup) and np.array_equal(tr, otr)
                for frac in (0.5, 0.05):
                    mask = (rng.random(nv) < frac).astype(np.uint8)
                    seu, sev, stra = a.run_truss(mask)
                    weu, wev, wtr = O.trussness_induced(o_rowptr, o_col, mask)
                    assert np.array_equal(seu, weu) and np.array_equal(sev, wev) and np.array_equal(stra, wtr)
                # a whole-graph run after a subgraph run: the resident canonical edge list is untouched
                eu2, ev2, tr2 = a.run_truss()
                assert np.array_equal(eu2, oeu) and np.array_equal(ev2, oev) and np.array_equal(tr2, otr)


def test_wedge_enumeration_paths(K, O, monkeypatch):
    """The triangle enumeration by wedges (truss_wedge.h) against the oracle and against round 3's probe kernel on shapes that
    reach each of its paths: K_700 (rows of up to 699 slots: unstaged, cut into PARTS of ~16 k wedges, all entries through the
    record stream), a clique chain of K_150s (single-row tasks of 32..149 slots whose records outgrow the LDS buffer: the
    wave-private scratch, switched off for comparison), a hub-heavy generator graph (rows around the 256-slot staging limit next
    to thousands of light vertices per task), with the induced-subgraph variant (lines and tasks built per run)."""
    rng = np.random.default_rng(5)
    iu = np.triu_indices(700, 1)
    k700 = np.stack(iu, axis=1).astype(np.int64)
    iu = np.triu_indices(150, 1)
    chain = np.concatenate([np.stack(iu, axis=1) + 140 * i for i in range(12)]).astype(np.int64)     # neighbours share 10 vertices
    cases = [("K_700", 700, k700), ("chain of K_150", 140 * 12 + 20, chain),
             ("hug 30k alpha 2.05", 30000, np.asarray(K.gen_hug_edges(30000, 100000, 2.05, 11)).reshape(-1, 2))]
    for name, nv, uv in cases:
        o_rowptr, o_col = O.simplify(nv, uv)
        osup, otri = O.support(o_rowptr, o_col)
        otr = O.trussness(o_rowptr, o_col)
        with K.KombAccel() as a:
            a.from_edges(nv, uv)
            for env in ({}, {"KOMB_NO_REC_SCRATCH": "1"}, {"KOMB_INDEX": "two_pass"}, {"KOMB_NO_OWN_DENSE": "1"}):
                for k, v in env.items(): monkeypatch.setenv(k, v)
                eu, ev, tr, sup = a.run_truss(with_support=True)
                st = a.stats()
                for k in env: monkeypatch.delenv(k, raising=False)
                assert st["triangles"] == otri and np.array_equal(sup, osup) and np.array_equal(tr, otr), (name, env)
            mask = (rng.random(nv) < 0.6).astype(np.uint8)
            seu, sev, stra = a.run_truss(mask)
            weu, wev, wtr = O.trussness_induced(o_rowptr, o_col, mask)
            assert np.array_equal(seu, weu) and np.array_equal(sev, wev) and np.array_equal(stra, wtr), name


def test_edge_state_bytes_and_retire_steps(K, O, monkeypatch):
    """The truss peel gathers one-byte edge states (peel_dev.h: state_of_round) whose sub-round codes are taken modulo 253 and
    kept unambiguous by the engine's RETIRE step.  General engine only (KOMB_FINISH=none: every sub-round runs in it, in-kernel
    chains included) on graphs with hundreds of sub-rounds -- more than the code window -- with the default period, and with a
    RETIRE step after every 1 / 2 / 7 sub-rounds; the local finish on top for the hand-over from byte states."""
    iu = np.triu_indices(90, 1)
    cases = [("hug 100k alpha 2.1", 100000, np.asarray(K.gen_hug_edges(100000, 300000, 2.1, 3)).reshape(-1, 2)),
             # a path of overlapping cliques of growing size: one level after the other, each a cascade of its own
             ("clique ladder", 4000, np.concatenate([np.stack(np.triu_indices(k, 1), axis=1) + 37 * i for i, k in enumerate(range(3, 100))]).astype(np.int64)),
             ("K_90 + tail", 600, np.concatenate([np.stack(iu, axis=1), np.stack([np.arange(89, 599), np.arange(90, 600)], axis=1)]).astype(np.int64))]
    most = 0
    for name, nv, uv in cases:
        uv = np.ascontiguousarray(uv, dtype=np.int64)
        o_rowptr, o_col = O.simplify(nv, uv)
        otr = O.trussness(o_rowptr, o_col)
        with K.KombAccel() as a:
            a.from_edges(nv, uv)
            rounds = 0
            for env in ({"KOMB_FINISH": "none"}, {"KOMB_FINISH": "none", "KOMB_RETIRE_EVERY": "1"}, {"KOMB_FINISH": "none", "KOMB_RETIRE_EVERY": "2"},
                        {"KOMB_FINISH": "none", "KOMB_RETIRE_EVERY": "7"}, {"KOMB_FINISH": "local", "KOMB_RETIRE_EVERY": "3", "KOMB_LOCAL_LIMIT": "200"},
                        {"KOMB_FINISH": "lds", "KOMB_RETIRE_EVERY": "5"}):
                for k, v in env.items(): monkeypatch.setenv(k, v)
                eu, ev, tr = a.run_truss()
                st = a.stats()
                for k in env: monkeypatch.delenv(k, raising=False)
                assert np.array_equal(tr, otr), (name, env)
                rounds = max(rounds, st["truss_subrounds"])
            print(name, "sub-rounds", rounds)
            most = max(most, rounds)
    assert most > 2 * 253, most                         # the codes did wrap


def test_truss_preparation_lifecycle(K, O):
    """The k-truss side of a graph (truss_prep.hip) is made by the first k-truss call, inside that call (ms_prepare), kept for
    the next ones, dropped by komb_truss_unprepare together with the last result, and never made by the graph build, k-core or
    CoreA; an induced subgraph gets a temporary one every time and leaves the whole graph's alone."""
    nv = 50000
    uv = np.asarray(K.gen_hug_edges(nv, 150000, 2.3, 31)).reshape(-1, 2)
    o_rowptr, o_col = O.simplify(nv, uv)
    otr = O.trussness(o_rowptr, o_col)
    for build in ("edges", "csr"):
        with K.KombAccel() as a:
            if build == "edges": a.from_edges(nv, uv)
            else: a.from_csr(o_rowptr, o_col)
            st = a.stats()
            assert st["ms_build_relabel"] == 0 and st["ms_prepare"] == 0 and st["truss_prepared"] == 0
            deg, core = a.run_core()
            a.get_anomaly_score(deg, core)
            assert a.stats()["ms_prepare"] == 0                               # k-core and CoreA never touch it
            eu, ev, tr = a.run_truss()
            st = a.stats()
            assert st["truss_prepared"] == 1 and st["ms_prepare"] > 0 and np.array_equal(tr, otr)
            a.truss_run()
            st = a.stats()
            assert st["truss_prepared"] == 0 and st["ms_prepare"] == 0        # found, not rebuilt
            a.truss_unprepare()
            with pytest.raises(K.KombError) as e:
                a.truss_fetch()                                               # the result went with it
            assert e.value.code == K._lib.KOMB_ERR_STATE
            a.truss_prepare()
            assert a.stats()["ms_prepare"] > 0
            a.truss_prepare()                                                 # a no-op now
            eu2, ev2, tr2 = a.run_truss()
            assert a.stats()["truss_prepared"] == 0 and np.array_equal(tr2, otr) and np.array_equal(eu2, eu) and np.array_equal(ev2, ev)
            mask = (core >= np.sort(core)[-nv // 20]).astype(np.uint8)
            for _ in range(2):
                seu, sev, stra = a.run_truss(mask)
                st = a.stats()
                assert st["truss_prepared"] == 1 and st["ms_prepare"] > 0      # the subgraph's own, every time
                weu, wev, wtr = O.trussness_induced(o_rowptr, o_col, mask)
                assert np.array_equal(seu, weu) and np.array_equal(sev, wev) and np.array_equal(stra, wtr)
            eu3, ev3, tr3 = a.run_truss()
            assert a.stats()["truss_prepared"] == 0 and np.array_equal(tr3, otr) and np.array_equal(eu3, eu)
            a.graph_moments()
            st = a.stats()
            assert st["sum_deg_sq"] == int((deg.astype(np.int64) ** 2).sum()) and st["max_degree"] == int(deg.max())
            a.set_option("INDEX", "no_such_layout")
            with pytest.raises(K.KombError) as e:
                a.truss_run()
            assert e.value.code == K._lib.KOMB_ERR_ARG
            a.set_option("INDEX", None)
            a.truss_run()


def test_results_made_on_demand(K, O, monkeypatch):
    """What the timed call does not make: the canonical edge LIST (igraph_edge's answer, reference src/graph.cpp:529-532) comes
    with the first komb_truss_fetch that asks for endpoints and stays with the graph; the supports in canonical order come with
    the first komb_truss_fetch_support after a run (from the slice table the result keeps); the oriented slots' sources are made
    when the LDS tail (FINISH=lds) asks.  Any fetch order, any of the outputs left out, runs in between, a vmask run in
    between (its result carries its own list), a new graph afterwards."""
    nv = 30000
    uv = np.asarray(K.gen_hug_edges(nv, 90000, 2.3, 17)).reshape(-1, 2)
    o_rowptr, o_col = O.simplify(nv, uv)
    rows = np.repeat(np.arange(nv), np.diff(o_rowptr))
    up = o_col > rows
    ceu, cev = rows[up].astype(np.int32), o_col[up].astype(np.int32)
    otr, osup = O.trussness(o_rowptr, o_col), O.support(o_rowptr, o_col)[0]
    m = len(ceu)
    with K.KombAccel() as a:
        a.from_edges(nv, uv)
        a.truss_run()
        tr = np.full(m, -1, np.int32)
        a.truss_fetch_into(None, None, tr)                                  # trussness alone: no edge list yet
        assert np.array_equal(tr, otr)
        ev = np.full(m, -1, np.int32)
        a.truss_fetch_into(None, ev, None)                                  # one endpoint array alone
        assert np.array_equal(ev, cev)
        eu2, ev2, tr2, sup2 = a.truss_fetch(with_support=True)
        assert np.array_equal(eu2, ceu) and np.array_equal(ev2, cev) and np.array_equal(tr2, otr) and np.array_equal(sup2, osup)
        a.truss_run()                                                       # a new result: the list stays, the supports are made again
        eu3, ev3, tr3, sup3 = a.truss_fetch(with_support=True)
        assert np.array_equal(eu3, ceu) and np.array_equal(ev3, cev) and np.array_equal(tr3, otr) and np.array_equal(sup3, osup)
        mask = np.zeros(nv, np.uint8); mask[::2] = 1
        seu, sev, stra, ssup = a.run_truss(mask, with_support=True)
        weu, wev, wtr = O.trussness_induced(o_rowptr, o_col, mask)
        assert np.array_equal(seu, weu) and np.array_equal(sev, wev) and np.array_equal(stra, wtr) and np.all(ssup + 2 >= stra)
        eu4, ev4, tr4 = a.run_truss()
        assert np.array_equal(eu4, ceu) and np.array_equal(ev4, cev) and np.array_equal(tr4, otr)
        a.truss_unprepare()                                                 # the preparation goes, the list is the graph's
        a.set_option("FINISH", "lds"); a.set_option("TAIL", "3000")          # the LDS tail reads the slots' sources: made for it
        eu5, ev5, tr5, sup5 = a.run_truss(with_support=True)
        assert np.array_equal(eu5, ceu) and np.array_equal(tr5, otr) and np.array_equal(sup5, osup) and a.stats()["truss_tail_runs"] >= 1
        seu, sev, stra = a.run_truss(mask)                                  # ... and for an induced subgraph's temporary preparation
        assert np.array_equal(seu, weu) and np.array_equal(stra, wtr)
        a.set_option("FINISH", None); a.set_option("TAIL", None)
        # a new graph in the same context: nothing of the old one's list survives
        uvb = np.stack([np.arange(0, 99), np.arange(1, 100)], axis=1).astype(np.int64)
        a.from_edges(100, uvb)
        eu6, ev6, tr6, sup6 = a.run_truss(with_support=True)
        assert np.array_equal(eu6, np.arange(99)) and np.array_equal(ev6, np.arange(1, 100)) and np.all(tr6 == 2) and np.all(sup6 == 0)


def test_preparation_long_rows(K, O, monkeypatch):
    """The preparation's workgroup paths (truss_prep.hip): oriented rows beyond one wavefront's 1024-entry sort (K_1500: rows of
    up to 1499 entries, ranked out of LDS by a workgroup -- and, with the staging switched off, out of global memory), symmetric
    rows beyond 2048 slots (a hub: walked in chunks by the whole grid, nearly all of its edges handed to other rows through their
    back cursors; and long rows that keep their edges), and the same rows inside an induced subgraph."""
    n = 1500
    uv = np.stack(np.triu_indices(n, 1), axis=1).astype(np.int64)
    for stage in (None, "100"):
        if stage: monkeypatch.setenv("KOMB_PREP_ROW_STAGE", stage)
        with K.KombAccel() as a:
            a.from_edges(n, uv)
            eu, ev, tr, sup = a.run_truss(with_support=True)
            assert np.all(sup == n - 2) and np.all(tr == n) and a.stats()["triangles"] == n * (n - 1) * (n - 2) // 6
            assert np.array_equal(eu, uv[:, 0]) and np.array_equal(ev, uv[:, 1])            # canonical order = the generator's
        monkeypatch.delenv("KOMB_PREP_ROW_STAGE", raising=False)
    # a hub with 6000 neighbours, 3000 of which form a clique chain among themselves; mask = hub + the chain
    rng = np.random.default_rng(8)
    hubn = 6000
    star = np.stack([np.zeros(hubn, np.int64), np.arange(1, hubn + 1)], axis=1)
    chain = np.concatenate([np.stack(np.triu_indices(30, 1), axis=1) + 1 + 25 * i for i in range(110)])
    noise = rng.integers(1, hubn + 1, (20000, 2))
    uv = np.concatenate([star, chain, noise]).astype(np.int64)
    nv = hubn + 1
    o_rowptr, o_col = O.simplify(nv, uv)
    with K.KombAccel() as a:
        a.from_edges(nv, uv)
        eu, ev, tr, sup = a.run_truss(with_support=True)
        assert np.array_equal(sup, O.support(o_rowptr, o_col)[0]) and np.array_equal(tr, O.trussness(o_rowptr, o_col))
        mask = np.zeros(nv, np.uint8); mask[:3000] = 1
        seu, sev, stra = a.run_truss(mask)
        weu, wev, wtr = O.trussness_induced(o_rowptr, o_col, mask)
        assert np.array_equal(seu, weu) and np.array_equal(sev, wev) and np.array_equal(stra, wtr)
    # long rows that KEEP their edges: a cycle of 2100 vertices (ids first), each of them joined to each of 2060 others.  Both
    # sides are beyond 2048 slots and the cycle's vertices have the lower (degree, id) rank, so the 2060 upper slots of each of
    # them stay in its own row -- placed through the row's counter by many wavefronts at once.  Known answer: a cycle edge is in
    # 2060 triangles, every other edge in 2 (its two cycle neighbours), and the whole graph is its own 4-truss.
    nb, na = 2100, 2060
    bip = np.stack(np.meshgrid(np.arange(nb), nb + np.arange(na), indexing="ij"), axis=-1).reshape(-1, 2)
    cyc = np.stack([np.arange(nb), (np.arange(nb) + 1) % nb], axis=1)
    uv = np.concatenate([bip, cyc]).astype(np.int64)
    with K.KombAccel() as a:
        a.from_edges(nb + na, uv)
        eu, ev, tr, sup = a.run_truss(with_support=True)
        assert len(eu) == nb * na + nb and a.stats()["triangles"] == nb * na
        assert np.all(eu < ev) and np.all(np.diff(eu * (nb + na) + ev.astype(np.int64)) > 0)
        assert np.all(tr == 4) and np.array_equal(sup, np.where(ev < nb, na, 2))


def test_retire_step_due_at_a_refused_hand_over(K, O, monkeypatch):
    """Regression test for commit 9b51a4a (round 4's last engine fix): a RETIRE step that falls due exactly when the remainder
    is offered to the local finish must stay pending when the finish REFUSES the offer (KOMB_LOCAL_ITEMS=0 refuses every one)
    and the general engine goes on.  With a RETIRE period of 1 or 3 the offer of these graphs coincides with a due RETIRE; the
    engine records the longest run of sub-rounds without a RETIRE (PeelCtrl::max_retire_gap) and the host fails the run when
    it exceeds the period -- which it does under the old rule (`if (retire && !done)`; build with -DKOMB_TEST_OLD_RETIRE_RULE:
    tests/manual/retire_rule_negative_control.sh).  Results against the oracle on the >= 506-sub-round graphs of the test above."""
    iu = np.triu_indices(90, 1)
    cases = [("hug 100k alpha 2.1", 100000, np.asarray(K.gen_hug_edges(100000, 300000, 2.1, 3)).reshape(-1, 2)),
             ("clique ladder", 4000, np.concatenate([np.stack(np.triu_indices(k, 1), axis=1) + 37 * i for i, k in enumerate(range(3, 100))]).astype(np.int64)),
             ("K_90 + tail", 600, np.concatenate([np.stack(iu, axis=1), np.stack([np.arange(89, 599), np.arange(90, 600)], axis=1)]).astype(np.int64))]
    most = 0
    for name, nv, uv in cases:
        uv = np.ascontiguousarray(uv, dtype=np.int64)
        o_rowptr, o_col = O.simplify(nv, uv)
        otr = O.trussness(o_rowptr, o_col)
        with K.KombAccel() as a:
            a.from_edges(nv, uv)
            for every in ("1", "3"):
                for limit in (None, "200", "3000"):           # hand-over thresholds: the default fraction, and small remainders late in the peel
                    env = {"KOMB_FINISH": "local", "KOMB_LOCAL_ITEMS": "0", "KOMB_RETIRE_EVERY": every}
                    if limit: env["KOMB_LOCAL_LIMIT"] = limit
                    for k, v in env.items(): monkeypatch.setenv(k, v)
                    eu, ev, tr = a.run_truss()
                    st = a.stats()
                    for k in env: monkeypatch.delenv(k, raising=False)
                    assert st["truss_local_units"] == 0, (name, env)             # every offer was refused: the engine peeled to the end
                    assert np.array_equal(tr, otr), (name, env)
                    most = max(most, st["truss_subrounds"])
    assert most > 2 * 253, most


def test_result_slices(K, O):
    """komb_truss_run_slice (what bench.py --gpus N runs on every rank): the whole path, the results of the rank's slice of
    the canonical edges, zeros elsewhere -- the slices of 1, 2, 3 and 7 ranks add up to the whole result (trussness and
    support), the edge list is complete on every rank; with a vertex mask every rank holds the whole (sub)result."""
    nv = 30000
    uv = np.asarray(K.gen_hug_edges(nv, 90000, 2.4, 9)).reshape(-1, 2)
    o_rowptr, o_col = O.simplify(nv, uv)
    osup, _ = O.support(o_rowptr, o_col)
    otr = O.trussness(o_rowptr, o_col)
    ne = len(otr)
    with K.KombAccel() as a:
        a.from_edges(nv, uv)
        eu0, ev0, _ = a.run_truss()
        for world in (1, 2, 3, 7):
            tot_tr = np.zeros(ne, np.int64); tot_sup = np.zeros(ne, np.int64)
            for rank in range(world):
                a.truss_run_slice(rank, world)
                eu, ev, tr, sup = a.truss_fetch(with_support=True)
                lo, hi = ne * rank // world, ne * (rank + 1) // world
                assert np.array_equal(eu, eu0) and np.array_equal(ev, ev0)
                assert np.array_equal(tr[lo:hi], otr[lo:hi]) and np.array_equal(sup[lo:hi], osup[lo:hi]), (world, rank)
                assert not tr[:lo].any() and not tr[hi:].any() and not sup[:lo].any() and not sup[hi:].any(), (world, rank)
                tot_tr += tr; tot_sup += sup
            assert np.array_equal(tot_tr, otr) and np.array_equal(tot_sup, osup), world
        mask = (np.random.default_rng(4).random(nv) < 0.7).astype(np.uint8)
        weu, wev, wtr = O.trussness_induced(o_rowptr, o_col, mask)
        a.truss_run_slice(1, 3, mask)
        seu, sev, stra = a.truss_fetch()
        assert np.array_equal(seu, weu) and np.array_equal(sev, wev) and np.array_equal(stra, wtr)
        for bad in ((3, 3), (-1, 2), (0, 0)):
            with pytest.raises(Exception):
                a.truss_run_slice(*bad)
        eu, ev, tr = a.run_truss()                       # a plain run afterwards is whole again
        assert np.array_equal(tr, otr)


def test_dense_uniform_stream_against_two_pass(K, monkeypatch):
    """A dense uniform random graph (30 000 vertices, ~12.5 M edges of degree ~830: oriented rows of ~400 slots, all beyond
    the enumeration's staging limit, most beyond the preparation's per-wavefront sort).  The record-stream build over the
    wedge enumeration against the exact two-pass build over the probe enumeration, and the size-independent properties;
    the oracle would need minutes here."""
    rng = np.random.default_rng(77)
    nv = 30000
    uv = rng.integers(0, nv, (12_600_000, 2)).astype(np.int64)
    for k in ("KOMB_TWO_PASS", "KOMB_NO_OWN_DENSE", "KOMB_INDEX"):
        monkeypatch.delenv(k, raising=False)
    with K.KombAccel() as a:
        a.from_edges(nv, uv)
        del uv
        deg, core = a.run_core()
        eu, ev, tr, sup = a.run_truss(with_support=True)                    # the default build (record stream)
        st = a.stats()
        assert st["index_layout"] == 0 and st["ms_tri_count"] == 0 and st["truss_prepared"] == 1 and st["ms_prepare"] > 0
        assert sup.sum(dtype=np.int64) == 3 * st["triangles"]
        assert np.all(tr >= 2) and np.all(tr <= sup + 2)
        assert np.all(np.minimum(core[eu], core[ev]) >= tr - 1)
        monkeypatch.setenv("KOMB_TWO_PASS", "1")
        r2 = a.run_truss(with_support=True)
        monkeypatch.delenv("KOMB_TWO_PASS", raising=False)
        st2 = a.stats()
        assert st2["ms_tri_count"] > 0 and st2["truss_prepared"] == 0 and st2["ms_prepare"] == 0     # the preparation stayed with the graph
        for x, y in zip((eu, ev, tr, sup), r2):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("layout", ["stream", "two_pass"])
def test_cliques_and_hubs(K, O, monkeypatch, layout):
    """Complete graphs: long oriented rows (the LDS staging falls back to global search), every
    edge a heavy unit (slices of n-2 > 64 items).  Star + clique: a hub row split into chunks."""
    monkeypatch.delenv("KOMB_TWO_PASS", raising=False)
    monkeypatch.setenv("KOMB_INDEX", layout)
    for n in (70, 200, 320):
        iu = np.triu_indices(n, 1)
        uv = np.stack(iu, axis=1).astype(np.int64)
        with K.KombAccel() as a:
            a.from_edges(n, uv)
            deg, core = a.run_core()
            assert np.all(deg == n - 1) and np.all(core == n - 1)
            if layout == "stream" and n == 320:
                # K_320 has 53 triangles per edge: the record stream's first reservation (two per edge) runs out, the enumeration runs
                # once more with what it asked for, and the build stays a stream build (no two-pass fallback)
                a.truss_run()
                st = a.stats()
                assert st["stream_retries"] == 1 and st["index_layout"] == 0, st
            eu, ev, tr, sup = a.run_truss(with_support=True)
            assert np.all(sup == n - 2) and np.all(tr == n)
            assert a.stats()["triangles"] == n * (n - 1) * (n - 2) // 6
    # hub with 150k leaves, 2000 of them also forming a ring with chords through the hub
    hub, leaves = 0, 150000
    star = np.stack([np.zeros(leaves, np.int64), np.arange(1, leaves + 1)], axis=1)
    ring = np.stack([np.arange(1, 2000), np.arange(2, 2001)], axis=1)
    uv = np.concatenate([star, ring]).astype(np.int64)
    o_rowptr, o_col = O.simplify(leaves + 1, uv)
    with K.KombAccel() as a:
        a.from_edges(leaves + 1, uv)
        deg, core = a.run_core()
        assert deg[hub] == leaves and np.array_equal(core, O.coreness(o_rowptr, o_col))
        eu, ev, tr, sup = a.run_truss(with_support=True)
        osup, _ = O.support(o_rowptr, o_col)
        assert np.array_equal(sup, osup)
        assert np.array_equal(tr, O.trussness(o_rowptr, o_col))


def test_many_tiny_graphs_vs_bruteforce(K):
    """Definitional checkers on 150 random graphs with 1..14 vertices (loops, duplicates, isolated vertices)."""
    import bruteforce as bf
    rng = np.random.default_rng(2024)
    with K.KombAccel() as a:
        for _ in range(150):
            nv = int(rng.integers(1, 15))
            ne = int(rng.integers(0, 4 * nv + 1))
            uv = rng.integers(0, nv, (ne, 2)).astype(np.int64)
            a.from_edges(nv, uv)
            adj = bf.simplify(nv, uv.tolist())
            edges = bf.edges_of(adj)
            deg, core = a.run_core()
            assert deg.tolist() == [len(x) for x in adj]
            assert core.tolist() == bf.coreness(adj)
            eu, ev, tr, sup = a.run_truss(with_support=True)
            assert list(zip(eu.tolist(), ev.tolist())) == edges
            want_sup, want_tr = bf.support(adj), bf.trussness(adj)
            assert sup.tolist() == [want_sup[e] for e in edges]
            assert tr.tolist() == [want_tr[e] for e in edges]


def test_incidence_limit_is_refused(K):
    """K_2100 has 1.54e9 triangles = 4.6e9 incidence entries: beyond the 32-bit slice offsets.
    The library must refuse with KOMB_ERR_LIMIT (never wrap); k-core still works."""
    n = 2100
    uv = np.stack(np.triu_indices(n, 1), axis=1).astype(np.int64)
    with K.KombAccel() as a:
        a.from_edges(n, uv)
        deg, core = a.run_core()
        assert np.all(core == n - 1)
        with pytest.raises(K.KombError) as e:
            a.truss_run()
        assert e.value.code == K._lib.KOMB_ERR_LIMIT and "triangles" in str(e.value)
        with pytest.raises(K.KombError):
            a.truss_fetch()                      # no stale result is exposed


@pytest.mark.parametrize("layout", ["stream", "two_pass"])
def test_kernel_threshold_boundaries(K, O, monkeypatch, layout):
    """Slice lengths around kLight=64 and kChunk=128 ("book" graphs: one spine edge with k pages), and
    16-vertex task blocks whose staged oriented rows straddle the 512-slot LDS budget (dense G(n,p))."""
    monkeypatch.delenv("KOMB_TWO_PASS", raising=False)
    monkeypatch.setenv("KOMB_INDEX", layout)

    def check(nv, uv):
        o_rowptr, o_col = O.simplify(nv, uv)
        with K.KombAccel() as a:
            a.from_edges(nv, uv)
            deg, core = a.run_core()
            assert np.array_equal(core, O.coreness(o_rowptr, o_col))
            eu, ev, tr, sup = a.run_truss(with_support=True)
            osup, _ = O.support(o_rowptr, o_col)
            assert np.array_equal(sup, osup)
            assert np.array_equal(tr, O.trussness(o_rowptr, o_col))

    for k in (63, 64, 65, 127, 128, 129, 255, 256, 257, 700):
        pages = np.arange(2, k + 2)
        uv = np.concatenate([[[0, 1]], np.stack([np.zeros(k, int), pages], 1), np.stack([np.ones(k, int), pages], 1)])
        # a second book sharing the spine's endpoint, and a clique on some pages, to keep several levels alive
        extra = np.stack(np.triu_indices(min(k, 12), 1), axis=1) + 2
        check(k + 2, np.concatenate([uv, extra]).astype(np.int64))

    rng = np.random.default_rng(99)
    for n in (48, 64, 72, 80, 96, 128):
        dense = np.stack(np.triu_indices(n, 1), axis=1)
        keep = rng.random(len(dense)) < 0.9
        check(n, dense[keep].astype(np.int64))


def test_full_size_c3_known_answer(K, O):
    """BASELINE config C3 (|V|=10M, |E|=100.1M, T=88.3M), the workload bench.py times.  Coreness against
    the oracle; trussness through properties and through the SHA-256 recorded when EVERY value was compared
    with the oracle's (tests/manual/c3_parity_oneoff.py, profiles/r01_c3_full_parity.log: 335 s of CPU)."""
    import hashlib
    nv = 10_000_000
    uv = K.gen_hug_edges(nv, 24_250_000, 2.6, 42)
    with K.KombAccel() as a:
        a.from_edges(nv, uv)
        del uv
        assert a.ne == 100_120_558
        rowptr, col = a.get_csr()
        deg, core = a.run_core()
        assert np.array_equal(deg, np.diff(rowptr).astype(np.int32))
        assert np.array_equal(core, O.coreness(rowptr, col))
        del rowptr, col
        eu, ev, tr, sup = a.run_truss(with_support=True)
        st = a.stats()
        assert st["triangles"] == 88_336_441 and int(sup.sum(dtype=np.int64)) == 3 * 88_336_441
        assert np.all(eu < ev) and np.all(tr >= 2) and np.all(tr <= sup + 2)
        assert np.all(tr <= np.minimum(core[eu], core[ev]) + 1)
        assert hashlib.sha256(core.tobytes()).hexdigest()[:16] == "120d47bf172d8b8f"
        assert hashlib.sha256(tr.tobytes()).hexdigest()[:16] == "5970a467914854ea"
        # CoreA at this size (10 M keys; coreness * n + degree reaches 7.2e8 here -- below 2^31, so the reference's `int` key,
        # src/CoreA.h:122, is still defined on this graph): ranks and scores against the oracle
        assert int(core.max()) * nv + int(deg.max()) < 2**31
        rd, rk = a.fractional_ranks(deg, core)
        assert np.array_equal(rd, O.fractional_rank_fast(deg.astype(np.int64)))
        assert np.array_equal(rk, O.fractional_rank_fast(core.astype(np.int64) * nv + deg))
        del rd, rk
        assert np.array_equal(a.get_anomaly_score(deg, core), O.corea_scores(deg, core))
        # full-value parity, every run: all 100.1M trussness values against the oracle's OpenMP variant (~45 s on 16
        # threads); without the native build the recorded hash above stays the only full-size check
        if O.native_lib() is not None:
            rowptr, col = a.get_csr()
            import os
            o_tr = O.trussness_native(rowptr, col, min(16, len(os.sched_getaffinity(0))))
            assert np.array_equal(tr, o_tr)
            del rowptr, col, o_tr
        # Independent of the recorded hash: (1) every truss class from 7 up, exactly.  The edges with trussness >= t ARE the
        # t-truss, and every higher truss lies inside it, so the oracle run on that subgraph alone (340k edges, all the levels
        # the peel's finish computes) must reproduce the GPU's values edge for edge.
        sel = tr >= 7
        vs = np.unique(np.concatenate([eu[sel], ev[sel]]))
        remap = -np.ones(nv, np.int64); remap[vs] = np.arange(len(vs))
        s_rowptr, s_col = O.simplify(len(vs), np.stack([remap[eu[sel]], remap[ev[sel]]], axis=1))
        s_eu, s_ev = O.edge_list(s_rowptr, s_col)
        assert np.array_equal(vs[s_eu], eu[sel]) and np.array_equal(vs[s_ev], ev[sel])      # same canonical order
        assert np.array_equal(O.trussness(s_rowptr, s_col), tr[sel])
        assert len(np.unique(tr[sel])) >= 20
        # (2) 2-hop neighbourhoods of 20 random vertices (capped at 30k vertices): in the induced subgraph the edges at the
        # centre keep ALL their triangles, so their supports equal the global ones; every other support and every trussness
        # can only be smaller (monotone under taking subgraphs).
        rowptr, col = a.get_csr()
        rng = np.random.default_rng(77)
        key = eu.astype(np.int64) * nv + ev
        checked = 0
        for c in rng.integers(0, nv, 60):
            n1 = col[rowptr[c]:rowptr[c + 1]]
            if len(n1) == 0 or len(n1) > 300:
                continue
            n2 = np.unique(np.concatenate([col[rowptr[u]:rowptr[u + 1]] for u in n1] + [n1, [c]]))
            if len(n2) > 30000:
                continue
            mask = np.zeros(nv, np.uint8); mask[n2] = 1
            srp, scol, inv = O.induced_subgraph(rowptr, col, mask)
            ssup, _ = O.support(srp, scol)
            str_ = O.trussness(srp, scol)
            su, sv = O.edge_list(srp, scol)
            pos = np.searchsorted(key, inv[su].astype(np.int64) * nv + inv[sv])
            assert np.array_equal(key[pos], inv[su].astype(np.int64) * nv + inv[sv])
            assert np.all(ssup <= sup[pos]) and np.all(str_ <= tr[pos])
            centre = (inv[su] == c) | (inv[sv] == c)
            assert centre.sum() == len(n1) and np.array_equal(ssup[centre], sup[pos][centre])
            checked += 1
            if checked == 20:
                break
        assert checked >= 10


def test_lds_tail_agrees_with_general_engine(K, O, monkeypatch):
    """The single-workgroup LDS tail (truss_tail.h) against the general engine (KOMB_TAIL=0) and the oracle:
    whole small graphs, hand-over in mid-peel at several thresholds, refusal (too many vertices; or a clique-like
    remainder, which the tail's cost model leaves to the general engine).  The spill of a frontier beyond the LDS
    queues is exercised by the full-size C3 run (656 LDS queue entries there)."""
    monkeypatch.setenv("KOMB_FINISH", "lds")
    rng = np.random.default_rng(5)
    cases = []
    for nv, ne in ((60, 900), (400, 14000), (900, 30000), (3000, 90000)):
        cases.append((nv, rng.integers(0, nv, (ne, 2)).astype(np.int64)))
    cases.append((30000, K.gen_hug_edges(30000, 90000, 2.3, 9)))
    cases.append((120000, K.gen_hug_edges(120000, 300000, 2.6, 10)))
    for n in (40, 300, 340):                                    # K_300: 44,850 edges, K_340: 57,630
        cases.append((n, np.stack(np.triu_indices(n, 1), axis=1).astype(np.int64)))
    # many disjoint triangles + one K_20: few edges left at the end but on > 1024 vertices at first
    tri = np.arange(6000).reshape(-1, 3)
    k20 = 6000 + np.stack(np.triu_indices(20, 1), axis=1)
    cases.append((6020, np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [0, 2]], k20]).astype(np.int64)))
    for nv, uv in cases:
        with K.KombAccel() as a:
            a.from_edges(nv, uv)
            rowptr, col = a.get_csr()
            want = O.trussness(rowptr, col)
            monkeypatch.setenv("KOMB_TAIL", "0")
            _, _, tr0 = a.run_truss()
            assert a.stats()["truss_tail_runs"] == 0
            assert np.array_equal(tr0, want)
            ran = 0
            for limit in (None, "65534", "2000", "150"):
                if limit is None:
                    monkeypatch.delenv("KOMB_TAIL", raising=False)
                else:
                    monkeypatch.setenv("KOMB_TAIL", limit)
                _, _, tr = a.run_truss()
                st = a.stats()
                assert np.array_equal(tr, want), (nv, limit)
                assert st["max_trussness"] == want.max()
                ran += st["truss_tail_runs"]
            clique = len(want) == nv * (nv - 1) // 2
            assert ran > 0 or want.max() <= 2 or clique, nv
    monkeypatch.delenv("KOMB_TAIL", raising=False)
    monkeypatch.delenv("KOMB_FINISH", raising=False)


def test_core_lds_tail_agrees_with_general_engine(K, O, monkeypatch):
    """The single-workgroup LDS tail of the k-core peel (core_tail.h) against the general engine
    (KOMB_CORE_TAIL=0) and the oracle: whole small graphs, hand-over in mid-peel at several thresholds."""
    monkeypatch.setenv("KOMB_FINISH", "lds")
    rng = np.random.default_rng(11)
    cases = []
    for nv, ne in ((50, 400), (700, 20000), (1024, 60000), (5000, 100000)):
        cases.append((nv, rng.integers(0, nv, (ne, 2)).astype(np.int64)))
    cases.append((60000, K.gen_hug_edges(60000, 160000, 2.3, 12)))
    cases.append((300000, K.gen_hug_edges(300000, 740000, 2.6, 13)))
    for n in (30, 500):
        cases.append((n, np.stack(np.triu_indices(n, 1), axis=1).astype(np.int64)))
    star = np.stack([np.zeros(3000, np.int64), np.arange(1, 3001)], axis=1)            # a hub row far longer than the tail
    cases.append((3001, np.concatenate([star, np.stack(np.triu_indices(40, 1), axis=1) + 1]).astype(np.int64)))
    for nv, uv in cases:
        with K.KombAccel() as a:
            a.from_edges(nv, uv)
            rowptr, col = a.get_csr()
            want = O.coreness(rowptr, col)
            for limit in ("0", None, "300", "17"):
                if limit is None:
                    monkeypatch.delenv("KOMB_CORE_TAIL", raising=False)
                else:
                    monkeypatch.setenv("KOMB_CORE_TAIL", limit)
                deg, core = a.run_core()
                assert np.array_equal(core, want), (nv, limit)
                assert a.stats()["max_coreness"] == want.max()
    monkeypatch.delenv("KOMB_CORE_TAIL", raising=False)
    monkeypatch.delenv("KOMB_FINISH", raising=False)


def _local_cases(K):
    rng = np.random.default_rng(23)
    cases = []
    for nv, ne in ((60, 900), (400, 14000), (3000, 90000)):                      # dense random: deep cascades
        cases.append(("gnm%d" % nv, nv, rng.integers(0, nv, (ne, 2)).astype(np.int64)))
    cases.append(("hug30k", 30000, K.gen_hug_edges(30000, 90000, 2.3, 9)))      # power-law unitig graphs
    cases.append(("hug300k", 300000, K.gen_hug_edges(300000, 740000, 2.6, 13)))
    cases.append(("hug50k_a21", 50000, K.gen_hug_edges(50000, 165000, 2.1, 5)))  # heavy tail: hubs, many levels
    for n in (40, 300):                                                            # cliques: every unit heavy at n = 300
        cases.append(("K%d" % n, n, np.stack(np.triu_indices(n, 1), axis=1).astype(np.int64)))
    n = 4000                                                                       # triangle strip: one unit per sweep
    path = np.stack([np.arange(n - 1), np.arange(1, n)], axis=1)
    cases.append(("strip", n, np.concatenate([path, np.stack([np.arange(n - 2), np.arange(2, n)], axis=1)]).astype(np.int64)))
    # a hub with 150k leaves (vertex value 150000: the histogram range is refined twice), a ring with chords, a K_40
    star = np.stack([np.zeros(150000, np.int64), np.arange(1, 150001)], axis=1)
    ring = np.stack([np.arange(1, 2000), np.arange(2, 2001)], axis=1)
    cases.append(("hub", 150001, np.concatenate([star, ring, np.stack(np.triu_indices(40, 1), axis=1) + 1]).astype(np.int64)))
    # books: a spine edge with k pages (k triangles on one edge: light below 256, heavy above, two histogram
    # passes from 4096 on), plus a clique on some pages to keep several levels alive
    for k in (255, 256, 257, 700, 5000, 20000):      # 20000: the spine edge is counted in three chunks by three workgroups
        pages = np.arange(2, k + 2)
        uv = np.concatenate([[[0, 1]], np.stack([np.zeros(k, int), pages], 1), np.stack([np.ones(k, int), pages], 1),
                             np.stack(np.triu_indices(12, 1), axis=1) + 2])
        cases.append(("book%d" % k, k + 2, uv.astype(np.int64)))
    return cases


def test_local_finish_agrees_with_general_engine(K, O, monkeypatch):
    """The local finish (local_dev.h: h-index fixed point on the remainder the peel hands over) against the general
    engine alone (KOMB_FINISH=none) and the oracle, for k-core and k-truss: whole graphs (KOMB_LOCAL_LIMIT larger than
    the graph), the default hand-over, and hand-overs in mid-peel at several thresholds."""
    for name, nv, uv in _local_cases(K):
        with K.KombAccel() as a:
            a.from_edges(nv, uv)
            rowptr, col = a.get_csr()
            want_core = O.coreness(rowptr, col)
            want_tr = O.trussness(rowptr, col)
            monkeypatch.setenv("KOMB_FINISH", "none")
            monkeypatch.delenv("KOMB_LOCAL_LIMIT", raising=False)
            assert np.array_equal(a.run_core()[1], want_core), name
            assert np.array_equal(a.run_truss()[2], want_tr), name
            st = a.stats()
            assert st["core_local_units"] == 0 and st["truss_local_units"] == 0
            monkeypatch.setenv("KOMB_FINISH", "local")
            monkeypatch.setenv("KOMB_LOCAL_DENSITY", "0")                 # take every remainder, however dense
            ran_core = ran_truss = 0
            for limit in ("4000000000", None, "20000", "1500", "100"):
                if limit is None:
                    monkeypatch.delenv("KOMB_LOCAL_LIMIT", raising=False)
                else:
                    monkeypatch.setenv("KOMB_LOCAL_LIMIT", limit)
                core = a.run_core()[1]
                st = a.stats()
                assert np.array_equal(core, want_core), (name, limit)
                assert st["max_coreness"] == want_core.max(), (name, limit)
                ran_core += st["core_local_units"]
                if limit == "4000000000":
                    assert st["core_local_units"] == int((np.diff(rowptr) > 0).sum()), name
                    assert st["core_levels"] == len(np.unique(want_core)), name
                tr = a.run_truss()[2]
                st = a.stats()
                assert np.array_equal(tr, want_tr), (name, limit)
                assert st["max_trussness"] == want_tr.max(), (name, limit)
                ran_truss += st["truss_local_units"]
                if limit == "4000000000" and len(want_tr):
                    sup = a.truss_fetch(with_support=True)[3]
                    assert st["truss_local_units"] == int((sup > 0).sum()), name
                    assert st["truss_levels"] == len(np.unique(want_tr)), name
            assert ran_core > 0 and (ran_truss > 0 or want_tr.max() <= 2), name
            # a remainder with more items than the limit is refused: the peel goes on and offers a smaller one later
            for items in ("0", "3000"):
                monkeypatch.setenv("KOMB_LOCAL_ITEMS", items)
                for limit in ("4000000000", "20000"):
                    monkeypatch.setenv("KOMB_LOCAL_LIMIT", limit)
                    assert np.array_equal(a.run_core()[1], want_core), (name, items, limit)
                    assert np.array_equal(a.run_truss()[2], want_tr), (name, items, limit)
                    st = a.stats()
                    assert st["truss_local_items"] <= int(items) and st["core_local_items"] <= int(items), (name, items, limit)
            monkeypatch.delenv("KOMB_LOCAL_ITEMS", raising=False)
            # the hubs' notification kernel (k_local_giant_notify; by default only from 256 hub chunks on) with and without
            # the list of marked units: forced on every graph that has a hub chunk at all
            monkeypatch.setenv("KOMB_LOCAL_DEFER_CHUNKS", "1")
            monkeypatch.setenv("KOMB_LOCAL_LIMIT", "4000000000")
            assert np.array_equal(a.run_core()[1], want_core), (name, "defer")
            assert np.array_equal(a.run_truss()[2], want_tr), (name, "defer")
            monkeypatch.delenv("KOMB_LOCAL_DEFER_CHUNKS", raising=False)
            # default density rule: a remainder with more than 160 triangles per edge stays with the peel for good
            monkeypatch.delenv("KOMB_LOCAL_DENSITY", raising=False)
            monkeypatch.setenv("KOMB_LOCAL_LIMIT", "4000000000")
            assert np.array_equal(a.run_truss()[2], want_tr), name
            st = a.stats()
            assert st["truss_local_units"] == 0 or st["truss_local_items"] <= 160 * st["truss_local_units"], name
    monkeypatch.delenv("KOMB_LOCAL_LIMIT", raising=False)
    monkeypatch.delenv("KOMB_LOCAL_ITEMS", raising=False)
    monkeypatch.delenv("KOMB_LOCAL_DENSITY", raising=False)
    monkeypatch.delenv("KOMB_FINISH", raising=False)


def test_densest_block_vs_reference_heap(K, O, golden):
    """a12 + a13 (dead code in the reference): komb_densest_block -- CombineCoreA::runMerge over the indexed min-heaps, on
    the device -- against the committed answers of the REFERENCE's own HashIndexedMinHeap.h (tests/golden, `merge_*`) and
    against the oracle on larger graphs: removal order, sides, block size and density bit for bit.  Heaps in LDS (<= 4096
    nodes) and in global memory (above); unweighted and weighted with CoreA scores."""
    n = 0
    for g in golden:
        with K.KombAccel() as a:
            a.from_csr(_i64(g["rowptr"]), np.asarray(g["col"], dtype=np.int32))
            for tag, susp in (("", None), ("w_", np.array([float.fromhex(h) for h in g["ref_corea_hex"]]) if g["nv"] else None)):
                if tag and susp is None:
                    continue
                order, side, nb, dens = a.densest_block(susp)
                assert order.tolist() == g["merge_" + tag + "order"] and side.tolist() == g["merge_" + tag + "side"], g["name"]
                assert nb == g["merge_" + tag + "n_block"] and dens == float.fromhex(g["merge_" + tag + "density_hex"]), g["name"]
                n += 1
    assert n >= 30
    rng = np.random.default_rng(41)
    cases = [(4096, rng.integers(0, 4096, (30000, 2)).astype(np.int64)), (4097, rng.integers(0, 4097, (20000, 2)).astype(np.int64)),
             (9000, rng.integers(0, 9000, (40000, 2)).astype(np.int64)), (20000, K.gen_hug_edges(20000, 50000, 2.6, 7))]
    for nv, uv in cases:
        with K.KombAccel() as a:
            a.from_edges(nv, uv)
            rowptr, col = a.get_csr()
            deg, core = a.run_core()
            for susp in (None, a.get_anomaly_score(deg, core)):
                got, want = a.densest_block(susp), O.run_merge(rowptr, col, susp)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2:] == want[2:], nv
