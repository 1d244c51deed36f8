"""CPU tests: the oracle against the golden vectors (networkx / brute force /
the reference's own CoreA.h), against definitional checkers on random graphs,
and its two CoreA rankers against each other."""
import os
import subprocess

import numpy as np
import pytest

import bruteforce as bf


@pytest.fixture(scope="module")
def O(built):
    from oracle import oracle
    return oracle


def _arr(x, dt=np.int32):
    return np.asarray(x, dtype=dt)


def test_golden_simplify_degree_core(O, golden):
    for g in golden:
        rowptr, col = O.simplify(g["nv"], np.asarray(g["raw"], dtype=np.int64).reshape(-1, 2))
        assert rowptr.tolist() == g["rowptr"], g["name"]
        assert col.tolist() == g["col"], g["name"]
        assert O.degree(rowptr).tolist() == g["degree"], g["name"]
        assert O.coreness(rowptr, col).tolist() == g["coreness"], g["name"]


def test_golden_support_truss(O, golden):
    for g in golden:
        rowptr, col = _arr(g["rowptr"], np.int64), _arr(g["col"])
        eu, ev = O.edge_list(rowptr, col)
        assert eu.tolist() == g["eu"] and ev.tolist() == g["ev"], g["name"]
        sup, tri = O.support(rowptr, col)
        assert sup.tolist() == g["support"] and tri == g["triangles"], g["name"]
        assert O.trussness(rowptr, col).tolist() == g["trussness"], g["name"]


def test_golden_maxcore_induced_truss(O, golden):
    """runTruss composes induced subgraph + trussness (src/graph.cpp:470-473,502,508)."""
    for g in golden:
        rowptr, col = _arr(g["rowptr"], np.int64), _arr(g["col"])
        eu, ev, tr = O.trussness_induced(rowptr, col, _arr(g["maxcore_mask"], np.uint8))
        assert eu.tolist() == g["sub_eu"] and ev.tolist() == g["sub_ev"], g["name"]
        assert tr.tolist() == g["sub_trussness"], g["name"]


def test_golden_corea_matches_reference_header(O, golden):
    """ref_corea_hex was produced by the REFERENCE's src/CoreA.h (oracle/_ref/corea_ref)."""
    for g in golden:
        if not g["nv"]:
            continue
        want = np.array([float.fromhex(h) for h in g["ref_corea_hex"]])
        for faithful in (True, False):
            got = O.corea_scores(g["degree"], g["coreness"], faithful=faithful)
            assert np.array_equal(got, want), (g["name"], faithful)


def test_kat_survey_appendix_d(O, golden):
    g = next(x for x in golden if x["name"] == "kat_survey_appD")
    score = O.corea_scores(g["degree"], g["coreness"], faithful=True)
    txt = ["%f" % s for s in score]
    assert txt[0] == "0.693147" and txt[1] == txt[2] == txt[3] == "0.287682"
    assert txt[4] == "1.609438" and set(txt[5:]) == {"0.000000"}
    rk = O.fractional_rank_faithful(np.array(g["coreness"], dtype=np.float64) * 11 + np.array(g["degree"]))
    assert rk.tolist() == [1.0, 3.0, 3.0, 3.0, 5.0, 8.0, 8.0, 8.0, 8.0, 8.0, 11.0]
    assert g["trussness"].count(4) == 6 and g["trussness"].count(2) == 6


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "corea_ref")),
                    reason="oracle/_ref/corea_ref not built (reference absent)")
def test_live_reference_corea_binary(O):
    """Random (degree, coreness) vectors through the reference's CoreA.h, live."""
    rng = np.random.default_rng(5)
    for n in (1, 2, 17, 400):
        core = rng.integers(0, 9, n).astype(np.int32)
        deg = (core + rng.integers(0, 30, n)).astype(np.int32)
        want = O.ref_corea_scores(deg, core)
        assert np.array_equal(O.corea_scores(deg, core, faithful=True), want)
        assert np.array_equal(O.corea_scores(deg, core, faithful=False), want)


def test_reference_tsv_reader(O, tmp_path):
    """a8: kcore.tsv written in the reference's format (src/graph.cpp:467,474) is
    parsed by the reference's readKOMBOutput (src/CoreA.h:24-56) to the same scores."""
    exe = O.ref_corea_path()
    if exe is None:
        pytest.skip("reference binary absent")
    deg = [4, 3, 3, 3, 6, 1, 1, 1, 1, 1, 0]
    core = [3, 3, 3, 3, 1, 1, 1, 1, 1, 1, 0]
    p = tmp_path / "kcore.tsv"
    with open(p, "w") as f:
        f.write("#VID\tName\tCoreness\tDegree\n")
        for i, (d, c) in enumerate(zip(deg, core)):
            f.write("%d\t%s\t%d\t%d\n" % (i, "u%d" % i, c, d))
    out = subprocess.run([exe, "tsv", str(p)], capture_output=True, text=True, check=True).stdout.split()
    assert np.array_equal(np.array([float(x) for x in out]), O.corea_scores(deg, core, faithful=True))


def test_rankers_agree(O):
    rng = np.random.default_rng(11)
    for n in (1, 5, 64, 1000):
        keys = rng.integers(0, max(2, n // 3), n).astype(np.int64)
        a = O.fractional_rank_faithful(keys.astype(np.float64))
        b = O.fractional_rank_fast(keys)
        assert np.array_equal(a, b)
        assert np.all(a * 2 == np.round(a * 2))            # exact half-integers
        assert a.sum() == n * (n + 1) / 2                  # ranks are a permutation of 1..n on average


def _random_raw(rng, nv, ne, loops=True):
    uv = rng.integers(0, nv, (ne, 2))
    if not loops:
        uv = uv[uv[:, 0] != uv[:, 1]]
    return uv.astype(np.int64)


def test_bruteforce_random_graphs(O):
    rng = np.random.default_rng(3)
    for nv, ne in ((1, 0), (2, 3), (8, 20), (12, 50), (30, 120), (40, 400), (60, 300)):
        uv = _random_raw(rng, nv, ne)
        rowptr, col = O.simplify(nv, uv)
        adj = bf.simplify(nv, uv.tolist())
        assert col.tolist() == [w for v in range(nv) for w in sorted(adj[v])]
        assert O.coreness(rowptr, col).tolist() == bf.coreness(adj)
        edges = bf.edges_of(adj)
        sup = bf.support(adj)
        tr = bf.trussness(adj)
        osup, _ = O.support(rowptr, col)
        assert osup.tolist() == [sup[e] for e in edges]
        assert O.trussness(rowptr, col).tolist() == [tr[e] for e in edges]


def test_simplify_rejects_bad_ids(O):
    with pytest.raises(ValueError):
        O.simplify(3, np.array([[0, 3]], dtype=np.int64))


def test_truss_core_bound(O, built):
    """trussness(e) <= min(core(u), core(v)) + 1 on a generated unitig graph."""
    import komb_amd
    nv = 3000
    uv = komb_amd.gen_hug_edges(nv, 8000, 2.6, 9)
    rowptr, col = O.simplify(nv, uv)
    core = O.coreness(rowptr, col)
    eu, ev = O.edge_list(rowptr, col)
    tr = O.trussness(rowptr, col)
    sup, tri = O.support(rowptr, col)
    assert np.all(tr >= 2) and np.all(tr <= sup + 2)
    assert np.all(tr <= np.minimum(core[eu], core[ev]) + 1)
    assert sup.sum() == 3 * tri


def test_all_cores_variant_matches(built):
    """orc_trussness_omp (level-synchronous parallel peel, the all-cores CPU baseline of bench.py) gives the values of
    the sequential restatement: goldens, random dense graphs, a power-law sample; 1, 3 and 8 threads."""
    import json
    import os
    import numpy as np
    import komb_amd
    from oracle import oracle as O
    if O.native_lib() is None:
        pytest.skip("native oracle build unavailable")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "graphs.json")) as f:
        golden = json.load(f)
    cases = [(np.asarray(g["rowptr"], dtype=np.int64), np.asarray(g["col"], dtype=np.int32)) for g in golden if g["nv"]]
    rng = np.random.default_rng(3)
    for nv, ne in ((50, 700), (400, 9000)):
        cases.append(O.simplify(nv, rng.integers(0, nv, (ne, 2)).astype(np.int64)))
    cases.append(O.simplify(20000, komb_amd.gen_hug_edges(20000, 60000, 2.3, 4)))
    for rowptr, col in cases:
        want = O.trussness(rowptr, col)
        for threads in (1, 3, 8):
            assert np.array_equal(O.trussness_native(rowptr, col, threads), want)


def _merge_cases():
    rng = np.random.default_rng(77)
    cases = [(1, np.zeros((0, 2), np.int64)), (5, np.zeros((0, 2), np.int64)), (2, np.array([[0, 1]], np.int64)),
             (6, np.stack(np.triu_indices(6, 1), axis=1).astype(np.int64))]
    for _ in range(25):
        nv = int(rng.integers(2, 250))
        cases.append((nv, rng.integers(0, nv, (int(rng.integers(0, 8 * nv)), 2)).astype(np.int64)))
    return cases


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "merge_ref")),
                    reason="oracle/_ref/merge_ref not built (reference absent)")
def test_run_merge_matches_reference_heap(O):
    """a12 + a13: the oracle's restated indexed min-heap + runMerge loop against the REFERENCE's own HashIndexedMinHeap.h
    compiled in place (oracle/_ref/merge_ref), unweighted and weighted with CoreA scores: removal order, sides, block size
    and density bit for bit -- ties between equal priorities are decided by the heap's layout on both sides."""
    for nv, uv in _merge_cases():
        rowptr, col = O.simplify(nv, uv)
        deg = O.degree(rowptr)
        for susp in (None, O.corea_scores(deg, O.coreness(rowptr, col))):
            a, b = O.run_merge(rowptr, col, susp), O.ref_run_merge(rowptr, col, susp)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3], nv


def test_golden_run_merge(O, golden):
    """The committed fixtures hold the reference heap's answers (tests/golden/make_golden.py, `merge_*` columns)."""
    n = 0
    for g in golden:
        if "merge_order" not in g:
            continue
        rowptr, col = np.asarray(g["rowptr"], np.int64), np.asarray(g["col"], np.int32)
        for tag, susp in (("", None), ("w_", np.array([float.fromhex(h) for h in g["ref_corea_hex"]]) if g["nv"] else None)):
            if tag and susp is None:
                continue
            order, side, nb, dens = O.run_merge(rowptr, col, susp)
            assert order.tolist() == g["merge_" + tag + "order"] and side.tolist() == g["merge_" + tag + "side"], g["name"]
            assert nb == g["merge_" + tag + "n_block"] and dens == float.fromhex(g["merge_" + tag + "density_hex"]), g["name"]
            n += 1
    assert n >= 10
