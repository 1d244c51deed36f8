"""Generates tests/golden/graphs.json -- committed golden vectors for the path.

Run in the build container (needs networkx and, for the `ref_corea` column,
oracle/_ref/corea_ref and oracle/_ref/merge_ref built from the reference's own src/CoreA.h and src/HashIndexedMinHeap.h):

    python tests/golden/make_golden.py

Columns per graph: raw edge list (with loops / duplicates where the case is
about simplify), simple CSR, degree, coreness (networkx.core_number), canonical
edge list, support (networkx triangles per edge), trussness (iterated
networkx.k_truss, the same k-2 definition as igraph), CoreA score from the
REFERENCE's CoreA.h (exact doubles as hex).  Everything is also cross-checked
here against tests/bruteforce.py before being written.
"""
import json
import os
import random
import sys

import networkx as nx
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bruteforce as bf                      # noqa: E402
from oracle import oracle as O               # noqa: E402


def nx_trussness(G):
    tr = {tuple(sorted(e)): 2 for e in G.edges()}
    k = 3
    while True:
        H = nx.k_truss(G, k)
        if H.number_of_edges() == 0:
            break
        for e in H.edges():
            tr[tuple(sorted(e))] = k
        k += 1
    return tr


def union_of_cliques(nv, ncl, seed):
    rnd = random.Random(seed)
    w = [(i + 1) ** -0.7 for i in range(nv)]
    perm = list(range(nv))
    rnd.shuffle(perm)
    raw = []
    for _ in range(ncl):
        k = 2
        while k < 6 and rnd.random() >= 0.45:
            k += 1
        mem = [perm[i] for i in rnd.choices(range(nv), weights=w, k=k)]
        for i in range(k):
            for j in range(i + 1, k):
                raw.append((mem[i], mem[j]))
    return raw


def cases():
    kat = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3), (0, 4), (4, 5), (4, 6), (4, 7), (4, 8), (4, 9)]
    yield "kat_survey_appD", 11, kat
    yield "empty_5", 5, []
    yield "single_edge", 2, [(0, 1)]
    yield "loops_and_dups", 4, [(0, 1), (1, 0), (0, 1), (2, 2), (1, 2), (2, 1), (3, 3)]
    yield "triangle", 3, [(0, 1), (1, 2), (0, 2)]
    yield "k5", 5, list(nx.complete_graph(5).edges())
    yield "k4_pendants", 8, list(nx.complete_graph(4).edges()) + [(0, 4), (1, 5), (2, 6), (3, 7)]
    yield "barbell_5_3", 13, list(nx.barbell_graph(5, 3).edges())
    yield "wheel_8", 8, list(nx.wheel_graph(8).edges())
    yield "path_6", 6, list(nx.path_graph(6).edges())
    yield "cycle_7", 7, list(nx.cycle_graph(7).edges())
    yield "star_9", 10, list(nx.star_graph(9).edges())
    yield "bipartite_3_4", 7, list(nx.complete_bipartite_graph(3, 4).edges())
    yield "two_k4_sharing_edge", 6, list(nx.complete_graph(4).edges()) + [(2, 4), (2, 5), (3, 4), (3, 5), (4, 5)]
    yield "isolated_mix", 9, [(1, 2), (2, 3), (1, 3), (5, 6)]
    yield "petersen", 10, list(nx.petersen_graph().edges())
    yield "karate", 34, list(nx.karate_club_graph().edges())
    yield "cliques_300", 300, union_of_cliques(300, 700, 1)
    yield "cliques_800", 800, union_of_cliques(800, 2000, 2)


def main():
    out = []
    for name, nv, raw in cases():
        raw = [(int(u), int(v)) for u, v in raw]
        G = nx.Graph()
        G.add_nodes_from(range(nv))
        G.add_edges_from((u, v) for u, v in raw if u != v)
        adj = bf.simplify(nv, raw)
        edges = bf.edges_of(adj)
        assert edges == sorted(tuple(sorted(e)) for e in G.edges())
        rowptr = [0]
        col = []
        for v in range(nv):
            col.extend(sorted(adj[v]))
            rowptr.append(len(col))
        degree = [len(adj[v]) for v in range(nv)]
        core_nx = nx.core_number(G)
        core = [int(core_nx[v]) for v in range(nv)]
        tr_nx = nx_trussness(G)
        sup = [len(adj[u] & adj[v]) for (u, v) in edges]
        truss = [int(tr_nx[e]) for e in edges]
        if nv <= 300:                                   # brute-force cross-check
            assert core == bf.coreness(adj), name
            tb = bf.trussness(adj)
            assert truss == [tb[e] for e in edges], name
        # max-core induced subgraph, as KOMB's runTruss would take it (src/graph.cpp:470-473,502)
        kmax = max(core) if core else 0
        mask = [1 if c == kmax else 0 for c in core]
        H = G.subgraph([v for v in range(nv) if mask[v]]).copy()
        tr_sub = nx_trussness(H)
        sub_edges = sorted(tuple(sorted(e)) for e in H.edges())
        rec = {
            "name": name, "nv": nv, "raw": raw, "rowptr": rowptr, "col": col,
            "degree": degree, "coreness": core,
            "eu": [e[0] for e in edges], "ev": [e[1] for e in edges],
            "support": sup, "trussness": truss, "triangles": sum(sup) // 3,
            "maxcore_mask": mask,
            "sub_eu": [e[0] for e in sub_edges], "sub_ev": [e[1] for e in sub_edges],
            "sub_trussness": [int(tr_sub[e]) for e in sub_edges],
        }
        if nv:
            ref = O.ref_corea_scores(degree, core)      # the REFERENCE's CoreA.h
            rec["ref_corea_hex"] = [float(x).hex() for x in ref]
        # a12 + a13: the REFERENCE's HashIndexedMinHeap.h under the restated runMerge loop (oracle/_ref/merge_ref),
        # unweighted and weighted with the reference's own CoreA scores
        import numpy as np
        rp, cl = np.asarray(rowptr, np.int64), np.asarray(col, np.int32)
        for tag, susp in (("", None), ("w_", ref if nv else None)):
            if tag and susp is None:
                continue
            order, side, nb, dens = O.ref_run_merge(rp, cl, susp)
            rec["merge_" + tag + "order"] = order.tolist()
            rec["merge_" + tag + "side"] = side.tolist()
            rec["merge_" + tag + "n_block"] = nb
            rec["merge_" + tag + "density_hex"] = float(dens).hex()
        out.append(rec)
        print(f"{name}: nv={nv} ne={len(edges)} T={rec['triangles']} kmax={kmax} tmax={max(truss) if truss else 0}")
    with open(os.path.join(HERE, "graphs.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote graphs.json", os.path.getsize(os.path.join(HERE, "graphs.json")), "bytes")


if __name__ == "__main__":
    main()
