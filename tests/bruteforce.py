"""Definitional (slow, obviously-correct) checkers, pure Python, small graphs only.

coreness(v)  = max k such that v belongs to a subgraph of minimum degree >= k.
trussness(e) = max k such that e belongs to a subgraph in which every edge lies
               in >= k-2 triangles of that subgraph (triangle-free edge -> 2);
               the definition igraph_trussness documents and KOMB relies on
               (src/graph.cpp:508).
"""


def simplify(nv, raw_edges):
    adj = [set() for _ in range(nv)]
    for u, v in raw_edges:
        if u != v:
            adj[u].add(v)
            adj[v].add(u)
    return adj


def edges_of(adj):
    return sorted((u, v) for u in range(len(adj)) for v in adj[u] if u < v)


def coreness(adj):
    n = len(adj)
    alive = [True] * n
    deg = [len(a) for a in adj]
    core = [0] * n
    k = 0
    left = n
    while left:
        changed = True
        while changed:
            changed = False
            for v in range(n):
                if alive[v] and deg[v] <= k:
                    alive[v] = False
                    core[v] = k
                    left -= 1
                    changed = True
                    for u in adj[v]:
                        if alive[u]:
                            deg[u] -= 1
        k += 1
    return core


def support(adj):
    return {(u, v): len(adj[u] & adj[v]) for (u, v) in edges_of(adj)}


def trussness(adj):
    cur = [set(a) for a in adj]
    truss = {}
    k = 3
    remaining = set(edges_of(cur))
    for e in remaining:
        truss[e] = 2
    while remaining:
        # keep only edges with >= k-2 triangles inside the current subgraph
        changed = True
        while changed:
            changed = False
            for (u, v) in list(remaining):
                if len(cur[u] & cur[v]) < k - 2:
                    remaining.discard((u, v))
                    cur[u].discard(v)
                    cur[v].discard(u)
                    changed = True
        for e in remaining:
            truss[e] = k
        k += 1
    return truss
