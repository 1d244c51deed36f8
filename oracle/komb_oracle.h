/*
 * komb_oracle.h -- CPU restatement of KOMB's k-core / k-truss / CoreA hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under komb_amd/ (the product) may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Parity status:
 *   - CoreA half (rows a8-a11 of SURVEY.md section 8): PINNED.  The reference's
 *     own std-only header src/CoreA.h is compiled in place by oracle/Makefile
 *     into oracle/_ref/corea_ref and orc_corea_* is checked against it (and
 *     against the SURVEY App. D known-answer test) in tests/.
 *   - heap + runMerge (rows a12, a13): PINNED since round 3.  src/HashIndexedMinHeap.h is std-only and is
 *     compiled in place into oracle/_ref/merge_ref under a restatement of runMerge's loop (the loop itself needs
 *     <igraph.h>); orc_run_merge is checked against it on random graphs and through tests/golden/.
 *   - igraph half (rows a1-a3, a5-a6): PARITY UNPINNED by the reference.  The
 *     arithmetic lives in igraph >= 0.10 (komb.yml:7, not vendored, absent
 *     from the image) and the reference holds no tests or golden vectors for
 *     it.  The restatement follows igraph's published algorithms
 *     (Batagelj-Zaversnik coreness; triangle-support + bucket peel trussness)
 *     at the reference's call sites, and is cross-checked against definitional
 *     brute-force checkers and networkx fixtures (tests/golden/).  Coreness,
 *     degree and trussness of a simple graph are unique integers, so any
 *     correct algorithm agrees bit for bit.
 */
#ifndef KOMB_ORACLE_H
#define KOMB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* a1  igraph_simplify(multiple=true, loops=true)  -- call site src/graph.cpp:438.
 * uv = n_raw (u,v) pairs as produced by generateGraph (src/graph.cpp:379-389).
 * Writes rowptr[nv+1] and col[2*ne] (rows sorted ascending, symmetric, no
 * loops, no duplicates); col must have room for 2*n_raw entries.
 * Returns ne (undirected edge count, what src/graph.cpp:444 prints) or -1. */
int64_t orc_simplify(int64_t nv, int64_t n_raw, const int64_t *uv,
                     int64_t *rowptr, int32_t *col);

/* a2  igraph_degree(ALL, NO_LOOPS) on the simplified graph -- src/graph.cpp:462. */
void orc_degree(int64_t nv, const int64_t *rowptr, int32_t *degree);

/* a3  igraph_coreness(IGRAPH_ALL) -- src/graph.cpp:463; Batagelj-Zaversnik
 * bin-sort sweep.  Returns max coreness. */
int32_t orc_coreness(int64_t nv, const int64_t *rowptr, const int32_t *col,
                     int32_t *coreness);

/* a5  igraph_induced_subgraph_map -- src/graph.cpp:502.  vmask[v]!=0 selects
 * vertices.  invmap[new]=old (used at src/graph.cpp:531-532).  sub_col needs
 * room for rowptr[nv] entries, sub_rowptr / invmap for nv+1 / nv.
 * Returns number of selected vertices. */
int64_t orc_induced_subgraph(int64_t nv, const int64_t *rowptr, const int32_t *col,
                             const uint8_t *vmask, int64_t *sub_rowptr,
                             int32_t *sub_col, int32_t *invmap);

/* Canonical edge list of a symmetric CSR: edges (u<v) in lexicographic order.
 * eu/ev have room for ne = rowptr[nv]/2 entries.  Returns ne. */
int64_t orc_edge_list(int64_t nv, const int64_t *rowptr, const int32_t *col,
                      int32_t *eu, int32_t *ev);

/* Per-edge triangle count (the "support" igraph_trussness starts from),
 * canonical edge order.  Returns the number of triangles T. */
int64_t orc_support(int64_t nv, const int64_t *rowptr, const int32_t *col,
                    int32_t *support);

/* a6  igraph_trussness -- src/graph.cpp:508: triangle
 * support, then bucket peel; trussness = level + 2, triangle-free edge -> 2.
 * Canonical edge order.  Returns max trussness (2 if ne>0 and no triangles,
 * 0 if ne==0). */
int32_t orc_trussness(int64_t nv, const int64_t *rowptr, const int32_t *col,
                      int32_t *trussness);

/* a10  CoreA::fractionalRank -- src/CoreA.h:142-187.  Faithful restatement:
 * unique values sorted descending, two full passes per unique value (O(U*n)).
 * Unlike the reference it does not free its input.  rank_out[n]. */
void orc_fractional_rank_faithful(const double *scores, int64_t n, double *rank_out);

/* Same ranks by sort + run lengths (O(n log n)); bit-identical to the faithful
 * form because every rank is the exact half-integer (first+last)/2. */
void orc_fractional_rank_fast(const int64_t *keys, int64_t n, double *rank_out);

/* a9  CoreA::getAnomalyScore -- src/CoreA.h:109-140.
 * key_i = coreness_i*n + degree_i; score = |ln rank_deg - ln rank_key|.
 * faithful!=0 : int arithmetic for the key exactly as src/CoreA.h:122 (caller
 * must keep max_coreness*n + max_degree < 2^31) and the O(U*n) ranker.
 * faithful==0 : int64 keys and the fast ranker. */
void orc_corea_scores(const int32_t *degree, const int32_t *coreness, int64_t n,
                      int faithful, double *score);

/* a12 + a13  HashIndexedMinHeap (src/HashIndexedMinHeap.h:10-238) and its only user, the dead
 * CombineCoreA::runMerge (src/CombineCoreA.h:45-219): greedy densest-block peel over a row copy and a column
 * copy of the graph.  susp: per-node suspiciousness or NULL (then priorities are plain degrees).  order[2*nv] /
 * side[2*nv] are filled from the back as the reference fills `order` / `modes`; the first n_block (the return
 * value) entries are the densest block: its rows are the order[i] with side[i] == 0, its columns those with
 * side[i] == 1.  PINNED against the reference's own heap class compiled in place (oracle/_ref/merge_ref). */
int32_t orc_run_merge(int64_t nv, const int64_t *rowptr, const int32_t *col, const double *susp,
                      int32_t *order, int32_t *side, double *max_density);

/* a6 with every host core (OpenMP builds only: oracle/Makefile target `native`): a level-synchronous parallel peel;
 * the same values as orc_trussness.  nthreads <= 0: OpenMP's default. */
int32_t orc_trussness_omp(int64_t nv, const int64_t *rowptr, const int32_t *col,
                          int32_t *truss, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
