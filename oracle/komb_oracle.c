/*
 * komb_oracle.c -- CPU restatement (plain C, single thread) of KOMB's
 * k-core / k-truss / CoreA hot path.  TEST INFRASTRUCTURE ONLY: see the
 * header for who may use it and for the parity status ("parity unpinned" for
 * the igraph half, pinned by oracle/_ref/corea_ref for the CoreA half).
 *
 * Every function names the reference call site (file:line under
 * /root/reference) whose observable result it restates.
 */
#include "komb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ utils */

static void radix_sort_u64(uint64_t *a, uint64_t *tmp, int64_t n)
{
    /* LSD radix sort, 16-bit digits, skips digits that are constant. */
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = pass * 16;
        int64_t *cnt = (int64_t *)calloc(65537, sizeof(int64_t));
        for (int64_t i = 0; i < n; ++i) cnt[((a[i] >> shift) & 0xFFFF) + 1]++;
        int trivial = 0;
        for (int d = 0; d < 65536; ++d)
            if (cnt[d + 1] == n) { trivial = 1; break; }
        if (!trivial) {
            for (int d = 0; d < 65536; ++d) cnt[d + 1] += cnt[d];
            for (int64_t i = 0; i < n; ++i) tmp[cnt[(a[i] >> shift) & 0xFFFF]++] = a[i];
            memcpy(a, tmp, (size_t)n * sizeof(uint64_t));
        }
        free(cnt);
    }
}

/* position of x in the ascending range col[lo,hi), or -1 */
static inline int64_t find_sorted(const int32_t *col, int64_t lo, int64_t hi, int32_t x)
{
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        int32_t c = col[mid];
        if (c < x) lo = mid + 1;
        else if (c > x) hi = mid;
        else return mid;
    }
    return -1;
}

/* --------------------------------------------------------------- a1 simplify
 * Reference: igraph_simplify(&graph, true, true, NULL) at src/graph.cpp:438 on
 * the graph igraph_create()d (src/graph.cpp:418) from the raw clique-expanded
 * pairs (src/graph.cpp:379-389).  Observable result: a simple undirected graph
 * on the same nv vertices; |E| printed at src/graph.cpp:444. */
int64_t orc_simplify(int64_t nv, int64_t n_raw, const int64_t *uv,
                     int64_t *rowptr, int32_t *col)
{
    if (nv < 0 || n_raw < 0 || nv > INT32_MAX) return -1;
    uint64_t *keys = (uint64_t *)malloc((size_t)(n_raw > 0 ? n_raw : 1) * sizeof(uint64_t));
    uint64_t *tmp = (uint64_t *)malloc((size_t)(n_raw > 0 ? n_raw : 1) * sizeof(uint64_t));
    int64_t nk = 0;
    for (int64_t i = 0; i < n_raw; ++i) {
        int64_t u = uv[2 * i], v = uv[2 * i + 1];
        if (u < 0 || v < 0 || u >= nv || v >= nv) { free(keys); free(tmp); return -1; }
        if (u == v) continue;                         /* loops=true */
        uint64_t lo = (uint64_t)(u < v ? u : v), hi = (uint64_t)(u < v ? v : u);
        keys[nk++] = (lo << 32) | hi;
    }
    radix_sort_u64(keys, tmp, nk);
    int64_t ne = 0;
    for (int64_t i = 0; i < nk; ++i)                  /* multiple=true */
        if (i == 0 || keys[i] != keys[i - 1]) keys[ne++] = keys[i];

    memset(rowptr, 0, (size_t)(nv + 1) * sizeof(int64_t));
    for (int64_t i = 0; i < ne; ++i) {
        rowptr[(keys[i] >> 32) + 1]++;
        rowptr[(keys[i] & 0xFFFFFFFFu) + 1]++;
    }
    for (int64_t v = 0; v < nv; ++v) rowptr[v + 1] += rowptr[v];
    int64_t *cur = (int64_t *)malloc((size_t)(nv + 1) * sizeof(int64_t));
    memcpy(cur, rowptr, (size_t)(nv + 1) * sizeof(int64_t));
    /* lower neighbours first (ascending because keys are sorted by min), then
     * upper neighbours (ascending max within one min): rows come out sorted. */
    for (int64_t i = 0; i < ne; ++i) {
        int32_t lo = (int32_t)(keys[i] >> 32), hi = (int32_t)(keys[i] & 0xFFFFFFFFu);
        col[cur[hi]++] = lo;
    }
    for (int64_t i = 0; i < ne; ++i) {
        int32_t lo = (int32_t)(keys[i] >> 32), hi = (int32_t)(keys[i] & 0xFFFFFFFFu);
        col[cur[lo]++] = hi;
    }
    free(cur); free(keys); free(tmp);
    return ne;
}

/* ---------------------------------------------------------------- a2 degree
 * Reference: igraph_degree(&graph,&deg,igraph_vss_all(),IGRAPH_ALL,
 * IGRAPH_NO_LOOPS) at src/graph.cpp:462, after simplify: distinct neighbours. */
void orc_degree(int64_t nv, const int64_t *rowptr, int32_t *degree)
{
    for (int64_t v = 0; v < nv; ++v) degree[v] = (int32_t)(rowptr[v + 1] - rowptr[v]);
}

/* -------------------------------------------------------------- a3 coreness
 * Reference: igraph_coreness(&graph,&coreness,IGRAPH_ALL) at src/graph.cpp:463.
 * igraph implements Batagelj & Zaversnik (2003): bin-sort vertices by degree,
 * sweep in that order, and for every neighbour u of the swept vertex v with
 * cores[u] > cores[v] move u one bin down (swap with the first vertex of its
 * bin) and decrement cores[u]. */
int32_t orc_coreness(int64_t nv, const int64_t *rowptr, const int32_t *col,
                     int32_t *cores)
{
    if (nv == 0) return 0;
    int32_t maxdeg = 0;
    for (int64_t v = 0; v < nv; ++v) {
        cores[v] = (int32_t)(rowptr[v + 1] - rowptr[v]);
        if (cores[v] > maxdeg) maxdeg = cores[v];
    }
    int64_t *bin = (int64_t *)calloc((size_t)maxdeg + 2, sizeof(int64_t));
    int64_t *pos = (int64_t *)malloc((size_t)nv * sizeof(int64_t));
    int32_t *vert = (int32_t *)malloc((size_t)nv * sizeof(int32_t));
    for (int64_t v = 0; v < nv; ++v) bin[cores[v]]++;
    int64_t start = 0;
    for (int32_t d = 0; d <= maxdeg; ++d) { int64_t c = bin[d]; bin[d] = start; start += c; }
    for (int64_t v = 0; v < nv; ++v) { pos[v] = bin[cores[v]]; vert[pos[v]] = (int32_t)v; bin[cores[v]]++; }
    for (int32_t d = maxdeg; d > 0; --d) bin[d] = bin[d - 1];
    bin[0] = 0;
    for (int64_t i = 0; i < nv; ++i) {
        int32_t v = vert[i];
        for (int64_t j = rowptr[v]; j < rowptr[v + 1]; ++j) {
            int32_t u = col[j];
            if (cores[u] > cores[v]) {
                int32_t du = cores[u];
                int64_t pu = pos[u], pw = bin[du];
                int32_t w = vert[pw];
                if (u != w) { pos[u] = pw; pos[w] = pu; vert[pu] = w; vert[pw] = u; }
                bin[du]++;
                cores[u]--;
            }
        }
    }
    int32_t mx = 0;
    for (int64_t v = 0; v < nv; ++v) if (cores[v] > mx) mx = cores[v];
    free(bin); free(pos); free(vert);
    return mx;
}

/* ------------------------------------------------------ a5 induced subgraph
 * Reference: igraph_induced_subgraph_map(&graph,&subgraph,vids,
 * IGRAPH_SUBGRAPH_AUTO,&map,&invmap) at src/graph.cpp:502; vids are the
 * max-coreness vertices collected in ascending vid order at
 * src/graph.cpp:468-473, so new ids keep the old relative order and
 * invmap[new] = old (read at src/graph.cpp:531-532). */
int64_t orc_induced_subgraph(int64_t nv, const int64_t *rowptr, const int32_t *col,
                             const uint8_t *vmask, int64_t *sub_rowptr,
                             int32_t *sub_col, int32_t *invmap)
{
    int32_t *map = (int32_t *)malloc((size_t)(nv > 0 ? nv : 1) * sizeof(int32_t));
    int64_t ns = 0;
    for (int64_t v = 0; v < nv; ++v) {
        if (vmask[v]) { map[v] = (int32_t)ns; invmap[ns] = (int32_t)v; ns++; }
        else map[v] = -1;
    }
    int64_t out = 0;
    sub_rowptr[0] = 0;
    for (int64_t s = 0; s < ns; ++s) {
        int32_t v = invmap[s];
        for (int64_t j = rowptr[v]; j < rowptr[v + 1]; ++j)
            if (map[col[j]] >= 0) sub_col[out++] = map[col[j]];
        sub_rowptr[s + 1] = out;
    }
    free(map);
    return ns;
}

/* ------------------------------------------------------- canonical edge ids
 * Edge identity across the C-ABI is the pair (min,max) in lexicographic
 * order; igraph's internal edge ids are never observable
 * in KOMB's outputs.  ebase[u] = id of u's first upper edge, ustart[u] = slot
 * of u's first neighbour > u. */
static void upper_index(int64_t nv, const int64_t *rowptr, const int32_t *col,
                        int64_t *ustart, int64_t *ebase)
{
    int64_t e = 0;
    for (int64_t u = 0; u < nv; ++u) {
        int64_t lo = rowptr[u], hi = rowptr[u + 1];
        while (lo < hi) {                       /* first slot with col > u */
            int64_t mid = lo + ((hi - lo) >> 1);
            if (col[mid] > (int32_t)u) hi = mid; else lo = mid + 1;
        }
        ustart[u] = lo;
        ebase[u] = e;
        e += rowptr[u + 1] - lo;
    }
    ebase[nv] = e;
}

int64_t orc_edge_list(int64_t nv, const int64_t *rowptr, const int32_t *col,
                      int32_t *eu, int32_t *ev)
{
    int64_t e = 0;
    for (int64_t u = 0; u < nv; ++u)
        for (int64_t j = rowptr[u]; j < rowptr[u + 1]; ++j)
            if (col[j] > (int32_t)u) { eu[e] = (int32_t)u; ev[e] = col[j]; e++; }
    return e;
}

/* slot -> canonical edge id for every one of the 2*ne CSR slots */
static int32_t *slot_edge_ids(int64_t nv, const int64_t *rowptr, const int32_t *col,
                              const int64_t *ustart, const int64_t *ebase)
{
    int64_t ns = rowptr[nv];
    int32_t *eid = (int32_t *)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
    for (int64_t u = 0; u < nv; ++u) {
        for (int64_t j = rowptr[u]; j < rowptr[u + 1]; ++j) {
            int32_t x = col[j];
            if (x > (int32_t)u) eid[j] = (int32_t)(ebase[u] + (j - ustart[u]));
            else {
                int64_t p = find_sorted(col, ustart[x], rowptr[x + 1], (int32_t)u);
                eid[j] = (int32_t)(ebase[x] + (p - ustart[x]));
            }
        }
    }
    return eid;
}

/* For every common neighbour w of u and v call
 *   f(ctx, eid of (u,w), eid of (v,w)).
 * Iterates the shorter row and binary-searches the longer one (igraph's
 * sorted intersection switches to the same strategy for skewed sizes). */
typedef void (*tri_fn)(void *ctx, int32_t e1, int32_t e2);
static void for_common(const int64_t *rowptr, const int32_t *col, const int32_t *eid,
                       int32_t u, int32_t v, tri_fn f, void *ctx)
{
    int64_t du = rowptr[u + 1] - rowptr[u], dv = rowptr[v + 1] - rowptr[v];
    int32_t s = du <= dv ? u : v, l = du <= dv ? v : u;
    for (int64_t j = rowptr[s]; j < rowptr[s + 1]; ++j) {
        int32_t w = col[j];
        if (w == l) continue;
        int64_t p = find_sorted(col, rowptr[l], rowptr[l + 1], w);
        if (p >= 0) {
            if (s == u) f(ctx, eid[j], eid[p]); else f(ctx, eid[p], eid[j]);
        }
    }
}

/* ---------------------------------------------------------------- support
 * igraph_trussness (called at src/graph.cpp:508) starts by listing every
 * triangle and counting, per edge, the triangles through it.
 * Restated as: support[(u,v)] = |N(u) & N(v)|. */
struct cnt_ctx { int64_t n; };
static void cnt_cb(void *c, int32_t e1, int32_t e2) { (void)e1; (void)e2; ((struct cnt_ctx *)c)->n++; }

int64_t orc_support(int64_t nv, const int64_t *rowptr, const int32_t *col,
                    int32_t *support)
{
    int64_t *ustart = (int64_t *)malloc((size_t)(nv + 1) * sizeof(int64_t));
    int64_t *ebase = (int64_t *)malloc((size_t)(nv + 1) * sizeof(int64_t));
    upper_index(nv, rowptr, col, ustart, ebase);
    int32_t *eid = slot_edge_ids(nv, rowptr, col, ustart, ebase);
    int64_t tri3 = 0;
    for (int64_t u = 0; u < nv; ++u) {
        for (int64_t j = ustart[u]; j < rowptr[u + 1]; ++j) {
            struct cnt_ctx c = {0};
            for_common(rowptr, col, eid, (int32_t)u, col[j], cnt_cb, &c);
            support[eid[j]] = (int32_t)c.n;
            tri3 += c.n;
        }
    }
    free(ustart); free(ebase); free(eid);
    return tri3 / 3;
}

/* -------------------------------------------------------------- a6 trussness
 * Reference: igraph_trussness(&subgraph,&trussness) at src/graph.cpp:508.
 * igraph: support per edge; buckets of edges by support; for level = 0..max,
 * while the level's bucket is non-empty take an edge (a,b), and for every
 * common neighbour n whose two edges (a,n),(b,n) are both not yet completed,
 * move each of them that has support > level one bucket down; then
 * trussness[(a,b)] = level + 2 and the edge is completed.
 * Restated with the bin-sorted array + position index of Batagelj-Zaversnik
 * applied to edges (same peel order class, same results; the result is the
 * unique trussness of a simple graph). */
struct peel_ctx {
    int32_t *sup; uint8_t *done; int64_t *bin; int64_t *pos; int32_t *ord; int32_t k;
};
static inline void peel_lower(struct peel_ctx *p, int32_t e)
{
    if (p->sup[e] > p->k) {
        int32_t s = p->sup[e];
        int64_t pe = p->pos[e], pw = p->bin[s];
        int32_t w = p->ord[pw];
        if (e != w) { p->pos[e] = pw; p->pos[w] = pe; p->ord[pe] = w; p->ord[pw] = e; }
        p->bin[s]++;
        p->sup[e]--;
    }
}
static void peel_cb(void *c, int32_t e1, int32_t e2)
{
    struct peel_ctx *p = (struct peel_ctx *)c;
    if (p->done[e1] || p->done[e2]) return;
    peel_lower(p, e1);
    peel_lower(p, e2);
}

int32_t orc_trussness(int64_t nv, const int64_t *rowptr, const int32_t *col,
                      int32_t *truss)
{
    int64_t ne = rowptr[nv] / 2;
    if (ne == 0) return 0;
    int64_t *ustart = (int64_t *)malloc((size_t)(nv + 1) * sizeof(int64_t));
    int64_t *ebase = (int64_t *)malloc((size_t)(nv + 1) * sizeof(int64_t));
    upper_index(nv, rowptr, col, ustart, ebase);
    int32_t *eid = slot_edge_ids(nv, rowptr, col, ustart, ebase);
    int32_t *eu = (int32_t *)malloc((size_t)ne * sizeof(int32_t));
    int32_t *ev = (int32_t *)malloc((size_t)ne * sizeof(int32_t));
    orc_edge_list(nv, rowptr, col, eu, ev);

    int32_t *sup = (int32_t *)malloc((size_t)ne * sizeof(int32_t));
    int32_t maxs = 0;
    for (int64_t e = 0; e < ne; ++e) {
        struct cnt_ctx c = {0};
        for_common(rowptr, col, eid, eu[e], ev[e], cnt_cb, &c);
        sup[e] = (int32_t)c.n;
        if (sup[e] > maxs) maxs = sup[e];
    }
    int64_t *bin = (int64_t *)calloc((size_t)maxs + 2, sizeof(int64_t));
    int64_t *pos = (int64_t *)malloc((size_t)ne * sizeof(int64_t));
    int32_t *ord = (int32_t *)malloc((size_t)ne * sizeof(int32_t));
    uint8_t *done = (uint8_t *)calloc((size_t)ne, 1);
    for (int64_t e = 0; e < ne; ++e) bin[sup[e]]++;
    int64_t start = 0;
    for (int32_t s = 0; s <= maxs; ++s) { int64_t c = bin[s]; bin[s] = start; start += c; }
    for (int64_t e = 0; e < ne; ++e) { pos[e] = bin[sup[e]]; ord[pos[e]] = (int32_t)e; bin[sup[e]]++; }
    for (int32_t s = maxs; s > 0; --s) bin[s] = bin[s - 1];
    bin[0] = 0;

    struct peel_ctx p = { sup, done, bin, pos, ord, 0 };
    int32_t mx = 0;
    for (int64_t i = 0; i < ne; ++i) {
        int32_t e = ord[i];
        p.k = sup[e];
        truss[e] = p.k + 2;
        if (truss[e] > mx) mx = truss[e];
        if (p.k > 0) for_common(rowptr, col, eid, eu[e], ev[e], peel_cb, &p);
        done[e] = 1;
    }
    free(ustart); free(ebase); free(eid); free(eu); free(ev);
    free(sup); free(bin); free(pos); free(ord); free(done);
    return mx;
}

/* ------------------------------------------------- a6, all-cores variant
 * The same trussness with every host core: supports by a parallel loop over the
 * edges, then a level-synchronous parallel peel in the manner of Kabir & Madduri's
 * PKT (2017): level l's edges form the current set; every thread takes edges of
 * the set, and a triangle whose other two edges are both unprocessed loses one
 * support on each edge that is above the level -- on one edge only, by the smaller
 * edge id, when the other one is in the current set too; an edge that lands on the
 * level joins the next set.  Trussness is unique, so the values equal
 * orc_trussness's (tests/test_oracle.py).  This is NOT how the reference runs
 * (igraph is single-threaded, src/graph.cpp:508 is called from one thread): it
 * is the "fairer CPU ceiling" of SURVEY section 8(d), reported next to the
 * single-thread figure.  Built only with OpenMP (oracle/Makefile target native). */
#ifdef _OPENMP
#include <omp.h>
struct pkt_ctx { int32_t *S; const uint8_t *processed; const uint8_t *in_curr; int32_t *next; int64_t *next_n; uint8_t *in_next; int32_t e, l; };
static inline void pkt_dec(struct pkt_ctx *p, int32_t x)
{
    int32_t old = __atomic_fetch_sub(&p->S[x], 1, __ATOMIC_RELAXED);
    if (old == p->l + 1) { int64_t i = __atomic_fetch_add(p->next_n, 1, __ATOMIC_RELAXED); p->next[i] = x; p->in_next[x] = 1; }
    else if (old <= p->l) __atomic_fetch_add(&p->S[x], 1, __ATOMIC_RELAXED);       /* it was on the level already */
}
static void pkt_cb(void *c, int32_t e1, int32_t e2)
{
    struct pkt_ctx *p = (struct pkt_ctx *)c;
    if (p->processed[e1] || p->processed[e2]) return;
    const int32_t s1 = __atomic_load_n(&p->S[e1], __ATOMIC_RELAXED), s2 = __atomic_load_n(&p->S[e2], __ATOMIC_RELAXED);
    if (s1 > p->l && s2 > p->l) { pkt_dec(p, e1); pkt_dec(p, e2); }
    else if (s1 > p->l) { if (!p->in_curr[e2] || p->e < e2) pkt_dec(p, e1); }
    else if (s2 > p->l) { if (!p->in_curr[e1] || p->e < e1) pkt_dec(p, e2); }
}
int32_t orc_trussness_omp(int64_t nv, const int64_t *rowptr, const int32_t *col, int32_t *truss, int nthreads)
{
    int64_t ne = rowptr[nv] / 2;
    if (ne == 0) return 0;
    if (nthreads > 0) omp_set_num_threads(nthreads);
    int64_t *ustart = (int64_t *)malloc((size_t)(nv + 1) * sizeof(int64_t));
    int64_t *ebase = (int64_t *)malloc((size_t)(nv + 1) * sizeof(int64_t));
    upper_index(nv, rowptr, col, ustart, ebase);
    int32_t *eid = slot_edge_ids(nv, rowptr, col, ustart, ebase);
    int32_t *eu = (int32_t *)malloc((size_t)ne * sizeof(int32_t));
    int32_t *ev = (int32_t *)malloc((size_t)ne * sizeof(int32_t));
    orc_edge_list(nv, rowptr, col, eu, ev);
    int32_t *S = (int32_t *)malloc((size_t)ne * sizeof(int32_t));
    uint8_t *processed = (uint8_t *)calloc((size_t)ne, 1), *in_curr = (uint8_t *)calloc((size_t)ne, 1), *in_next = (uint8_t *)calloc((size_t)ne, 1);
    int32_t *curr = (int32_t *)malloc((size_t)ne * sizeof(int32_t)), *next = (int32_t *)malloc((size_t)ne * sizeof(int32_t));
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t e = 0; e < ne; ++e) {
        struct cnt_ctx c = {0};
        for_common(rowptr, col, eid, eu[e], ev[e], cnt_cb, &c);
        S[e] = (int32_t)c.n;
    }
    int64_t todo = ne;
    int32_t mx = 0;
    for (int32_t l = 0; todo > 0; ++l) {
        int64_t curr_n = 0;
#pragma omp parallel for schedule(static)
        for (int64_t e = 0; e < ne; ++e)
            if (!processed[e] && S[e] == l) { int64_t i = __atomic_fetch_add(&curr_n, 1, __ATOMIC_RELAXED); curr[i] = (int32_t)e; in_curr[e] = 1; }
        while (curr_n > 0) {
            todo -= curr_n;
            mx = l + 2;
            int64_t next_n = 0;
#pragma omp parallel for schedule(dynamic, 64)
            for (int64_t i = 0; i < curr_n; ++i) {
                struct pkt_ctx p = { S, processed, in_curr, next, &next_n, in_next, curr[i], l };
                if (l > 0) for_common(rowptr, col, eid, eu[p.e], ev[p.e], pkt_cb, &p);
            }
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < curr_n; ++i) { const int32_t e = curr[i]; processed[e] = 1; in_curr[e] = 0; truss[e] = l + 2; }
            int32_t *t = curr; curr = next; next = t;
            uint8_t *tf = in_curr; in_curr = in_next; in_next = tf;
            curr_n = next_n;
        }
    }
    free(ustart); free(ebase); free(eid); free(eu); free(ev); free(S); free(processed); free(in_curr); free(in_next); free(curr); free(next);
    return mx;
}
#endif

/* ------------------------------------------------------ a10 fractionalRank
 * Reference: CoreA::fractionalRank, src/CoreA.h:142-187.  Scores truncated to
 * int (:149), sorted descending (:152), uniqued (:154); then for every unique
 * value one pass that advances an int rank counter over the matching entries
 * and sums the ranks in a double (:164-172), avg /= cnt (:174), and a second
 * pass that assigns avg (:176-182). */
static int cmp_int_desc(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x < y) - (x > y);
}
void orc_fractional_rank_faithful(const double *scores, int64_t n, double *out)
{
    int *list = (int *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    for (int64_t i = 0; i < n; ++i) list[i] = (int)scores[i];
    qsort(list, (size_t)n, sizeof(int), cmp_int_desc);
    int64_t nu = 0;
    for (int64_t i = 0; i < n; ++i)
        if (i == 0 || list[i] != list[i - 1]) list[nu++] = list[i];
    int rank = 0;
    for (int64_t t = 0; t < nu; ++t) {
        int value = list[t];
        double avg = 0.0;
        int cnt = 0;
        for (int64_t i = 0; i < n; ++i)
            if (scores[i] == value) { rank++; cnt++; avg += rank; }
        avg /= cnt;
        for (int64_t i = 0; i < n; ++i)
            if (scores[i] == value) out[i] = avg;
    }
    free(list);
}

struct kv { int64_t key; int64_t idx; };
static int cmp_kv_desc(const void *a, const void *b)
{
    const struct kv *x = (const struct kv *)a, *y = (const struct kv *)b;
    return (x->key < y->key) - (x->key > y->key);
}
void orc_fractional_rank_fast(const int64_t *keys, int64_t n, double *out)
{
    struct kv *a = (struct kv *)malloc((size_t)(n > 0 ? n : 1) * sizeof(struct kv));
    for (int64_t i = 0; i < n; ++i) { a[i].key = keys[i]; a[i].idx = i; }
    qsort(a, (size_t)n, sizeof(struct kv), cmp_kv_desc);
    int64_t i = 0;
    while (i < n) {
        int64_t j = i;
        while (j + 1 < n && a[j + 1].key == a[i].key) ++j;
        /* positions i+1 .. j+1 (1-based): mean = (first+last)/2, exact */
        double avg = (double)((i + 1) + (j + 1)) / 2.0;
        for (int64_t t = i; t <= j; ++t) out[a[t].idx] = avg;
        i = j + 1;
    }
    free(a);
}

/* ------------------------------------------------------ a9 getAnomalyScore
 * Reference: CoreA::getAnomalyScore, src/CoreA.h:109-140:
 *   corenessWithDegree[i] = coreness[i] * n + degree[i]   (int arithmetic, :122)
 *   anomaly[i] = |log(degreeRank[i]) - log(corenessRank[i])|            (:131) */
void orc_corea_scores(const int32_t *degree, const int32_t *coreness, int64_t n,
                      int faithful, double *score)
{
    double *rd = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    double *rc = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    if (faithful) {
        double *kd = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
        double *kc = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
        int ni = (int)n;
        for (int64_t i = 0; i < n; ++i) {
            kd[i] = degree[i];
            kc[i] = coreness[i] * ni + degree[i];      /* int, as src/CoreA.h:122 */
        }
        orc_fractional_rank_faithful(kc, n, rc);
        orc_fractional_rank_faithful(kd, n, rd);
        free(kd); free(kc);
    } else {
        int64_t *kd = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
        int64_t *kc = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
        for (int64_t i = 0; i < n; ++i) {
            kd[i] = degree[i];
            kc[i] = (int64_t)coreness[i] * n + degree[i];
        }
        orc_fractional_rank_fast(kc, n, rc);
        orc_fractional_rank_fast(kd, n, rd);
        free(kd); free(kc);
    }
    for (int64_t i = 0; i < n; ++i) score[i] = fabs(log(rd[i]) - log(rc[i]));
    free(rd); free(rc);
}

/* ------------------------------------------- a12 + a13 HashIndexedMinHeap, runMerge
 * Reference: src/HashIndexedMinHeap.h:10-238 (array-backed binary min-heap over int ids with a position index:
 * insert = append + refreshPriority; poll = move the last id to the root + sift down; refreshPriority = sift down,
 * and only if nothing moved, sift up while the parent's value is GREATER) and its only user,
 * CombineCoreA::runMerge, src/CombineCoreA.h:45-219 (dead code in the reference: no caller): a greedy
 * densest-block peel over a "row" copy and a "column" copy of the graph.  Every node starts on both sides with
 * priority = suspiciousness + degree; each step polls the smaller of the two heap minima (the row side only when
 * strictly smaller, or the column side is empty), subtracts its priority from the running sum, records
 * density = sum / nodes left, and lowers by one the priority of the node's neighbours on the OTHER side that are
 * still there.  Output: removal order and sides (filled from the back, as :137-138), the number of nodes of the
 * densest block and its density (:140-145).  Which of several equal priorities is polled first is decided by
 * the heap's array layout, so the heap is restated operation for operation; tests/ pin this against the
 * reference's own class compiled in place (oracle/_ref/merge_ref).  removed[][] is zero-initialised (the
 * reference reads it uninitialised, :104-108), and the reference's `cols` vector sized by the number of rows
 * (:191) is not modelled: rows / cols of the block are order[i] for i < n_block with side[i] == 0 / 1. */
struct mheap { int32_t *arr; int32_t *pos; double *val; int32_t size; };
static int mh_down(struct mheap *h, int32_t p)                /* minHeapfy, :171-217 (iterative; same moves) */
{
    int moved = 0;
    for (;;) {
        const int32_t l = 2 * (p + 1) - 1, r = 2 * (p + 1);
        const int32_t cur = h->arr[p];
        int32_t sp = p, sk = cur;
        if (l < h->size && h->val[h->arr[l]] < h->val[cur]) { sp = l; sk = h->arr[l]; }
        if (r < h->size && h->val[h->arr[r]] < h->val[sk]) { sp = r; sk = h->arr[r]; }
        if (sp == p) return moved;
        h->arr[p] = sk; h->pos[sk] = p;
        h->arr[sp] = cur; h->pos[cur] = sp;
        p = sp; moved = 1;
    }
}
static void mh_refresh(struct mheap *h, int32_t key, double v)   /* refreshPriority, :138-167 */
{
    h->val[key] = v;
    int32_t p = h->pos[key];
    if (mh_down(h, p) || p <= 0) return;
    int32_t pp = (p + 1) / 2 - 1;
    while (p > 0 && h->val[h->arr[pp]] > h->val[key]) {
        const int32_t pe = h->arr[pp];
        h->arr[pp] = key; h->pos[key] = pp;
        h->arr[p] = pe; h->pos[pe] = p;
        p = pp; pp = (p + 1) / 2 - 1;
    }
}
static void mh_insert(struct mheap *h, int32_t key, double v)    /* insert, :83-98 */
{
    const int32_t p = h->size++;
    h->arr[p] = key; h->pos[key] = p; h->val[key] = v;
    mh_refresh(h, key, v);
}
static int32_t mh_poll(struct mheap *h, double *v)               /* poll, :55-81 (size > 0) */
{
    const int32_t top = h->arr[0];
    *v = h->val[top];
    h->pos[top] = -1;
    if (h->size != 1) {
        const int32_t last = h->arr[h->size - 1];
        h->arr[0] = last; h->pos[last] = 0;
        h->size--;
        mh_down(h, 0);
    } else h->size--;
    h->arr[h->size] = 0;
    return top;
}
int32_t orc_run_merge(int64_t nv, const int64_t *rowptr, const int32_t *col, const double *susp,
                      int32_t *order, int32_t *side, double *max_density)
{
    const int32_t n = (int32_t)nv;
    struct mheap h[2];
    for (int s = 0; s < 2; ++s) {
        h[s].arr = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
        h[s].pos = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
        h[s].val = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
        h[s].size = 0;
        for (int32_t v = 0; v < n; ++v) h[s].pos[v] = -1;
    }
    double *p0 = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)), *p1 = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    double sum = 0;
    if (susp) for (int32_t v = 0; v < n; ++v) { p0[v] = susp[v]; p1[v] = susp[v]; sum += 2 * susp[v]; }   /* :58-67 */
    long slots = 0;
    for (int32_t v = 0; v < n; ++v)                                                                       /* :71-85 */
        for (int64_t j = rowptr[v]; j < rowptr[v + 1]; ++j) { p0[v] += 1; p1[col[j]] += 1; ++slots; }
    sum += (double)slots;                                                                                   /* :87 */
    for (int32_t v = 0; v < n; ++v) mh_insert(&h[0], v, p0[v]);                                            /* :89-93 */
    for (int32_t v = 0; v < n; ++v) mh_insert(&h[1], v, p1[v]);                                            /* :95-99 */
    uint8_t *gone = (uint8_t *)calloc((size_t)2 * (size_t)(n > 0 ? n : 1), 1);
    double best = 0;
    int32_t best_left = 0;
    for (int32_t left = 2 * n; left >= 1;) {                                                                /* :114-174 */
        const int s = (h[0].size > 0 && (h[1].size == 0 || h[0].val[h[0].arr[0]] < h[1].val[h[1].arr[0]])) ? 0 : 1;
        double pv;
        const int32_t node = mh_poll(&h[s], &pv);
        sum -= pv;
        --left;
        order[left] = node; side[left] = s;
        if (left >= 1) { const double d = sum / left; if (d > best) { best = d; best_left = left; } }
        gone[(size_t)s * (size_t)(n > 0 ? n : 1) + node] = 1;
        struct mheap *o = &h[s ^ 1];
        for (int64_t j = rowptr[node]; j < rowptr[node + 1]; ++j) {
            const int32_t w = col[j];
            if (!gone[(size_t)(s ^ 1) * (size_t)(n > 0 ? n : 1) + w]) mh_refresh(o, w, o->val[w] - 1);
        }
    }
    for (int s = 0; s < 2; ++s) { free(h[s].arr); free(h[s].pos); free(h[s].val); }
    free(p0); free(p1); free(gone);
    if (max_density) *max_density = best;
    return best_left;
}
