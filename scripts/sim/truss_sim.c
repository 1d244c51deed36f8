// truss_sim.c -- CPU study (not product, not oracle): dependent steps of k-truss schemes on a given graph.
// Builds the triangle incidence index (edge -> pairs of the other two edges of each triangle), then
//  (1) level-synchronous peel: sub-rounds per level, live edges at each level start;
//  (2) h-index iteration (Sariyuce, Seshadhri, Pinar 2017) from the supports: iterations / changed / work;
//  (3) peel the levels < L0, then h-index on the remainder (with the "crossing" notification filter).
// usage: truss_sim graph.bin [L0 ...]        (gcc -O2 -fopenmp)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

static int64_t nv, ns, m;
static int64_t *rowptr;
static int32_t *col;
static int64_t *off;          // [m+1]
static int32_t *ix, *iy;      // incidence pairs

static int64_t upper_first(int64_t u)
{
    int64_t lo = rowptr[u], hi = rowptr[u + 1];
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (col[mid] > u) hi = mid; else lo = mid + 1; }
    return lo;
}
static int64_t *ebase;
static int64_t edge_id(int32_t a, int32_t b)   // a < b
{
    int64_t f = upper_first(a), lo = f, hi = rowptr[a + 1];
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (col[mid] < b) lo = mid + 1; else hi = mid; }
    return ebase[a] + (lo - f);
}

static int32_t hidx(int64_t e, const int32_t *t, const uint8_t *alive, int32_t cap, int32_t *cnt, int64_t *work)
{
    for (int32_t k = 0; k <= cap; ++k) cnt[k] = 0;
    for (int64_t j = off[e]; j < off[e + 1]; ++j) {
        const int32_t x = ix[j], y = iy[j];
        if (alive && (!alive[x] || !alive[y])) continue;
        int32_t v = t[x] < t[y] ? t[x] : t[y];
        if (v > cap) v = cap;
        cnt[v]++; ++*work;
    }
    int32_t sum = 0;
    for (int32_t k = cap; k >= 0; --k) { sum += cnt[k]; if (sum >= k) return k; }
    return 0;
}

static void hrun(const char *tag, int32_t *t, const uint8_t *alive, const int32_t *truth, int async, int filter)
{
    int32_t maxc = 0; int64_t nalive = 0;
    for (int64_t e = 0; e < m; ++e) if (!alive || alive[e]) { ++nalive; if (t[e] > maxc) maxc = t[e]; }
    int32_t *cnt = malloc((size_t)(maxc + 2) * 4), *nt = malloc((size_t)m * 4);
    uint8_t *act = malloc((size_t)m), *nact = calloc((size_t)m, 1);
    for (int64_t e = 0; e < m; ++e) act[e] = (!alive || alive[e]);
    printf("%s: %lld edges, max bound %d, %s, %s notification\n", tag, (long long)nalive, maxc, async ? "in-place" : "synchronous",
           filter == 2 ? "lower-only" : filter ? "crossing-filtered" : "any-change");
    long long total = 0;
    for (int it = 1; it < 100000; ++it) {
        long long changed = 0, active = 0; int64_t work = 0;
        if (!async) memcpy(nt, t, (size_t)m * 4);
        for (int64_t e = 0; e < m; ++e) {
            if (!act[e]) continue;
            ++active;
            const int32_t a = t[e];
            const int32_t h = hidx(e, t, alive, a, cnt, &work);
            if (h != a) {
                ++changed;
                if (async) t[e] = h; else nt[e] = h;
                for (int64_t j = off[e]; j < off[e + 1]; ++j) {
                    const int32_t x = ix[j], y = iy[j];
                    if (alive && (!alive[x] || !alive[y])) continue;
                    if (!filter) { nact[x] = 1; nact[y] = 1; }
                    else if (filter == 2) { if (t[x] > h) nact[x] = 1; if (t[y] > h) nact[y] = 1; }
                    else { if (t[x] > h && t[x] <= a) nact[x] = 1; if (t[y] > h && t[y] <= a) nact[y] = 1; }
                }
            }
        }
        if (!async) memcpy(t, nt, (size_t)m * 4);
        total += work;
        if (it <= 10 || it % 10 == 0 || changed == 0)
            printf("  iter %3d: active %9lld changed %9lld live items visited %11lld\n", it, active, changed, (long long)work);
        if (!changed) { printf("  converged after %d iterations, items visited %lld\n", it, total); break; }
        memcpy(act, nact, (size_t)m); memset(nact, 0, (size_t)m);
    }
    long long bad = 0;
    for (int64_t e = 0; e < m; ++e) if ((!alive || alive[e]) && t[e] != truth[e]) ++bad;
    printf("  mismatches vs peel: %lld\n", bad);
    free(cnt); free(nt); free(act); free(nact);
}

int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    int64_t hdr[2];
    if (!f || fread(hdr, 8, 2, f) != 2) return 1;
    nv = hdr[0]; ns = hdr[1]; m = ns / 2;
    rowptr = malloc((size_t)(nv + 1) * 8); col = malloc((size_t)ns * 4);
    if (fread(rowptr, 8, (size_t)nv + 1, f) != (size_t)nv + 1 || fread(col, 4, (size_t)ns, f) != (size_t)ns) return 1;
    fclose(f);
    ebase = malloc((size_t)(nv + 1) * 8);
    ebase[0] = 0;
    for (int64_t u = 0; u < nv; ++u) ebase[u + 1] = ebase[u] + (rowptr[u + 1] - upper_first(u));

    // ---- supports (canonical edge ids), then the incidence index
    int32_t *sup = calloc((size_t)m, 4);
    double t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t u = 0; u < nv; ++u) {
        const int64_t fu = upper_first(u);
        for (int64_t j = fu; j < rowptr[u + 1]; ++j) {
            const int32_t v = col[j];
            int64_t a = fu, b = upper_first(v);
            // common neighbours w > v of u and v: triangle u < v < w
            while (a < rowptr[u + 1] && b < rowptr[v + 1]) {
                if (col[a] < col[b]) ++a; else if (col[a] > col[b]) ++b;
                else {
                    const int32_t w = col[a];
                    if (w > v) {
                        const int64_t e0 = ebase[u] + (j - fu), e1 = ebase[u] + (a - fu), e2 = edge_id(v, w);
#pragma omp atomic
                        sup[e0]++;
#pragma omp atomic
                        sup[e1]++;
#pragma omp atomic
                        sup[e2]++;
                    }
                    ++a; ++b;
                }
            }
        }
    }
    off = malloc((size_t)(m + 1) * 8); off[0] = 0;
    for (int64_t e = 0; e < m; ++e) off[e + 1] = off[e] + sup[e];
    const int64_t T3 = off[m];
    ix = malloc((size_t)T3 * 4); iy = malloc((size_t)T3 * 4);
    int32_t *fill = calloc((size_t)m, 4);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t u = 0; u < nv; ++u) {
        const int64_t fu = upper_first(u);
        for (int64_t j = fu; j < rowptr[u + 1]; ++j) {
            const int32_t v = col[j];
            int64_t a = fu, b = upper_first(v);
            while (a < rowptr[u + 1] && b < rowptr[v + 1]) {
                if (col[a] < col[b]) ++a; else if (col[a] > col[b]) ++b;
                else {
                    const int32_t w = col[a];
                    if (w > v) {
                        const int64_t e0 = ebase[u] + (j - fu), e1 = ebase[u] + (a - fu), e2 = edge_id(v, w);
                        int32_t p0, p1, p2;
#pragma omp atomic capture
                        p0 = fill[e0]++;
#pragma omp atomic capture
                        p1 = fill[e1]++;
#pragma omp atomic capture
                        p2 = fill[e2]++;
                        ix[off[e0] + p0] = (int32_t)e1; iy[off[e0] + p0] = (int32_t)e2;
                        ix[off[e1] + p1] = (int32_t)e0; iy[off[e1] + p1] = (int32_t)e2;
                        ix[off[e2] + p2] = (int32_t)e0; iy[off[e2] + p2] = (int32_t)e1;
                    }
                    ++a; ++b;
                }
            }
        }
    }
    printf("graph: nv %lld m %lld triangles %lld (index built in %.1f s)\n", (long long)nv, (long long)m, (long long)(T3 / 3), omp_get_wtime() - t0);

    // ---- level-synchronous peel
    int32_t *s = malloc((size_t)m * 4), *truth = malloc((size_t)m * 4), *stamp = malloc((size_t)m * 4);
    uint8_t *alive = malloc((size_t)m);
    int32_t *cur = malloc((size_t)m * 4), *nxt = malloc((size_t)m * 4);
    memcpy(s, sup, (size_t)m * 4);
    for (int64_t e = 0; e < m; ++e) { alive[e] = 1; stamp[e] = 0x7fffffff; }
    int nL0 = argc - 2; int32_t L0s[16]; for (int i = 0; i < nL0 && i < 16; ++i) L0s[i] = atoi(argv[2 + i]);
    int32_t **snap_s = calloc(16, sizeof(void *)); uint8_t **snap_alive = calloc(16, sizeof(void *));
    int64_t left = m; int total_rounds = 0, levels = 0, round = 0;
    t0 = omp_get_wtime();
    for (int32_t L = 0; left > 0; ++L) {
        for (int i = 0; i < nL0; ++i) if (L0s[i] == L) {
            snap_s[i] = malloc((size_t)m * 4); snap_alive[i] = malloc((size_t)m);
            memcpy(snap_s[i], s, (size_t)m * 4); memcpy(snap_alive[i], alive, (size_t)m);
        }
        int64_t nc = 0;
        for (int64_t e = 0; e < m; ++e) if (alive[e] && s[e] <= L) cur[nc++] = (int32_t)e;
        if (!nc) continue;
        ++levels;
        const int64_t at_start = left;
        int rounds = 0; int64_t peeled = 0, items = 0, live_items = 0;
        while (nc) {
            ++rounds; ++round;
            for (int64_t i = 0; i < nc; ++i) { stamp[cur[i]] = round; truth[cur[i]] = L; }
            int64_t nn = 0;
            for (int64_t i = 0; i < nc; ++i) {
                const int32_t e = cur[i];
                items += off[e + 1] - off[e];
                for (int64_t j = off[e]; j < off[e + 1]; ++j) {
                    const int32_t x = ix[j], y = iy[j];
                    if (stamp[x] < round || stamp[y] < round) continue;
                    ++live_items;
                    const int xin = stamp[x] == round, yin = stamp[y] == round;
                    if (!xin && (!yin || e < y)) { if (--s[x] == L) nxt[nn++] = x; }
                    if (!yin && (!xin || e < x)) { if (--s[y] == L) nxt[nn++] = y; }
                }
            }
            for (int64_t i = 0; i < nc; ++i) alive[cur[i]] = 0;
            peeled += nc; left -= nc;
            int32_t *t = cur; cur = nxt; nxt = t; nc = nn;
        }
        total_rounds += rounds;
        printf("level %3d: live at start %9lld rounds %3d peeled %9lld items %11lld live items %10lld\n", L, (long long)at_start, rounds,
               (long long)peeled, (long long)items, (long long)live_items);
    }
    printf("peel: %d levels, %d sub-rounds (%.1f s)\n", levels, total_rounds, omp_get_wtime() - t0);
    fflush(stdout);

    int32_t *t = malloc((size_t)m * 4);
    if (getenv("SIM_WHOLE")) {
        for (int async = 0; async <= 1; ++async) { memcpy(t, sup, (size_t)m * 4); hrun("h-index, whole graph", t, NULL, truth, async, 2); }
    }
    for (int i = 0; i < nL0; ++i) {
        if (!snap_s[i]) continue;
        char tag[64]; snprintf(tag, sizeof tag, "h-index after peeling levels < %d", L0s[i]);
        memcpy(t, snap_s[i], (size_t)m * 4); hrun(tag, t, snap_alive[i], truth, 0, 2);
        memcpy(t, snap_s[i], (size_t)m * 4); hrun(tag, t, snap_alive[i], truth, 1, 2);
        fflush(stdout);
    }
    return 0;
}
