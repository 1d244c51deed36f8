// core_sim.c -- CPU study (not product, not oracle): how many dependent steps do different k-core schemes need
// on a given graph?  (1) level-synchronous peel: sub-rounds per level; (2) h-index iteration (Lu et al. 2016,
// Sariyuce et al. 2017 "local algorithms") from the degrees: iterations, vertices changed, work of the
// notified vertices per iteration; (3) peel up to level L0, then h-index on what is left.
// usage: core_sim graph.bin [L0 ...]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int64_t nv, ns;
static int64_t *rowptr;
static int32_t *col;

static int32_t hindex(int64_t v, const int32_t *c, const uint8_t *alive, int32_t cap, int32_t *cnt)
{
    // largest h <= cap with at least h live neighbours u having c[u] >= h
    for (int32_t k = 0; k <= cap; ++k) cnt[k] = 0;
    for (int64_t j = rowptr[v]; j < rowptr[v + 1]; ++j) {
        const int32_t u = col[j];
        if (alive && !alive[u]) continue;
        const int32_t x = c[u] < cap ? c[u] : cap;
        cnt[x]++;
    }
    int32_t sum = 0;
    for (int32_t k = cap; k >= 0; --k) { sum += cnt[k]; if (sum >= k) return k; }
    return 0;
}

static void hindex_run(const char *tag, int32_t *c, const uint8_t *alive, const int32_t *truth, int async, int filter)
{
    // c = upper bounds on entry (live degrees); iterate to the fixed point
    int32_t maxc = 0;
    for (int64_t v = 0; v < nv; ++v) if ((!alive || alive[v]) && c[v] > maxc) maxc = c[v];
    int32_t *cnt = malloc((size_t)(maxc + 2) * sizeof(int32_t));
    int32_t *nc = malloc((size_t)nv * sizeof(int32_t));
    uint8_t *act = malloc((size_t)nv), *nact = calloc((size_t)nv, 1);
    int64_t nalive = 0;
    for (int64_t v = 0; v < nv; ++v) { act[v] = (!alive || alive[v]); nalive += act[v]; }
    printf("%s: %lld vertices, max bound %d, %s updates, filter %d (0 any change, 1 crossing, 2 lower-only)\n", tag, (long long)nalive, maxc, async ? "in-place (async)" : "synchronous", filter);
    long long total_work = 0;
    for (int it = 1; it < 100000; ++it) {
        long long changed = 0, active = 0, work = 0;
        if (!async) memcpy(nc, c, (size_t)nv * sizeof(int32_t));
        for (int64_t v = 0; v < nv; ++v) {
            if (!act[v]) continue;
            ++active;
            for (int64_t j = rowptr[v]; j < rowptr[v + 1]; ++j) if (!alive || alive[col[j]]) ++work;   // live slots only (compacted rows)
            const int32_t a = c[v];
            const int32_t h = hindex(v, c, alive, c[v], cnt);
            if (h != c[v]) {
                ++changed;
                if (async) c[v] = h; else nc[v] = h;
                for (int64_t j = rowptr[v]; j < rowptr[v + 1]; ++j) {
                    const int32_t u = col[j];
                    if (alive && !alive[u]) continue;
                    if (filter == 0 || (filter == 2 && c[u] > h) || (filter == 1 && c[u] > h && c[u] <= a)) nact[u] = 1;
                }
            }
        }
        if (!async) memcpy(c, nc, (size_t)nv * sizeof(int32_t));
        total_work += work;
        if (it <= 12 || it % 10 == 0 || changed == 0)
            printf("  iter %3d: active %9lld changed %9lld slots walked %11lld\n", it, active, changed, work);
        if (!changed) { printf("  converged after %d iterations, total slots walked %lld (graph has %lld)\n", it, total_work, (long long)ns); break; }
        memcpy(act, nact, (size_t)nv); memset(nact, 0, (size_t)nv);
    }
    long long bad = 0;
    for (int64_t v = 0; v < nv; ++v) if ((!alive || alive[v]) && c[v] != truth[v]) ++bad;
    printf("  mismatches vs peel: %lld\n", bad);
    free(cnt); free(nc); free(act); free(nact);
}

int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    int64_t hdr[2];
    if (!f || fread(hdr, 8, 2, f) != 2) return 1;
    nv = hdr[0]; ns = hdr[1];
    rowptr = malloc((size_t)(nv + 1) * 8); col = malloc((size_t)ns * 4);
    if (fread(rowptr, 8, (size_t)nv + 1, f) != (size_t)nv + 1 || fread(col, 4, (size_t)ns, f) != (size_t)ns) return 1;
    fclose(f);

    // ---- level-synchronous peel: sub-rounds per level
    int32_t *deg = malloc((size_t)nv * 4), *core = malloc((size_t)nv * 4);
    uint8_t *alive = malloc((size_t)nv);
    int32_t *cur = malloc((size_t)nv * 4), *nxt = malloc((size_t)nv * 4);
    for (int64_t v = 0; v < nv; ++v) { deg[v] = (int32_t)(rowptr[v + 1] - rowptr[v]); alive[v] = 1; core[v] = -1; }
    int64_t left = nv; int total_rounds = 0, levels = 0;
    // snapshots for the hybrid runs
    int nL0 = argc - 2; int32_t L0s[16]; for (int i = 0; i < nL0 && i < 16; ++i) L0s[i] = atoi(argv[2 + i]);
    int32_t **snap_deg = calloc(16, sizeof(void *)); uint8_t **snap_alive = calloc(16, sizeof(void *));
    for (int32_t L = 0; left > 0; ++L) {
        for (int i = 0; i < nL0; ++i) if (L0s[i] == L) {
            snap_deg[i] = malloc((size_t)nv * 4); snap_alive[i] = malloc((size_t)nv);
            memcpy(snap_deg[i], deg, (size_t)nv * 4); memcpy(snap_alive[i], alive, (size_t)nv);
        }
        int64_t nc = 0;
        for (int64_t v = 0; v < nv; ++v) if (alive[v] && deg[v] <= L) cur[nc++] = (int32_t)v;
        if (!nc) continue;
        ++levels;
        int rounds = 0; int64_t peeled = 0, slots = 0;
        while (nc) {
            ++rounds;
            for (int64_t i = 0; i < nc; ++i) { alive[cur[i]] = 0; core[cur[i]] = L; }
            int64_t nn = 0;
            for (int64_t i = 0; i < nc; ++i) {
                const int32_t v = cur[i];
                slots += rowptr[v + 1] - rowptr[v];
                for (int64_t j = rowptr[v]; j < rowptr[v + 1]; ++j) {
                    const int32_t u = col[j];
                    if (alive[u] && --deg[u] == L) nxt[nn++] = u;   // exactly-on-level trigger (deg was > L)
                }
            }
            // a vertex may be triggered once only: deg hits L exactly once
            peeled += nc; left -= nc;
            int32_t *t = cur; cur = nxt; nxt = t; nc = nn;
        }
        total_rounds += rounds;
        printf("level %3d: rounds %3d peeled %9lld slots %11lld left %9lld\n", L, rounds, (long long)peeled, (long long)slots, (long long)left);
    }
    printf("peel: %d levels, %d sub-rounds\n", levels, total_rounds);

    // ---- h-index from the degrees, whole graph
    int32_t *c = malloc((size_t)nv * 4);
    if (getenv("SIM_WHOLE")) for (int async = 0; async <= 1; ++async) {
        for (int64_t v = 0; v < nv; ++v) c[v] = (int32_t)(rowptr[v + 1] - rowptr[v]);
        hindex_run("h-index, whole graph", c, NULL, core, async, 2);
    }
    // ---- hybrid: peel below L0, h-index above
    for (int i = 0; i < nL0; ++i) {
        if (!snap_deg[i]) continue;
        for (int mode = 0; mode < 4; ++mode) {
            memcpy(c, snap_deg[i], (size_t)nv * 4);
            char tag[64]; snprintf(tag, sizeof tag, "h-index after peeling levels < %d", L0s[i]);
            hindex_run(tag, c, snap_alive[i], core, mode & 1, mode < 2 ? 0 : 2);
        }
    }
    return 0;
}
