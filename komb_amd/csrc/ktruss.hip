// ktruss.hip -- rows a5 + a6 of the hot-path table: per-edge trussness of the
// (optionally vertex-induced) simple graph, replacing
// igraph_induced_subgraph_map + igraph_trussness (reference src/graph.cpp:502,
// src/graph.cpp:508).  Trussness(e) = 2 + the support level at which e is
// peeled; a triangle-free edge has trussness 2.
//
// MI355X-first design (no intersections inside the peel loop):
//   1. every edge points from its lower to its higher (degree,id) endpoint.  The
//      k-truss side of a graph -- INTERNAL ids = (degree,id) ranks, the oriented
//      CSR in those ids (orow/ocol), the canonical edge map, the enumeration's
//      lines and tasks -- is made by truss_prep.hip when the first k-truss call of
//      a graph needs it (inside the call and its time) and kept with the graph; an
//      induced subgraph is first made a symmetric CSR of its own and then prepared
//      like any graph.  The internal edge id is the oriented slot.  On power-law
//      unitig graphs the oriented rows are tiny (max ~10^2), whatever the hub
//      degrees are.
//   2. enumerate every triangle once and build the incidence index: for every
//      edge, the pairs of the other two edges of its triangles (24 bytes per
//      triangle).  Enumeration by WEDGES (truss_wedge.h): the slots of
//      the LDS-staged row of a behind edge a->b are tested against a 64-byte line
//      of b (pivots + Bloom signature), survivors looked up in N+(b) with one
//      trip to memory.  ONE pass: the entries of a task's own edges leave it as a
//      dense block, every other entry as a 12-byte record of one stream -- no
//      atomic, no scattered store per triangle; the records are radix-sorted by
//      BIN (64-edge chunks dealt round-robin, truss_index.h) and one workgroup per
//      bin assembles its window of the index in LDS (k_bin_count, k_bin_finish),
//      which also writes the slices' (start, length) pairs, the peel's initial
//      state and the first level's frontier.  The exact count-scan-fill two-pass
//      layout over rounds 1-3's probe enumeration (truss_tri.h) is the fallback
//      when the stream does not fit (option INDEX=two_pass forces it).
//   3. peel: level-synchronous sub-rounds driven by the device control block
//      (peel_dev.h).  A frontier edge walks its incidence slice; a triangle
//      whose other two edges are both still present loses one support on each
//      (ties between two frontier edges are broken by edge id so every
//      triangle is destroyed exactly once); the decrement that lands an edge
//      exactly on the level enqueues it for the next sub-round.  The peel
//      never touches the adjacency again.
//   4. gather results into canonical (min,max)-lexicographic edge order with
//      ORIGINAL vertex ids -- the identity the C ABI promises.  No search: the
//      preparation carries the internal edge id of every canonical edge.
#include "peel_dev.h"
#include "truss_tail.h"
#include "local_dev.h"
#include "shard_dev.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "truss_line.h"
#include "truss_tri.h"
#include "truss_wedge.h"
#include "truss_index.h"
#include "truss_gather.h"

namespace komb {

namespace {

// ------------------------------------------------------------------ the peel
// peel_dev.h's engine with: unit = edge, key = live support, slice = the
// edge's incidence slice (pairs of the other two edges of each triangle).
// stamp[e] = an alive marker (common.h) while e is live, else the sub-round in which e is (to be)
// peeled.  For a frontier edge `me` (stamp == round r) and a triangle {me,x,y}:
//   - x or y peeled in an earlier sub-round (stamp < r): the triangle is gone.
//   - otherwise the triangle is destroyed now; each of x,y that is not itself
//     in this frontier loses one support -- by `me` alone if the other edge is
//     live, or by the smaller edge id if two of the three are in the frontier
//     -- so every triangle is destroyed exactly once.
// A decrement that returns level+1 triggers the edge (trussness level+2).
struct TrussProblem {
    static constexpr bool kChain = false;
    static constexpr bool kSingleStep = false;
    uint32_t units;
    const uint2 *off2;                   // (start, length) of every edge's incidence slice
    const int2 *inc;
    int32_t *sup;
    int32_t *stamp;                      // alive marker, then the sub-round the edge was peeled in: its trussness is
                                         // rlevel[stamp] + 2 (PeelQueues::rlevel, written once per sub-round) -- no result store per edge
    uint8_t *st8;                        // [units + 16] the stamps' one-byte shadow the triangle visits gather from (peel_dev.h: state_of_round)
    static constexpr int32_t kRetireEvery = komb::kRetireEvery;   // (marks the problem as one with byte states: peel_dev.h)
    int32_t retire_every;                // sub-rounds between two RETIRE steps: kRetireEvery, or less (KOMB_RETIRE_EVERY, tests)
#ifdef KOMB_DEBUG_SWITCHES
    int ablate;                          // KOMB_PEEL_ABLATE (breaks results on purpose; timing of single steps): 1 no decrements, 2 no state gathers, 4 no index loads
#endif

    // RETIRE step: codes of sub-rounds before the current one become ST_GONE, 16 states per lane per trip
    __device__ __forceinline__ void retire(const CtrlView &cv, uint32_t block, uint32_t nblocks) const
    {
        const uint32_t n16 = (units + 15u) / 16u;           // (the allocation is padded; bytes past `units` are never read as states)
        uint4 *v = reinterpret_cast<uint4 *>(st8);
        for (uint32_t i = block * (uint32_t)kPeelBlock + threadIdx.x; i < n16; i += nblocks * (uint32_t)kPeelBlock) {
            uint4 q = v[i];
            uint32_t *w = &q.x;
            bool changed = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t out = w[k];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t c = (w[k] >> (8 * b)) & 0xFFu;
                    if (c > ST_GONE && state_rel(c, cv.round) == REL_GONE) out = (out & ~(0xFFu << (8 * b))) | ((uint32_t)ST_GONE << (8 * b));
                }
                changed |= out != w[k];
                w[k] = out;
            }
            if (changed) v[i] = q;
        }
    }

    __device__ __forceinline__ const int32_t *scan_marker() const { return stamp; }
    __device__ __forceinline__ const uint8_t *scan_state() const { return st8; }
    __device__ __forceinline__ const int32_t *scan_key() const { return sup; }
    __device__ __forceinline__ void mark_scanned(uint32_t e, const CtrlView &cv) const { stamp[e] = cv.round; st8[e] = state_of_round(cv.round); }
    __device__ __forceinline__ void slice(uint32_t e, uint32_t &b, uint32_t &len) const
    {
        const uint2 o = off2[e];
        b = o.x;
        len = o.y;
    }
    struct Loaded { int32_t me, x, y; uint32_t cx, cy; };
    __device__ __forceinline__ Loaded item_load(int32_t me, uint32_t pos, const CtrlView &) const
    {
        Loaded ld;
        // (an index entry is read once per visit and never again soon: the non-temporal hint keeps it from displacing the state
        // and support lines the gathers and atomics reuse -- same box: peel 9.2 -> 9.0 ms; the same hint on the stamp gathers: 10.2)
#ifdef KOMB_DEBUG_SWITCHES
        if (ablate & 6) {
            int2 p = make_int2((int)(((uint32_t)me * 2654435761u) % units), (int)(((uint32_t)me * 40503u + pos) % units));
            if (!(ablate & 4)) { const unsigned long long pq = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(inc) + pos); p = make_int2((int)(uint32_t)pq, (int)(uint32_t)(pq >> 32)); }
            ld.me = me; ld.x = p.x; ld.y = p.y;
            ld.cx = (ablate & 2) ? 0u : st8[p.x]; ld.cy = (ablate & 2) ? 0u : st8[p.y];
            return ld;
        }
#endif
        const unsigned long long pq = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(inc) + pos);
        const int2 p = make_int2((int)(uint32_t)pq, (int)(uint32_t)(pq >> 32));
        ld.me = me; ld.x = p.x; ld.y = p.y;
        ld.cx = st8[p.x]; ld.cy = st8[p.y];
        return ld;
    }
    __device__ __forceinline__ void item_apply(const Loaded &ld, const CtrlView &cv, int32_t &t0, int32_t &t1, uint32_t &c0, uint32_t &c1) const
    {
        const int32_t r = cv.round, L = cv.level;
        const int rx = state_rel(ld.cx, r), ry = state_rel(ld.cy, r);
        if (rx == REL_GONE || ry == REL_GONE) return;   // an edge of the triangle is already gone
        const bool xin = (rx == REL_NOW), yin = (ry == REL_NOW);
        bool decx = !xin && (!yin || ld.me < ld.y);
        bool decy = !yin && (!xin || ld.me < ld.x);
#ifdef KOMB_DEBUG_SWITCHES
        if (ablate & 1) { decx = false; decy = false; }
#endif
        // (a triggered heavy edge's chunk count is in the alive marker its stamp still holds)
        if (decx && atomicSub(&sup[ld.x], 1) == L + 1) {
            c0 = ld.cx == ST_ALIVE_HEAVY ? marker_chunks(stamp[ld.x]) : 0u;
            stamp[ld.x] = r + 1; st8[ld.x] = state_of_round(r + 1); t0 = ld.x;
        }
        if (decy && atomicSub(&sup[ld.y], 1) == L + 1) {
            c1 = ld.cy == ST_ALIVE_HEAVY ? marker_chunks(stamp[ld.y]) : 0u;
            stamp[ld.y] = r + 1; st8[ld.y] = state_of_round(r + 1); t1 = ld.y;
        }
    }
};

// The same peel with the supports owned by edge range (shard_dev.h): every rank walks every frontier edge's triangles and
// makes the same decisions from the replicated stamps; a decrement is applied -- and can trigger -- only on the rank that
// owns its target.  (The owner stamps a triggered edge at once, the other ranks when the next frontier arrives: both
// values are "later than this sub-round" to everything that reads them in between.)
struct ShardTruss {
    static constexpr bool kChain = false;
    static constexpr bool kSingleStep = true;
    uint32_t units;
    const uint2 *off2;
    const int2 *inc;
    int32_t *sup;
    int32_t *stamp;
    uint32_t lo, hi;                     // internal edge ids this rank owns

    __device__ __forceinline__ const int32_t *scan_marker() const { return stamp; }
    __device__ __forceinline__ const int32_t *scan_key() const { return sup; }
    __device__ __forceinline__ void mark_scanned(uint32_t e, const CtrlView &cv) const { stamp[e] = cv.round; }
    __device__ __forceinline__ void slice(uint32_t e, uint32_t &b, uint32_t &len) const
    {
        const uint2 o = off2[e];
        b = o.x;
        len = o.y;
    }
    __device__ __forceinline__ bool mine(int32_t e) const { return (uint32_t)e - lo < hi - lo; }
    struct Loaded { int32_t me, x, y, sx, sy; };
    __device__ __forceinline__ Loaded item_load(int32_t me, uint32_t pos, const CtrlView &) const
    {
        Loaded ld;
        const int2 p = inc[pos];
        ld.me = me; ld.x = p.x; ld.y = p.y;
        ld.sx = stamp[p.x]; ld.sy = stamp[p.y];
        return ld;
    }
    __device__ __forceinline__ void item_apply(const Loaded &ld, const CtrlView &cv, int32_t &t0, int32_t &t1, uint32_t &, uint32_t &) const
    {
        const int32_t r = cv.round, L = cv.level;
        if (ld.sx < r || ld.sy < r) return;             // an edge of the triangle is already gone
        const bool xin = (ld.sx == r), yin = (ld.sy == r);
        const bool decx = !xin && (!yin || ld.me < ld.y) && mine(ld.x);
        const bool decy = !yin && (!xin || ld.me < ld.x) && mine(ld.y);
        // (triggered edges are reported as light units: the next frontier is re-classified after the exchange)
        if (decx && atomicSub(&sup[ld.x], 1) == L + 1) { stamp[ld.x] = r + 1; t0 = ld.x; }
        if (decy && atomicSub(&sup[ld.y], 1) == L + 1) { stamp[ld.y] = r + 1; t1 = ld.y; }
    }
};

// ---- hand-over to the local finish (local_dev.h)
// Collect pass: the peel engine run once over all live edges; a triangle whose other two edges are both live
// becomes an entry of the compact slice (ids of the remainder).
struct TrussCollect {
    static constexpr bool kChain = false;
    static constexpr bool kSingleStep = false;
    uint32_t units;
    const uint2 *off2;                   // the peel's incidence index
    const int2 *inc;
    const int32_t *stamp;                // alive markers
    const int32_t *num;                  // [m] id in the remainder (live edges only)
    const uint32_t *coff;                // [n+1] compact slice offsets
    uint32_t *cur;                       // [n] fill cursors
    uint2 *cpair;                        // compact slices

    __device__ __forceinline__ const int32_t *scan_marker() const { return stamp; }
    __device__ __forceinline__ const int32_t *scan_key() const { return nullptr; }   // every live edge enters the one frontier of this pass
    __device__ __forceinline__ void mark_scanned(uint32_t, const CtrlView &) const {}
    __device__ __forceinline__ void slice(uint32_t e, uint32_t &b, uint32_t &len) const
    {
        const uint2 o = off2[e];
        b = o.x;
        len = o.y;
    }
    struct Loaded { int32_t me, x, y, sx, sy; };
    __device__ __forceinline__ Loaded item_load(int32_t me, uint32_t pos, const CtrlView &) const
    {
        Loaded ld;
        const int2 p = inc[pos];
        ld.me = me; ld.x = p.x; ld.y = p.y;
        ld.sx = stamp[p.x]; ld.sy = stamp[p.y];
        return ld;
    }
    __device__ __forceinline__ void item_apply(const Loaded &ld, const CtrlView &, int32_t &, int32_t &, uint32_t &, uint32_t &) const
    {
        if (!marker_alive(ld.sx) || !marker_alive(ld.sy)) return;
        const uint32_t id = (uint32_t)num[ld.me];
        // (bounded: a unit whose live items outnumber its live key -- an inconsistent index -- is reported by k_local_check
        // from its cursor, never written past its slice)
        const uint32_t b0 = coff[id], k = local_slot(id, cur);
        if (k < coff[id + 1] - b0) cpair[b0 + k] = make_uint2((uint32_t)num[ld.x], (uint32_t)num[ld.y]);
    }
};

// Fixed-point problem: item = a live triangle, value = the smaller bound of its other two edges.
// remainders with more live triangle entries than this stay with the peel (see local_item_limit)
constexpr uint64_t kTrussLocalItems = 32ull << 20;
// ... and so do remainders with more live triangles per edge than this (measured: 24 at C3 and 105 on the alpha = 2.3 shape
// gain 2 and 9 ms, 280 on the alpha = 2.1 shape loses 4)
constexpr uint32_t kTrussLocalDensity = 160;
struct TrussLocal {
    static constexpr int kU = 8;         // light unit: <= 512 live triangles (one batch = 64 lanes x 8 values)
#ifndef KOMB_TRUSS_GROUPS
#define KOMB_TRUSS_GROUPS 8
#endif
    static constexpr int kGroups = KOMB_TRUSS_GROUPS;
    static constexpr int kN = 2;
    const uint2 *cpair;
    __device__ __forceinline__ void ids(uint32_t pos, uint32_t (&id)[2]) const
    {
        const uint2 p = cpair[pos];
        id[0] = p.x; id[1] = p.y;
    }
};

} // namespace

void truss_free(komb_ctx *ctx)
{
    if (ctx->t_own_edges) { ctx->pool.put(ctx->d_t_eu); ctx->pool.put(ctx->d_t_ev); }
    ctx->pool.put(ctx->d_t_truss);
    ctx->pool.put(ctx->d_t_sup);
    ctx->pool.put(ctx->d_t_slice);
    ctx->d_t_eu = ctx->d_t_ev = ctx->d_t_truss = ctx->d_t_sup = nullptr;
    ctx->d_t_slice = nullptr;
    ctx->t_sup_ready = false;
    ctx->t_own_edges = false;
    ctx->t_ne = -1; ctx->truss_done = false;
}

// the canonical edge list of the resident graph (what igraph_edge answers after igraph_trussness, reference src/graph.cpp:529-532):
// made by the first fetch that asks for endpoints, kept with the graph
int truss_edges_canonical(komb_ctx *ctx)
{
    if (ctx->t_own_edges || ctx->t_ne <= 0) return KOMB_OK;         // an induced subgraph's result carries its own list
    if (!ctx->d_ceu) {
        int32_t *eu = nullptr, *ev = nullptr;                       // (published only when both exist and are filled)
        hipError_t e = ctx->pool.get((void **)&eu, (size_t)ctx->ne * sizeof(int32_t));
        if (e == hipSuccess) e = ctx->pool.get((void **)&ev, (size_t)ctx->ne * sizeof(int32_t));
        const int rc = e == hipSuccess ? edge_list(ctx, ctx->d_o_rowptr, ctx->d_o_col, ctx->nv, nullptr, eu, ev) : KOMB_ERR_NOMEM;
        if (rc != KOMB_OK) {
            ctx->pool.put(eu); ctx->pool.put(ev);
            if (e != hipSuccess) KOMB_FAIL(ctx, KOMB_ERR_NOMEM, "komb_truss_fetch: no memory for the canonical edge list (%s)", hipGetErrorString(e));
            return rc;
        }
        ctx->d_ceu = eu; ctx->d_cev = ev;
    }
    ctx->d_t_eu = ctx->d_ceu; ctx->d_t_ev = ctx->d_cev;
    return KOMB_OK;
}

// the supports the last whole-graph run started from, in canonical order (komb_truss_fetch_support's first call after a run)
int truss_support_canonical(komb_ctx *ctx)
{
    if (ctx->t_sup_ready) return KOMB_OK;
    if (!ctx->truss_done || !ctx->d_t_slice || !ctx->prep.valid || ctx->prep.ne != ctx->t_ne)
        KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_truss_fetch_support: no k-truss result to take the supports from");
    hipStream_t s = ctx->stream;
    const int64_t m = ctx->t_ne;
    KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_sup, (size_t)m * sizeof(int32_t)));
    if (ctx->t_k_lo) KOMB_HIP(ctx, hipMemsetAsync(ctx->d_t_sup, 0, (size_t)ctx->t_k_lo * sizeof(int32_t), s));
    if (ctx->t_k_hi < (uint32_t)m) KOMB_HIP(ctx, hipMemsetAsync(ctx->d_t_sup + ctx->t_k_hi, 0, ((size_t)m - ctx->t_k_hi) * sizeof(int32_t), s));
    k_scatter_len<<<grid_for(m), kBlock, 0, s>>>(ctx->d_t_slice, ctx->prep.e2k, m, ctx->t_k_lo, ctx->t_k_hi, ctx->d_t_sup);
    KOMB_HIP(ctx, hipStreamSynchronize(s));
    ctx->t_sup_ready = true;
    return KOMB_OK;
}

// rank/world/fn: support counting is sharded by source-vertex range; fn sums the
// partial support vectors over the ranks (RCCL all-reduce on the host side).
int truss_run(komb_ctx *ctx, const uint8_t *vmask_host, int rank, int world, komb_allreduce_fn fn, void *user)
{
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !fn))
        KOMB_FAIL(ctx, KOMB_ERR_ARG, "komb_truss_run_sharded: bad rank %d / world %d / callback", rank, world);
    if (ctx->nv < 0) KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_truss_run: no graph loaded");
    truss_free(ctx);
    hipStream_t s = ctx->stream;
    komb_stats &st = ctx->stats;
    st.triangles = 0; st.truss_levels = st.truss_subrounds = st.truss_launches = 0;
    st.max_trussness = 0; st.ms_support = st.ms_peel = st.ms_orient = st.ms_tri_count = st.ms_tri_fill = st.ms_gather = st.ms_allreduce = st.ms_compact = 0.0;
    st.truss_scans = 0; st.ms_prepare = 0.0; st.truss_prepared = 0; st.ms_prep_vertex = st.ms_prep_edges = st.ms_prep_rows = 0.0;
    auto empty_result = [&]() -> int {
        KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_eu, 4)); KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_ev, 4));
        KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_truss, 4)); KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_sup, 4));
        ctx->t_own_edges = true; ctx->t_ne = 0; ctx->truss_done = true; ctx->t_sup_ready = true;
        return KOMB_OK;
    };
    if (ctx->nv == 0 || ctx->ne == 0) return empty_result();
    Range r_all("komb_truss_run");
    DevBufs bufs(ctx);

    // ---- a5 + the k-truss side of the graph (truss_prep.hip).  The whole graph's is made on its first k-truss call and kept;
    // a subgraph induced by vmask becomes a symmetric CSR of its own (new ids = ranks among the kept vertices: monotone, so
    // its canonical edge order is the whole graph's restricted) and gets a temporary preparation like any graph's.
    Range phase("truss: preparation");
    InducedCsr sub;
    TrussPrep sub_prep;
    struct SubGuard { komb_ctx *c; InducedCsr *g; TrussPrep *p; ~SubGuard() { prep_free(c, p); induced_free(c, g); } } sub_guard{ctx, &sub, &sub_prep};
    TrussPrep *tp = &ctx->prep;
    if (vmask_host) {
        ctx->timer.start(s);
        KOMB_TRY(induce_csr(ctx, vmask_host, &sub));
        st.ms_orient = ctx->timer.stop(s);
        if (sub.ns == 0) return empty_result();
        KOMB_TRY(prep_build(ctx, sub.rowptr, sub.col, sub.nv, sub.ns, &sub_prep));
        st.ms_prepare = sub_prep.ms; st.truss_prepared = 1;
        st.ms_prep_vertex = sub_prep.ms_part[0]; st.ms_prep_edges = sub_prep.ms_part[1]; st.ms_prep_rows = sub_prep.ms_part[2];
        tp = &sub_prep;
    } else if (!ctx->prep.valid) {
        KOMB_TRY(prep_ensure(ctx));
        st.ms_prepare = ctx->prep.ms; st.truss_prepared = 1;
        st.ms_prep_vertex = ctx->prep.ms_part[0]; st.ms_prep_edges = ctx->prep.ms_part[1]; st.ms_prep_rows = ctx->prep.ms_part[2];
    }
    const int64_t nv = tp->nv;
    const int64_t m = tp->ne;
    const uint32_t *d_orow = tp->orow;
    const int32_t *d_ocol = tp->ocol;
    const uint4 *d_line = tp->vline;
    const uint2 *d_wtasks = (const uint2 *)tp->wtasks;
    const int64_t n_wtasks = tp->n_wtasks;
    const int gw = grid_for(n_wtasks, kTriWaves);
    phase.next("truss: triangles + incidence index");

    // ---- triangle support + incidence index
    const int ge = grid_for(m);
    // vertices per task of the two-pass fallback's enumeration: kTriV, fewer when that leaves the chip without enough tasks (>= 4 per resident wavefront)
    const int tri_tv = (int)std::max<int64_t>(1, std::min<int64_t>(kTriV, nv / (256 * KOMB_TRI_EU * kTriWaves * 4)));
    const int64_t ntasks = (nv + tri_tv - 1) / tri_tv;
    const int gt = grid_for(ntasks, kTriWaves);
    uint32_t *d_own = nullptr, *d_other = nullptr, *d_cnt = nullptr, *d_off = nullptr;
    uint2 *d_off2 = nullptr;                                         // (start, length) of every edge's slice of the index: what the peel and the gather read
    KOMB_HIP(ctx, bufs.alloc(&d_own, 2 * ((size_t)m + 1)));         // [own | other] contiguous: one all-reduce
    d_other = d_own + ((size_t)m + 1);
    KOMB_HIP(ctx, bufs.alloc(&d_cnt, (size_t)m + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_off2, (size_t)m + 4));
    unsigned long long *d_mom = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_mom, 12));                           // [5] sum of supports, [8] sharded-check mismatches
    KOMB_HIP(ctx, hipMemsetAsync(d_mom, 0, 12 * sizeof(unsigned long long), s));
    st.ms_allreduce = 0.0;

    // Two layouts of the index build:
    //   stream   (default) ONE enumeration by wedges; own-role entries leave a task as dense blocks, everything else as records
    //            of one stream that is then sorted by destination bin and merged with the blocks -- no atomic and no scattered
    //            store per triangle, no slices sized by a bound.
    //   two_pass count, scan, second enumeration into exact slices (the probe enumeration of truss_tri.h): the fallback when
    //            the stream does not fit in memory or runs out; option INDEX=two_pass forces it.
    // A sharded run (world > 1) first counts the supports of its own range of the task table and sums them over the ranks
    // (fn: the RCCL all-reduce), then builds the index whole with the layout above; the summed supports must equal the
    // supports the build finds.
    // how the peel ends (common.h): local fixed point (default), LDS tail (truss_tail.h), or the general engine alone
    const FinishMode fin = finish_mode(ctx, FIN_LOCAL);
    if (fin == FIN_LDS) KOMB_TRY(prep_sources(ctx, tp));            // (the LDS tail reads every edge's source)
    const int32_t *d_osrc = tp->osrc;
    uint32_t tail_limit = 0;
    if (fin == FIN_LDS) {
        tail_limit = kTailEdges;
        if (const char *tl = ctx_opt(ctx, "TAIL")) tail_limit = (uint32_t)strtoul(tl, nullptr, 10);
        if (tail_limit > kTailMaxEdges) tail_limit = kTailMaxEdges;
    } else if (fin == FIN_LOCAL) tail_limit = local_limit(ctx, (uint64_t)m, 32);
    // (a graph small enough for the finish to take the whole peel is handed over before any step: no frontier may be queued)
    const bool whole_peel_finish = tail_limit && (uint64_t)m <= tail_limit;
    // the peel sharded by edge range, one exchange per sub-round (shard_dev.h): komb_set_shard_peel; option SHARD_ENGINE=1 runs
    // the same engine with one rank and no collective -- a test of its logic
    const bool shard_peel_on = (ctx->shard_peel && world > 1) || ctx_flag(ctx, "SHARD_ENGINE");
    enum { IDX_STREAM = 0, IDX_TWO_PASS = 2 };
    int layout = IDX_STREAM;
    if (const char *ix = ctx_opt(ctx, "INDEX")) {
        if (!strcmp(ix, "two_pass")) layout = IDX_TWO_PASS;
        else if (strcmp(ix, "stream")) KOMB_FAIL(ctx, KOMB_ERR_ARG, "option INDEX=%s: expected stream or two_pass", ix);
    }
    int2 *d_owndense = nullptr;                // own-role entries as compact per-task blocks (truss_wedge.h)
    unsigned long long *d_ownoff = nullptr, *d_dcur = nullptr;
    uint32_t *d_cnt_ref = nullptr;             // world > 1: the all-reduced supports, kept to check the build against
    uint32_t *d_toff = nullptr;                // stream: first sorted record of every bin
    uint32_t *d_grp = nullptr;                 // the peel's ticket / init words (stream: allocated before the build's last kernels, which fill them)
    int32_t *d_light0 = nullptr;               // stream: light queue 0 of the peel, which k_bin_finish fills with the first frontier
    uint32_t *d_bintot = nullptr;              // stream: supports summed per bin, then their exclusive scan (every bin's window of the index)
    uint32_t *d_reckey = nullptr;              // stream: the sorted records' keys
    int2 *d_recval = nullptr;                  // stream: ... and values
    const BinGeom geom = bin_geom(m);                               // stream: the bins of the index build (truss_index.h)
    const int64_t n_bins = (int64_t)geom.nb;
    const TriStream no_stream{nullptr, nullptr, nullptr, 0ull, 0u, 0u, 0};
#ifdef KOMB_DEBUG_SWITCHES
    const int ablate = getenv("KOMB_TRI_ABLATE") ? atoi(getenv("KOMB_TRI_ABLATE")) : 0;    // breaks results on purpose: debug builds only
#else
    const int ablate = 0;
#endif
    st.ms_sort = 0.0; st.tri_records = 0; st.ms_compact = 0.0; st.ms_tri_count = 0.0; st.ms_tri_fill = 0.0; st.stream_retries = 0;
    st.index_layout = layout;
    auto zero_counts = [&]() -> hipError_t { return hipMemsetAsync(d_own, 0, 2 * ((size_t)m + 1) * sizeof(uint32_t), s); };
    // (the wedge enumeration writes own[] of every edge it owns, and the stream build on one GPU never touches the third-role
    // counters: no fill at all)
    if (!(world == 1 && layout == IDX_STREAM)) KOMB_HIP(ctx, zero_counts());
    bool have_counts = false;                  // d_cnt holds the supports (and d_mom[5] their sum)
    if (world > 1) {
        const int64_t task_lo = n_wtasks * rank / world, task_hi = n_wtasks * (rank + 1) / world;
        ctx->timer.start(s);
        k_wedges<TRI_COUNT><<<gw, kBlock, 0, s>>>(d_orow, d_ocol, d_line, d_wtasks, task_lo, task_hi, d_own, d_other, nullptr, nullptr, 0ull, nullptr, no_stream, nullptr, ablate);
        st.ms_tri_count = ctx->timer.stop(s);
        k_sum_counts<<<ge, kBlock, 0, s>>>(d_own, d_other, nullptr, m + 1, d_cnt, d_mom + 5);
        // sum the partial support vectors over the ranks (|E|+1 int32), then recompute the 64-bit total
        ctx->timer.start(s);
        KOMB_HIP(ctx, hipStreamSynchronize(s));          // the buffer is complete when the callback runs
        if (fn(user, d_cnt, (int64_t)m + 1) != 0)
            KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "komb_truss_run_sharded: all-reduce callback failed");
        st.ms_allreduce = ctx->timer.stop(s);
        KOMB_HIP(ctx, hipMemsetAsync(d_mom + 5, 0, sizeof(unsigned long long), s));
        k_total_u32<<<ge, kBlock, 0, s>>>(d_cnt, m + 1, d_mom + 5);
        have_counts = true;
        if (layout != IDX_TWO_PASS) {
            KOMB_HIP(ctx, bufs.alloc(&d_cnt_ref, (size_t)m + 1));
            KOMB_HIP(ctx, hipMemcpyAsync(d_cnt_ref, d_cnt, ((size_t)m + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
            KOMB_HIP(ctx, zero_counts());
        }
    }

    if (layout == IDX_STREAM) {
        // capacities: T <= sum_a C(d+(a), 2) triangles, at most three records each (a triangle of a task without a dense
        // block), plus what the chunked claims leave unused; with less memory than that, a stream that runs out falls back
        // (sum over the vertices of d+ (d+ - 1): it comes with the preparation)
        const unsigned long long bound = (unsigned long long)tp->own_bound;
        const int gws = std::min(gw, 256 * KOMB_WEDGE_EU);           // resident workgroups only: every wavefront ends with one partly used claim
        const unsigned long long slack = (unsigned long long)gws * kTriWaves * kRecChunk + kRecChunk;
        const unsigned long long own_slack = (unsigned long long)gws * kTriWaves * kOwnChunk + kOwnChunk;
        const unsigned long long t_bound = bound / 2;
        // the bounds: what NO graph with this preparation can exceed ...
        const unsigned long long own_full = bound + bound / 8 + own_slack, rec_full = 3 * t_bound + (3 * t_bound) / 14 + slack;
        // ... and what the FIRST attempt reserves: room for two triangles per edge (a unitig graph has about one: C3 0.88, and the
        // bounds are 7 x / 20 x what it touches -- 34 GB whose first allocation costs a one-shot komb2 run up to a second).  A graph
        // with more finds the region / the stream too small -- the claim cursors keep counting past the capacities, so the
        // need is then known -- and the enumeration runs ONCE more with exactly that; only if that does not fit either, or
        // the memory is not there, does the build fall back to the two-pass layout.
        unsigned long long own_cap = std::min(own_full, 4ull * (unsigned long long)m + own_slack);
        unsigned long long rec_cap = std::min(rec_full, 2ull * (unsigned long long)m + slack);
        bool fixed_caps = false;                                     // (tests: capacities that run out on purpose are not grown)
        if (const char *oc = ctx_opt(ctx, "OWN_DENSE_CAP")) { own_cap = strtoull(oc, nullptr, 10) + 1; rec_cap = rec_full; fixed_caps = true; }
        if (ctx_flag(ctx, "NO_OWN_DENSE")) { own_cap = 1; rec_cap = rec_full; fixed_caps = true; }      // (tests: every entry a record)
        if (const char *rc = ctx_opt(ctx, "REC_CAP")) { rec_cap = strtoull(rc, nullptr, 10) + 1; fixed_caps = true; }
        if (ctx_flag(ctx, "FULL_CAPS")) { own_cap = own_full; rec_cap = rec_full; }                     // (the bounds at once, as rounds 3-4 did)
        if (!fixed_caps && ctx->cap_hint.nv == nv && ctx->cap_hint.m == m) {                            // what this graph needed last time
            own_cap = std::min(own_full, std::max(own_cap, ctx->cap_hint.own_cap));
            rec_cap = std::min(rec_full, std::max(rec_cap, ctx->cap_hint.rec_cap));
        }
        // keys of unused record positions: above every edge id, their bin field taken from the position (they spread over the bins)
        const uint32_t sentinel = geom.sentinel_base();
        uint32_t *d_key = nullptr; int2 *d_val = nullptr;
        unsigned long long n_claimed = 0;
        bool ok = false;
        st.ms_tri_fill = 0.0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            // (what the pool holds unused counts as free: the choice must not depend on what an earlier call left cached)
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            const unsigned long long budget = (unsigned long long)((free_b + ctx->pool.unused_bytes()) * 0.7);
            if (own_cap * sizeof(int2) > budget / 2) own_cap = budget / 2 / sizeof(int2);
            if (rec_cap * 12ull > budget / 2) rec_cap = budget / 2 / 12ull;
            if (rec_cap > 0xFFFFFFF0ull) rec_cap = 0xFFFFFFF0ull;    // 32-bit record positions
            ok = bufs.alloc(&d_key, (size_t)rec_cap) == hipSuccess && bufs.alloc(&d_val, (size_t)rec_cap) == hipSuccess &&
                 bufs.alloc(&d_owndense, (size_t)own_cap) == hipSuccess &&
                 (d_ownoff || bufs.alloc(&d_ownoff, (size_t)m + 1) == hipSuccess) && (d_dcur || bufs.alloc(&d_dcur, 4) == hipSuccess);
            // the wedge enumeration's wave-private record scratch (truss_wedge.h); without the memory for it a sub-range whose
            // records outgrow the LDS buffer gives up its dense block
            uint2 *d_scratch = nullptr;
            if (ok && !ctx_flag(ctx, "NO_REC_SCRATCH") && bufs.alloc(&d_scratch, (size_t)gws * kTriWaves * kScratchRec) != hipSuccess) { (void)hipGetLastError(); d_scratch = nullptr; }
            if (!ok) { (void)hipGetLastError(); break; }
            KOMB_HIP(ctx, hipMemsetAsync(d_dcur, 0, 4 * sizeof(unsigned long long), s));
            const TriStream ts{d_key, d_val, d_dcur + 2, rec_cap, sentinel, geom.nb - 1u, kChunkBits};
            ctx->timer.start(s);
            k_wedges<TRI_SINGLE><<<gws, kBlock, 0, s>>>(d_orow, d_ocol, d_line, d_wtasks, 0, n_wtasks, d_own, d_other, d_owndense, d_dcur, own_cap, d_ownoff, ts, d_scratch, ablate);
            st.ms_tri_fill += ctx->timer.stop(s);
            unsigned long long dc[4] = {0, 0, 0, 0};
            KOMB_HIP(ctx, d2h(ctx, dc, d_dcur, sizeof(dc)));
            n_claimed = dc[2];
            bufs.release(d_scratch);
            if (ctx_flag(ctx, "TRI_DEBUG"))
                fprintf(stderr, "komb triangles: stream build (attempt %d): %llu record positions claimed of %llu, dense own-role region %llu entries claimed of %llu, %llu task ranges overflowed their record buffer\n",
                        attempt, dc[2], rec_cap, dc[0], own_cap, dc[1]);
            if (n_claimed <= rec_cap) {                              // every record is there (a dense region that ran out only made more of them)
                if (!fixed_caps) { ctx->cap_hint.nv = nv; ctx->cap_hint.m = m; ctx->cap_hint.own_cap = own_cap; ctx->cap_hint.rec_cap = rec_cap; }
                break;
            }
            ok = false;                                              // the stream ran out: records were dropped
            if (attempt == 1 || fixed_caps) break;
            // once more, with what this attempt asked for: the dense region it claimed (all of it this time, so fewer records than
            // it claimed now) and the records it claimed
            bufs.release(d_key); bufs.release(d_val); bufs.release(d_owndense);
            d_key = nullptr; d_val = nullptr; d_owndense = nullptr;
            own_cap = std::min(own_full, std::max(own_cap, dc[0] + own_slack));
            rec_cap = std::min(rec_full, n_claimed + slack);
            st.stream_retries += 1;
        }
        uint32_t *d_skey = nullptr;
        if (ok) {
            // sort the records by destination bin
            uint32_t *d_key2 = nullptr; unsigned long long *d_val2 = nullptr;
            ok = bufs.alloc(&d_key2, (size_t)n_claimed + 1) == hipSuccess && bufs.alloc(&d_val2, (size_t)n_claimed + 1) == hipSuccess;
            if (ok) {
                // sort the records by BIN (the bin field of the key only): two radix passes instead of four
                ctx->timer.start(s);
                unsigned long long *sv = nullptr;
                if (geom.nb_bits > 0) KOMB_TRY(prim_sort_pairs_u32_u64(ctx, d_key, d_key2, (unsigned long long *)d_val, d_val2, (int64_t)n_claimed, kChunkBits, kChunkBits + geom.nb_bits, &d_skey, &sv));
                else { d_skey = d_key; sv = (unsigned long long *)d_val; }      // a single bin: nothing to sort
                st.ms_sort = ctx->timer.stop(s);
                d_recval = (int2 *)sv;
                if (d_skey == d_key) { bufs.release(d_key2); bufs.release(d_val2); } else { bufs.release(d_key); bufs.release(d_val); }
                d_reckey = d_skey;
                KOMB_HIP(ctx, bufs.alloc(&d_toff, (size_t)n_bins + 2));
                KOMB_HIP(ctx, bufs.alloc(&d_bintot, (size_t)n_bins + 2));
                KOMB_HIP(ctx, bufs.alloc(&d_grp, (size_t)kInitOff + 4));
                if (!whole_peel_finish && !shard_peel_on && !ctx_flag(ctx, "NO_FIRST_QUEUE")) KOMB_HIP(ctx, bufs.alloc(&d_light0, (size_t)m));
                ctx->timer.start(s);
                peel_ctrl_pre(s, d_grp);
                KOMB_HIP(ctx, hipMemsetAsync(d_mom + 5, 0, sizeof(unsigned long long), s));
                k_bin_offsets<<<grid_for(n_bins + 1), kBlock, 0, s>>>(d_skey, (int64_t)n_claimed, geom, d_toff);
                k_bin_count<<<grid_for(n_bins, 1, 256 * 8), kBlock, 0, s>>>(d_skey, d_toff, geom, d_own, m, d_cnt, d_bintot, d_mom + 5, (int32_t *)(d_grp + kInitOff + 1));
                st.ms_compact = ctx->timer.stop(s);
                st.tri_records = (int64_t)n_claimed;
            } else (void)hipGetLastError();
        }
        if (!ok) {
            // no memory for the stream, or it ran out: exact two-pass build
            bufs.release(d_key); bufs.release(d_val); bufs.release(d_owndense); bufs.release(d_ownoff); bufs.release(d_dcur);
            bufs.release(d_grp); bufs.release(d_light0); d_grp = nullptr; d_light0 = nullptr;
            d_owndense = nullptr; d_ownoff = nullptr; d_dcur = nullptr;
            KOMB_HIP(ctx, zero_counts());
            KOMB_HIP(ctx, hipMemsetAsync(d_mom + 5, 0, sizeof(unsigned long long), s));
            st.ms_tri_fill = 0.0;
            layout = IDX_TWO_PASS;
        }
    }
    if (layout != IDX_TWO_PASS && d_cnt_ref) {
        // sharded run: the supports summed over the ranks must be the supports the whole build has just found
        k_count_mismatch<<<ge, kBlock, 0, s>>>(d_cnt, d_cnt_ref, m + 1, d_mom + 8);
    }
    if (layout == IDX_TWO_PASS) {
        if (d_cnt_ref) {
            KOMB_HIP(ctx, hipMemcpyAsync(d_cnt, d_cnt_ref, ((size_t)m + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
            KOMB_HIP(ctx, hipMemsetAsync(d_mom + 5, 0, sizeof(unsigned long long), s));
            k_total_u32<<<ge, kBlock, 0, s>>>(d_cnt, m + 1, d_mom + 5);
        } else if (!have_counts) {
            ctx->timer.start(s);
            k_triangles<TRI_COUNT><<<gt, kBlock, 0, s>>>(d_orow, d_ocol, nv, 0, ntasks, d_own, d_other, nullptr, nullptr, ablate, tri_tv);
            st.ms_tri_count = ctx->timer.stop(s);
            k_sum_counts<<<ge, kBlock, 0, s>>>(d_own, d_other, nullptr, m + 1, d_cnt, d_mom + 5);
        }
    }
    st.index_layout = layout;
    {
        unsigned long long mom[9];
        KOMB_HIP(ctx, d2h(ctx, mom, d_mom, sizeof(mom)));
        st.triangles = (int64_t)(mom[5] / 3);
        bufs.release(d_mom);
        if (mom[8] != 0)
            KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "komb_truss_run_sharded: %llu edges whose all-reduced support differs from the support of the index build", mom[8]);
        if (mom[5] > 0xFFFFFFF0ull)
            KOMB_FAIL(ctx, KOMB_ERR_LIMIT, "graph has %llu triangles; the incidence index is limited to 2^32-16 entries (3 per triangle)",
                      mom[5] / 3);
    }
    bufs.release(d_cnt_ref);
    uint32_t total = 0;
    const uint32_t sum_supports = (uint32_t)st.triangles * 3u;      // (checked above: below 2^32 - 16) = the index's entries
    int2 *d_inc = nullptr;
    int32_t *d_sup = nullptr, *d_stamp = nullptr, *d_truss = nullptr;
    uint8_t *d_st8 = nullptr;                  // the stamps' one-byte shadow (peel_dev.h: state_of_round)
    bool peel_inited = false;                  // the stream layout's finish also writes the peel's initial state
    if (layout == IDX_STREAM) {
        // every bin's window of the index from the scan of the per-bin totals (the slice offsets themselves are a workgroup scan inside k_bin_finish)
        KOMB_TRY(prim_exclusive_sum_u32(ctx, d_bintot, d_bintot, n_bins + 1));
        total = sum_supports;                                        // (= the scan's last element: no round trip for it)
        KOMB_HIP(ctx, bufs.alloc(&d_inc, (size_t)total));
        KOMB_HIP(ctx, bufs.alloc(&d_sup, (size_t)m));
        KOMB_HIP(ctx, bufs.alloc(&d_stamp, (size_t)m));
        KOMB_HIP(ctx, bufs.alloc(&d_st8, (size_t)m + 16));
        KOMB_HIP(ctx, bufs.alloc(&d_truss, (size_t)m));
        ctx->timer.start(s);
        k_bin_finish<<<grid_for(n_bins, 1, 256 * kFinPerCu), kFinBlock, 0, s>>>(d_reckey, d_recval, d_toff, geom, d_own, d_cnt, d_bintot, d_owndense, d_ownoff, d_inc, m,
                                                                      d_off2, d_sup, d_stamp, d_st8, d_grp + kInitOff, d_light0);
        st.ms_compact += ctx->timer.stop(s);
        peel_inited = true;
        bufs.release(d_toff); bufs.release(d_bintot); bufs.release((void *)d_recval); bufs.release(d_reckey);
        bufs.release(d_owndense); bufs.release(d_ownoff); bufs.release(d_dcur);
    } else {
        // exact two-pass: the slices follow each other in edge order; the (start, length) pairs are made with the peel's initial
        // state; second enumeration, into the EXACT slices: own-role entries from the front and third-role entries from the
        // back meet precisely -- no compaction
        KOMB_HIP(ctx, bufs.alloc(&d_off, (size_t)m + 1));
        KOMB_TRY(prim_exclusive_sum_u32(ctx, d_cnt, d_off, m + 1));
        KOMB_HIP(ctx, d2h(ctx, &total, d_off + m, sizeof(uint32_t)));
        KOMB_HIP(ctx, bufs.alloc(&d_inc, (size_t)total));
        ctx->timer.start(s);
        KOMB_HIP(ctx, zero_counts());
        k_back_cursors<<<ge, kBlock, 0, s>>>(d_off, m, d_other);
        k_triangles<TRI_SINGLE><<<gt, kBlock, 0, s>>>(d_orow, d_ocol, nv, 0, ntasks, d_own, d_other, d_off, d_inc, ablate, tri_tv);
        st.ms_tri_fill = ctx->timer.stop(s);
    }
    st.ms_support = st.ms_tri_count + st.ms_tri_fill + st.ms_sort + st.ms_compact;
    bufs.release(d_cnt); bufs.release(d_own);

    // ---- peel
    phase.next("truss: peel");
    PeelCtrl *d_ctrl = nullptr;
    PeelQueues Q{nullptr, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, 0};
    const size_t heavy_cap = (size_t)total / 32 + 64;             // see kcore.hip
    if (!peel_inited) {
        KOMB_HIP(ctx, bufs.alloc(&d_sup, (size_t)m));
        KOMB_HIP(ctx, bufs.alloc(&d_stamp, (size_t)m));
        KOMB_HIP(ctx, bufs.alloc(&d_st8, (size_t)m + 16));
        KOMB_HIP(ctx, bufs.alloc(&d_truss, (size_t)m));
        KOMB_HIP(ctx, bufs.alloc(&d_grp, (size_t)kInitOff + 4));
    }
    for (int i = 0; i < 2; ++i) {
        if (i == 0 && d_light0) Q.light[0] = d_light0;
        else KOMB_HIP(ctx, bufs.alloc(&Q.light[i], (size_t)m));
        KOMB_HIP(ctx, bufs.alloc(&Q.heavy[i], heavy_cap));
        KOMB_HIP(ctx, bufs.alloc(&Q.live[i], (size_t)m / 2 + 64));
    }
    KOMB_HIP(ctx, bufs.alloc(&d_ctrl, 1));
    KOMB_HIP(ctx, bufs.alloc(&Q.code, (size_t)m));
    // rlevel[r] = the level sub-round r peeled at (one store per PROCESS step, by its finaliser): with the sub-round stamps the
    // peel writes anyway, that IS every peeled edge's trussness -- no result store per edge (0.6 ms of scattered 4-byte stores
    // at C3).  A sub-round peels at least one edge: m + 2 entries; [0] = level 0 (the triangle-free edges' stamp).
    KOMB_HIP(ctx, bufs.alloc(&Q.rlevel, (size_t)m + 2));
    KOMB_HIP(ctx, hipMemsetAsync(Q.rlevel, 0, 2 * sizeof(int32_t), s));
    int32_t retire_every = kRetireEvery;
    if (const char *e = ctx_opt(ctx, "RETIRE_EVERY")) retire_every = std::max(1, std::min((int)kRetireEvery, atoi(e)));   // (tests: RETIRE steps on small graphs)
#ifdef KOMB_DEBUG_SWITCHES
    TrussProblem P{(uint32_t)m, d_off2, d_inc, d_sup, d_stamp, d_st8, retire_every, getenv("KOMB_PEEL_ABLATE") ? atoi(getenv("KOMB_PEEL_ABLATE")) : 0};
#else
    TrussProblem P{(uint32_t)m, d_off2, d_inc, d_sup, d_stamp, d_st8, retire_every};
#endif
    TailBufs T{};
    if (fin == FIN_LDS && tail_limit) {
        KOMB_HIP(ctx, bufs.alloc(&T.vmap, (size_t)nv));
        KOMB_HIP(ctx, bufs.alloc(&T.cnt, 64));
        KOMB_HIP(ctx, bufs.alloc(&T.vlist, (size_t)kTailMaxV));
        KOMB_HIP(ctx, bufs.alloc(&T.rows, (size_t)kTailMaxV * kTailRowWords));
        KOMB_HIP(ctx, bufs.alloc(&T.pair, (size_t)tail_limit));
        KOMB_HIP(ctx, bufs.alloc(&T.sup, (size_t)tail_limit));
        KOMB_HIP(ctx, bufs.alloc(&T.gid, (size_t)tail_limit));
        KOMB_HIP(ctx, bufs.alloc(&T.gid_by_rank, (size_t)tail_limit));
        KOMB_HIP(ctx, bufs.alloc(&T.pair_by_rank, (size_t)tail_limit));
        KOMB_HIP(ctx, bufs.alloc(&T.truss_by_rank, (size_t)tail_limit));
        KOMB_HIP(ctx, bufs.alloc(&T.spill[0], (size_t)tail_limit));
        KOMB_HIP(ctx, bufs.alloc(&T.spill[1], (size_t)tail_limit));
        T.max_edges = tail_limit;
    }
    // the live edges are those of `list` (or all m when list is null) whose stamp is still an alive marker
    auto run_tail = [&](const int32_t *list, uint32_t n_in) -> int {
        KOMB_HIP(ctx, hipMemsetAsync(T.vmap, 0, (size_t)nv * sizeof(int32_t), s));
        KOMB_HIP(ctx, hipMemsetAsync(T.cnt, 0, 64 * sizeof(uint32_t), s));
        KOMB_HIP(ctx, hipMemsetAsync(T.rows, 0, (size_t)kTailMaxV * kTailRowWords * sizeof(unsigned long long), s));
        const int g = grid_for(n_in, kBlock, 256);
        // KOMB_TAIL_DEBUG=1: one line per hand-over on stderr (HIP-event times; building with -DKOMB_TAIL_TIMERS
        // adds the kernel's own per-phase stopwatch)
        const bool dbg = ctx_flag(ctx, "TAIL_DEBUG");
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
        for (auto &e : ev) (void)hipEventCreate(&e);
        (void)hipEventRecord(ev[0], s);
        k_tail_mark<<<g, kBlock, 0, s>>>(list, n_in, d_sup, d_stamp, d_osrc, d_ocol, T);
        k_tail_number<<<1, kTailMaxV, 0, s>>>(T);
        k_tail_rows<<<g, kBlock, 0, s>>>(list, n_in, d_sup, d_stamp, d_osrc, d_ocol, T);
        (void)hipEventRecord(ev[1], s);
        k_truss_tail<<<1, 1024, 0, s>>>(d_ctrl, T, d_truss);
        (void)hipEventRecord(ev[2], s);
        KOMB_HIP(ctx, d2h(ctx, &ctx->h_ctrl[0], d_ctrl, sizeof(PeelCtrl)));
        ++st.truss_tail_runs;
        float m0 = 0, m1 = 0, m2 = 0;
        (void)hipEventElapsedTime(&m0, ctx->timer.a, ev[0]); (void)hipEventElapsedTime(&m1, ev[0], ev[1]); (void)hipEventElapsedTime(&m2, ev[1], ev[2]);
        for (auto &e : ev) (void)hipEventDestroy(e);
        st.ms_tail += (double)m1 + (double)m2;
        if (dbg) {
            uint32_t h[16];
            KOMB_HIP(ctx, d2h(ctx, h, T.cnt, sizeof(h)));
            fprintf(stderr, "komb tail: %u vertices, %u edges, %s; general engine before it %.1f us, setup %.1f us, tail kernel %.1f us\n",
                    h[0], h[1], ctx->h_ctrl[0].done == 1 ? "done" : "refused", m0 * 1000.f, m1 * 1000.f, m2 * 1000.f);
#ifdef KOMB_TAIL_TIMERS
            fprintf(stderr, "komb tail: us in kernel: load %.1f ranks %.1f supports %.1f scan %.1f mark %.1f triangles %.1f retire %.1f results %.1f\n",
                    h[14] / 100.0, h[15] / 100.0, h[8] / 100.0, h[9] / 100.0, h[10] / 100.0, h[11] / 100.0, h[12] / 100.0, h[13] / 100.0);
#endif
        }
        return KOMB_OK;
    };
    const int gp = peel_grid(m);
    // local finish: compact the live sub-index, sweep the h-index operator to its fixed point (local_dev.h)
    auto run_local = [&]() -> int {
        const PeelCtrl hc = ctx->h_ctrl[0];
        EventSet evs;
        hipEvent_t ev[2] = {nullptr, nullptr};
        for (auto &e : ev) KOMB_HIP(ctx, evs.make(&e));
        (void)hipEventRecord(ev[0], s);
        LocalStats ls;
        const int lrc = local_finish(ctx, bufs, hc, d_ctrl, (uint32_t)m, d_stamp, d_sup, Q.live[hc.live_sel],
            (uint32_t)kWave * TrussLocal::kU, sizeof(uint2), local_item_limit(ctx, kTrussLocalItems), local_density_limit(ctx, kTrussLocalDensity), true, 2, d_truss,
            [&](const LocalGraph &lg, const int32_t *num, void *items, PeelCtrl *d_cctrl, int32_t launch) {
                TrussCollect C{(uint32_t)m, d_off2, d_inc, d_stamp, num, lg.off, lg.cur, (uint2 *)items};
                PeelQueues Qc = Q;
                Qc.rlevel = nullptr;                   // (the collect pass has sub-rounds of its own: they are nobody's trussness)
                k_peel_step<TrussCollect><<<gp, kPeelBlock, 0, s>>>(d_cctrl, d_grp, Qc, C, launch);
            },
            [&](const LocalGraph &lg, void *items, uint64_t total_items, LocalCtrl *d_lctrl, uint32_t *d_cnt, int *nl) -> int {
                return local_fixpoint(ctx, d_lctrl, d_cnt, lg, TrussLocal{(const uint2 *)items}, total_items, nl);
            },
            &ls, [](const LocalGraph &) {});
        (void)hipEventRecord(ev[1], s);
        (void)hipEventSynchronize(ev[1]);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
        KOMB_TRY(lrc);
        if (ls.refused) { ctx->h_ctrl[0].done = 0; ctx->h_ctrl[0].tail_limit = ls.new_limit; return KOMB_OK; }
        st.truss_local_units = (int32_t)ls.units; st.truss_local_sweeps = ls.sweeps; st.truss_local_items = (int64_t)ls.items;
        st.ms_truss_local = (double)ms;
        PeelCtrl fin_c = hc;
        fin_c.done = 1; fin_c.remaining = 0;
        fin_c.n_levels += (int32_t)ls.levels;
        fin_c.max_level = ls.max_val > fin_c.max_level ? ls.max_val : fin_c.max_level;
        ctx->h_ctrl[0] = fin_c;
        return KOMB_OK;
    };
    ctx->timer.start(s);
    if (!peel_inited) {
        peel_ctrl_pre(s, d_grp);
        k_peel_init<<<grid_for(m, kBlock, 1024), kBlock, 0, s>>>(m, d_off, d_off2, d_sup, d_stamp, d_st8, d_grp + kInitOff);
    }
    peel_ctrl_init(s, d_ctrl, d_grp, (uint32_t)m, tail_limit);
    int launches = 0, rc = KOMB_OK;
    st.truss_tail_runs = 0; st.ms_tail = 0.0;
    st.truss_local_units = 0; st.truss_local_sweeps = 0; st.truss_local_items = 0; st.ms_truss_local = 0.0;
    st.shard_exchanges = 0; st.ms_exchange = 0.0; st.exchange_words = 0;
    if (shard_peel_on) {
        // supports owned by edge range, the frontier exchanged every sub-round; the remainder goes to the replicated local
        // finish under the same rule as in the replicated peel, after one exchange of the live supports (shard_dev.h)
        uint32_t iw[2] = {0u, 0u};
        KOMB_HIP(ctx, d2h(ctx, iw, d_grp + kInitOff, sizeof(iw)));     // triangle-free edges; the smallest positive support
        ShardTruss SP{(uint32_t)m, d_off2, d_inc, d_sup, d_stamp, 0u, 0u};
        shard_bounds((uint64_t)m, rank, world, &SP.lo, &SP.hi);
        ShardStats ss;
        rc = shard_peel(ctx, bufs, SP, d_sup, (uint32_t)m, iw[0], (int32_t)iw[1], rank, world, fn, user, Q, d_ctrl,
                        [&](int32_t launch) { k_peel_step<ShardTruss><<<gp, kPeelBlock, 0, s>>>(d_ctrl, d_grp, Q, SP, launch); },
                        fin == FIN_LOCAL ? tail_limit : 0u, [&]() -> int { return run_local(); }, &ss);
        st.shard_exchanges = (int32_t)ss.exchanges; st.ms_exchange = ss.ms_exchange; st.exchange_words = ss.words;
        launches = ss.launches;
        PeelCtrl fc{};
        fc.done = rc == KOMB_OK ? 1 : 2; fc.n_levels = ss.levels; fc.n_rounds = ss.rounds; fc.n_scans = ss.scans; fc.max_level = ss.max_level;
        ctx->h_ctrl[0] = fc;
    } else if (whole_peel_finish) {
        // small graph: the finish takes the whole peel (unless it is refused, or nothing is left to peel)
        rc = d2h(ctx, &ctx->h_ctrl[0], d_ctrl, sizeof(PeelCtrl)) == hipSuccess ? KOMB_OK : KOMB_ERR_DEVICE;
        if (rc == KOMB_OK && !ctx->h_ctrl[0].done) rc = (fin == FIN_LOCAL) ? run_local() : run_tail(nullptr, (uint32_t)m);
    } else {
        ctx->h_ctrl[0].done = 0;
    }
    for (int guard = 0; rc == KOMB_OK && ctx->h_ctrl[0].done != 1 && ctx->h_ctrl[0].done != 2 && guard < 64; ++guard) {
        if (ctx->h_ctrl[0].done == 3) {
            const PeelCtrl &c = ctx->h_ctrl[0];
            if (fin == FIN_LOCAL) rc = run_local();
            else rc = c.live_mode ? run_tail(Q.live[c.live_sel], c.live_count) : run_tail(nullptr, (uint32_t)m);
            continue;
        }
        int batch = 0;
        rc = drive_peel(ctx, d_ctrl, m, [&](int32_t launch) {
            k_peel_step<TrussProblem><<<gp, kPeelBlock, 0, s>>>(d_ctrl, d_grp, Q, P, launch);
        }, &batch);
        launches += batch;
    }
    st.ms_peel = ctx->timer.stop(s);
    KOMB_TRY(rc);
#ifdef KOMB_STEP_TIMERS
    {
        const PeelCtrl &c = ctx->h_ctrl[0];
        const double n = c.pad1[6] ? (double)c.pad1[6] : 1.0;
        fprintf(stderr, "komb step timers (block 0, %u small multi-workgroup PROCESS steps), us per step: ctrl %.2f queue+slice %.2f items %.2f flush %.2f barrier %.2f ticket %.2f\n",
                c.pad1[6], c.pad1[0] / n / 100.0, c.pad1[1] / n / 100.0, c.pad1[2] / n / 100.0, c.pad1[3] / n / 100.0, c.pad1[4] / n / 100.0, c.pad1[5] / n / 100.0);
    }
#endif
    if (ctx->h_ctrl[0].done != 1) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "k-truss peel ended in an inconsistent state");
    // the one-byte edge states are exact only while a RETIRE step runs every retire_every sub-rounds: the engine records the
    // longest gap it ever saw (peel_dev.h: finalize_step)
    if (ctx->h_ctrl[0].max_retire_gap > retire_every)
        KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "k-truss peel: %d sub-rounds passed without a RETIRE step (period %d): edge state codes may have wrapped",
                  ctx->h_ctrl[0].max_retire_gap, retire_every);
    st.engine_flags = (layout == IDX_TWO_PASS ? KOMB_ENGINE_TWO_PASS : 0) | (shard_peel_on ? KOMB_ENGINE_SHARD_PEEL : 0) |
                      (st.truss_local_units ? KOMB_ENGINE_LOCAL_FINISH : 0) | (st.truss_tail_runs ? KOMB_ENGINE_LDS_TAIL : 0);
    st.truss_levels = ctx->h_ctrl[0].n_levels;
    st.truss_subrounds = ctx->h_ctrl[0].n_rounds;
    st.truss_launches = launches;
    st.truss_scans = ctx->h_ctrl[0].n_scans;
    st.max_trussness = ctx->h_ctrl[0].max_level + 2;
    for (int i = 0; i < 2; ++i) { bufs.release(Q.light[i]); bufs.release(Q.heavy[i]); bufs.release(Q.live[i]); }
    bufs.release(Q.code);
    bufs.release(d_sup); bufs.release(d_inc);

    // ---- canonical-order results with original vertex ids: the preparation carries the canonical edge list and the internal
    // edge id of every canonical edge (truss_prep.hip), so the gather is one pass.  An induced subgraph's canonical order is the
    // whole graph's restricted to it (its ids are monotone): the same pass, and its endpoints mapped back to the original ids.
    phase.next("truss: canonical gather");
    ctx->timer.start(s);
    // every edge's trussness (from its sub-round stamp, or from the finish) goes where its canonical id says; its initial support
    // is the length of its index slice: the slice table stays with the result for komb_truss_fetch_support
    KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_truss, (size_t)m * sizeof(int32_t)));
    uint32_t k_lo = 0, k_hi = (uint32_t)m;
    if (!vmask_host) {
        ctx->d_t_eu = ctx->d_ceu; ctx->d_t_ev = ctx->d_cev;          // (the graph's canonical edge list: made when a fetch asks for endpoints)
        ctx->t_own_edges = false;
        // (komb_truss_run_slice: this rank's slice of the canonical edges only, zeros elsewhere)
        if (ctx->slice_world > 1) {
            shard_bounds((uint64_t)m, ctx->slice_rank, ctx->slice_world, &k_lo, &k_hi);
            if (k_lo) KOMB_HIP(ctx, hipMemsetAsync(ctx->d_t_truss, 0, (size_t)k_lo * sizeof(int32_t), s));
            if (k_hi < (uint32_t)m) KOMB_HIP(ctx, hipMemsetAsync(ctx->d_t_truss + k_hi, 0, ((size_t)m - k_hi) * sizeof(int32_t), s));
        }
    } else {
        ctx->t_own_edges = true;
        KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_eu, (size_t)m * sizeof(int32_t)));
        KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_ev, (size_t)m * sizeof(int32_t)));
        KOMB_TRY(edge_list(ctx, sub.rowptr, sub.col, sub.nv, sub.vold, ctx->d_t_eu, ctx->d_t_ev));
    }
    ctx->t_k_lo = k_lo; ctx->t_k_hi = k_hi;
    k_truss_results<<<grid_for(m), kBlock, 0, s>>>(d_stamp, Q.rlevel, d_truss, tp->e2k, m, k_lo, k_hi, ctx->d_t_truss);
    ctx->d_t_slice = d_off2;
    bufs.detach(d_off2);
    ctx->t_sup_ready = false;
    if (vmask_host) {                                                // (the subgraph's preparation goes with this call: its supports are put in order now)
        ctx->t_ne = m;
        KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_sup, (size_t)m * sizeof(int32_t)));
        k_scatter_len<<<grid_for(m), kBlock, 0, s>>>(ctx->d_t_slice, tp->e2k, m, 0u, (uint32_t)m, ctx->d_t_sup);
        ctx->t_sup_ready = true;
    }
    st.ms_gather = ctx->timer.stop(s);
    if (ctx_flag(ctx, "POOL_DEBUG")) {
        size_t held = 0;
        for (const auto &b : ctx->pool.blocks) held += b.bytes;
        fprintf(stderr, "komb pool: %zu blocks, %.2f GB held, %zu hipMalloc calls so far (%.1f ms inside them), %zu trims\n", ctx->pool.blocks.size(), held / 1e9, ctx->pool.n_malloc, ctx->pool.ms_malloc, ctx->pool.n_trim);
    }
    if (vmask_host) KOMB_HIP(ctx, hipStreamSynchronize(s));         // (the subgraph's arrays go back to the pool on return)
    ctx->t_ne = m;
    ctx->truss_done = true;
    return KOMB_OK;
}

} // namespace komb
