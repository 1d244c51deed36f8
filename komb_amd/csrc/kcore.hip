// kcore.hip -- rows a2 + a3 of the hot-path table: per-vertex degree and
// coreness, replacing igraph_degree(ALL,NO_LOOPS) + igraph_coreness(ALL)
// (reference src/graph.cpp:462-463).
//
// The peel engine of peel_dev.h with: unit = vertex, key = live degree,
// slice = the vertex's CSR row.  Peeling v at level k walks its row (coalesced
// col reads) and atomically decrements every live neighbour; the decrement
// that lands a neighbour exactly on k triggers it.  core[] doubles as the
// liveness flag (an alive marker, common.h, until peeled).  Hub rows are split into 256-slot
// chunks across wavefronts; low-degree frontier vertices are packed 64 to a
// wavefront with their rows flattened over the lanes.
// Coreness is a unique integer per vertex, so peeling order inside a level
// does not matter; results equal Batagelj-Zaversnik's.
// The peel walks the graph in the caller's (ORIGINAL) vertex ids, not in the (degree,id)-ranked internal ids the k-truss path
// uses: in degree order the hubs' live-degree words share cache lines, and the decrements that nearly every peeled vertex
// sends to some hub serialise on those few lines (measured at |V| = 10M: 18.6 ms against 10.5).
#include "peel_dev.h"
#include "core_tail.h"
#include "local_dev.h"
#include "shard_dev.h"

#include <cstdlib>

namespace komb {

namespace {

// degree (a2) + peel state.  Isolated vertices are peeled here (coreness 0); init[0] counts them and
// init[1] receives the smallest positive degree = the first populated level.
__global__ __launch_bounds__(kBlock) void k_core_init(const uint32_t *__restrict__ rowptr, int64_t nv,
                                                      int32_t *__restrict__ deg, int32_t *__restrict__ degw,
                                                      int32_t *__restrict__ core, uint32_t *__restrict__ init)
{
    uint32_t zeros = 0;
    int32_t lmin = 0x7FFFFFFF;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += (int64_t)gridDim.x * kBlock) {
        const int32_t d = (int32_t)(rowptr[v + 1] - rowptr[v]);
        deg[v] = d; degw[v] = d;
        if (d == 0) { core[v] = 0; ++zeros; }
        else { core[v] = alive_marker((uint32_t)d); lmin = min(lmin, d); }
    }
    block_add_min(zeros, lmin, &init[0], (int32_t *)&init[1]);
}

struct CoreProblem {
    static constexpr bool kChain = true;
    static constexpr bool kSingleStep = false;
    uint32_t units;
    const uint32_t *rowptr;
    const int32_t *col;
    int32_t *degw;
    int32_t *core;

    __device__ __forceinline__ const int32_t *scan_marker() const { return core; }
    __device__ __forceinline__ const int32_t *scan_key() const { return degw; }
    __device__ __forceinline__ void mark_scanned(uint32_t v, const CtrlView &cv) const { core[v] = cv.level; }
    __device__ __forceinline__ void slice(uint32_t v, uint32_t &b, uint32_t &len) const
    {
        b = rowptr[v];
        len = rowptr[v + 1] - b;
    }
    struct Loaded { int32_t u, c; };
    __device__ __forceinline__ Loaded item_load(int32_t, uint32_t pos, const CtrlView &) const
    {
        Loaded ld;
        ld.u = col[pos];
        ld.c = core[ld.u];
        return ld;
    }
    __device__ __forceinline__ void item_apply(const Loaded &ld, const CtrlView &cv, int32_t &t0, int32_t &, uint32_t &c0, uint32_t &) const
    {
        if (marker_alive(ld.c)) {                       // a stale "alive" only costs a no-op decrement
            if (atomicSub(&degw[ld.u], 1) == cv.level + 1) { core[ld.u] = cv.level; t0 = ld.u; c0 = marker_chunks(ld.c); }
        }
    }
};

// The same peel with the live degrees owned by vertex range (shard_dev.h): every rank walks every frontier vertex's row;
// a decrement is applied -- and can trigger -- only on the rank that owns the neighbour.  No in-launch chaining: a
// triggered vertex has to reach the other ranks before its row is walked.
struct ShardCore {
    static constexpr bool kChain = false;
    static constexpr bool kSingleStep = true;
    uint32_t units;
    const uint32_t *rowptr;
    const int32_t *col;
    int32_t *degw;
    int32_t *core;
    uint32_t lo, hi;                     // vertices this rank owns

    __device__ __forceinline__ const int32_t *scan_marker() const { return core; }
    __device__ __forceinline__ const int32_t *scan_key() const { return degw; }
    __device__ __forceinline__ void mark_scanned(uint32_t v, const CtrlView &cv) const { core[v] = cv.level; }
    __device__ __forceinline__ void slice(uint32_t v, uint32_t &b, uint32_t &len) const
    {
        b = rowptr[v];
        len = rowptr[v + 1] - b;
    }
    struct Loaded { int32_t u, c; };
    __device__ __forceinline__ Loaded item_load(int32_t, uint32_t pos, const CtrlView &) const
    {
        Loaded ld;
        ld.u = col[pos];
        ld.c = ((uint32_t)ld.u - lo < hi - lo) ? core[ld.u] : 0;      // 0: not this rank's to decrement
        return ld;
    }
    __device__ __forceinline__ void item_apply(const Loaded &ld, const CtrlView &cv, int32_t &t0, int32_t &, uint32_t &, uint32_t &) const
    {
        if (marker_alive(ld.c)) {
            if (atomicSub(&degw[ld.u], 1) == cv.level + 1) { core[ld.u] = cv.level; t0 = ld.u; }
        }
    }
};

// ---- hand-over to the local finish (local_dev.h)
// Collect pass: the peel engine run once over all live vertices; a live neighbour of a live vertex becomes an
// entry of the compact row (ids of the remainder).  Liveness of the neighbour comes from a bitmap (nv / 8 bytes,
// L2-resident) instead of a gather into core[].
struct CoreCollect {
    static constexpr bool kChain = false;
    static constexpr bool kSingleStep = false;
    uint32_t units;
    const uint32_t *rowptr;
    const int32_t *col;
    const int32_t *core;                 // alive markers
    const unsigned long long *livebits;
    const int32_t *num;                  // [nv] id in the remainder (live vertices only)
    const uint32_t *off;                 // [n+1] compact row offsets
    uint32_t *cur;                       // [n] fill cursors
    uint32_t *nbr;                       // compact rows

    __device__ __forceinline__ const int32_t *scan_marker() const { return core; }
    __device__ __forceinline__ const int32_t *scan_key() const { return nullptr; }   // every live vertex enters the one frontier of this pass
    __device__ __forceinline__ void mark_scanned(uint32_t, const CtrlView &) const {}
    __device__ __forceinline__ void slice(uint32_t v, uint32_t &b, uint32_t &len) const
    {
        b = rowptr[v];
        len = rowptr[v + 1] - b;
    }
    struct Loaded { int32_t me, u; bool live; };
    __device__ __forceinline__ Loaded item_load(int32_t me, uint32_t pos, const CtrlView &) const
    {
        Loaded ld;
        ld.me = me;
        ld.u = col[pos];
        ld.live = (livebits[(uint32_t)ld.u >> 6] >> ((uint32_t)ld.u & 63u)) & 1ull;
        return ld;
    }
    __device__ __forceinline__ void item_apply(const Loaded &ld, const CtrlView &, int32_t &, int32_t &, uint32_t &, uint32_t &) const
    {
        if (!ld.live) return;
        const uint32_t id = (uint32_t)num[ld.me];
        const uint32_t b0 = off[id], k = local_slot(id, cur);           // (bounded: see TrussCollect)
        if (k < off[id + 1] - b0) nbr[b0 + k] = (uint32_t)num[ld.u];
    }
};

// Fixed-point problem: item = a live neighbour, value = its current bound.
// remainders with more slots than this -- or a sixth of the graph's, if that is more -- stay with the peel (see
// local_item_limit; measured: C3 hands over 25M of 200M slots, 3x C3 75M of 600M, the alpha = 2.2 stress shape does best
// with 24M of 215M and loses 20% with 73M)
constexpr uint64_t kCoreLocalItems = 32ull << 20;
struct CoreLocal {
    static constexpr int kU = 8;
    static constexpr int kGroups = 1;         // light unit: <= 512 live neighbours (one batch = 64 lanes x 8 values)
    static constexpr int kN = 1;
    const uint32_t *nbr;
    __device__ __forceinline__ void ids(uint32_t pos, uint32_t (&id)[1]) const { id[0] = nbr[pos]; }
};

__global__ void k_live_bits(const int32_t *__restrict__ gid, uint32_t n, unsigned long long *__restrict__ bits)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t v = (uint32_t)gid[i];
        atomicOr(&bits[v >> 6], 1ull << (v & 63u));
    }
}

// control block of a collect pass: one SCAN over the live units of `from` (everything is a hit), one PROCESS
__global__ void k_collect_ctrl(PeelCtrl *c2, const PeelCtrl *from)
{
    if (threadIdx.x == 0) {
        PeelCtrl c{};
        c.mode = MODE_SCAN; c.round = 1; c.level = 0x3FFFFFFF; c.seq = 1;
        c.remaining = from->remaining;
        c.live_mode = from->live_mode; c.live_sel = from->live_sel; c.live_count = from->live_count;
        c.next_min = 0x7FFFFFFF;
        *c2 = c;
    }
}

// grp_done[kInitOff] = #units peeled by the init kernel, grp_done[kInitOff+1] = min live key, grp_done[kInitOff+2] = light units the
// init kernel has already put into queue 0 as the first level's frontier (k-truss stream build; 0 = the peel starts with a SCAN)
__global__ void k_ctrl_pre(uint32_t *grp_done)
{
    for (int i = threadIdx.x; i < kInitOff; i += blockDim.x) grp_done[i] = 0u;
    if (threadIdx.x == 0) { grp_done[kInitOff] = 0u; grp_done[kInitOff + 1] = 0x7FFFFFFFu; grp_done[kInitOff + 2] = 0u; grp_done[kInitOff + 3] = 0u; }
}
__global__ void k_ctrl_init(PeelCtrl *ctrl, const uint32_t *grp_done, uint32_t units, uint32_t tail_limit)
{
    if (threadIdx.x == 0) {
        const uint32_t peeled = grp_done[kInitOff];
        const int32_t first = (int32_t)grp_done[kInitOff + 1];
        const uint32_t front = grp_done[kInitOff + 2];
        PeelCtrl c{};
        c.mode = MODE_SCAN; c.round = 1; c.seq = 1;
        c.last_retire = 1;                           // (the RETIRE-gap invariant counts from the first sub-round)
        c.remaining = units - peeled;
        c.done = (c.remaining == 0) ? 1 : 0;
        c.level = c.remaining ? first : 0;           // start at the first populated level
        c.n_levels = peeled ? 1 : 0;                 // level 0 was populated by item-less units
        c.next_min = 0x7FFFFFFF;
        c.tail_limit = tail_limit;
        if (front) {
            // the first level's frontier is in light queue 0 already (stamped with round 1): what the first SCAN would have left
            c.mode = MODE_PROCESS; c.cur_light = front; c.cur_heavy = 0; c.cur_sel = 0;
            c.remaining -= front;
            c.n_levels += 1; c.max_level = first;
        }
        *ctrl = c;
    }
}

} // namespace

void peel_ctrl_pre(hipStream_t s, uint32_t *d_grp_done) { k_ctrl_pre<<<1, 128, 0, s>>>(d_grp_done); }
void peel_collect_ctrl(hipStream_t s, PeelCtrl *d_collect, const PeelCtrl *d_from) { k_collect_ctrl<<<1, 64, 0, s>>>(d_collect, d_from); }

void peel_ctrl_init(hipStream_t s, PeelCtrl *d_ctrl, uint32_t *d_grp_done, uint32_t units, uint32_t tail_limit)
{
    k_ctrl_init<<<1, 64, 0, s>>>(d_ctrl, d_grp_done, units, tail_limit);
}

int peel_grid(int64_t units)
{
    int64_t g = (units + kPeelBlock - 1) / kPeelBlock;
    if (g < 1) g = 1;
    if (g > 256 * kPeelPerCu) g = 256 * kPeelPerCu;        // one 1024-thread workgroup per CU (16 waves/CU; more workgroups cost more ticket traffic than they hide latency)
    return (int)g;
}

int core_run(komb_ctx *ctx, int rank, int world, komb_allreduce_fn fn, void *user, bool sharded)
{
    if (ctx->nv < 0) KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_core_run: no graph loaded");
    const int64_t nv = ctx->nv;
    hipStream_t s = ctx->stream;
    ctx->core_done = false;
    if (!ctx->d_deg) {
        KOMB_HIP(ctx, hipMalloc(&ctx->d_deg, (size_t)(nv > 0 ? nv : 1) * sizeof(int32_t)));
        KOMB_HIP(ctx, hipMalloc(&ctx->d_core, (size_t)(nv > 0 ? nv : 1) * sizeof(int32_t)));
    }
    komb_stats &stt = ctx->stats;
    stt.core_levels = stt.core_subrounds = stt.core_launches = 0;
    stt.max_coreness = 0; stt.ms_core = 0.0;
    stt.core_local_units = 0; stt.core_local_sweeps = 0; stt.core_local_items = 0; stt.ms_core_local = 0.0;
    if (ctx_flag(ctx, "SHARD_ENGINE")) sharded = true;          // (one rank: the sharded engine without a collective -- a test of its logic)
    stt.shard_exchanges = 0; stt.ms_exchange = 0.0; stt.exchange_words = 0;
    stt.engine_flags = sharded ? KOMB_ENGINE_SHARD_PEEL : 0;
    if (nv == 0) { ctx->core_done = true; return KOMB_OK; }

    Range r_all("komb_core_run");
    DevBufs bufs(ctx);
    const size_t heavy_cap = (size_t)(2 * ctx->ne) / 32 + 64;    // sum over units with > kLight items of ceil(items / kChunk) <= 3/128 of all items
    int32_t *d_degw = nullptr; PeelCtrl *d_ctrl = nullptr; uint32_t *d_grp = nullptr;
    CoreTailBufs T{};
    PeelQueues Q{nullptr, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, 0};
    KOMB_HIP(ctx, bufs.alloc(&d_degw, (size_t)nv));
    for (int i = 0; i < 2; ++i) {
        KOMB_HIP(ctx, bufs.alloc(&Q.light[i], (size_t)nv));
        KOMB_HIP(ctx, bufs.alloc(&Q.heavy[i], heavy_cap));
        KOMB_HIP(ctx, bufs.alloc(&Q.live[i], (size_t)nv / 2 + 64));
    }
    KOMB_HIP(ctx, bufs.alloc(&Q.code, (size_t)nv));
    KOMB_HIP(ctx, bufs.alloc(&d_ctrl, 1));
    KOMB_HIP(ctx, bufs.alloc(&d_grp, (size_t)kInitOff + 4));
    // how the peel ends (common.h): local fixed point (default), LDS tail, or the general engine alone
    const FinishMode fin = finish_mode(ctx, FIN_LOCAL);
    uint32_t tail_limit = 0;
    const size_t live_words = ((size_t)nv + 63) / 64;
    unsigned long long *d_livebits = nullptr;
    if (fin == FIN_LDS) {
        tail_limit = kCoreTailV;
        if (const char *tl = ctx_opt(ctx, "CORE_TAIL")) tail_limit = (uint32_t)strtoul(tl, nullptr, 10);
        if (tail_limit > kCoreTailV) tail_limit = kCoreTailV;
        if (tail_limit) {
            KOMB_HIP(ctx, bufs.alloc(&T.livebits, live_words));
            KOMB_HIP(ctx, bufs.alloc(&T.vnum, (size_t)nv));
            KOMB_HIP(ctx, bufs.alloc(&T.cnt, 4));
            KOMB_HIP(ctx, bufs.alloc(&T.vlist, (size_t)kCoreTailV));
            KOMB_HIP(ctx, bufs.alloc(&T.rows, (size_t)kCoreTailV * kCoreTailWords));
        }
    } else if (fin == FIN_LOCAL) {
        tail_limit = local_limit(ctx, (uint64_t)nv, 16);
        if (tail_limit) KOMB_HIP(ctx, bufs.alloc(&d_livebits, live_words));
    }

    int64_t g = (nv + kBlock - 1) / kBlock;
    const int grid_init = (int)(g > 1024 ? 1024 : g);
    const int grid = peel_grid(nv);
    CoreProblem P{(uint32_t)nv, ctx->d_o_rowptr, ctx->d_o_col, d_degw, ctx->d_core};
    ctx->timer.start(s);
    peel_ctrl_pre(s, d_grp);
    k_core_init<<<grid_init, kBlock, 0, s>>>(ctx->d_o_rowptr, nv, ctx->d_deg, d_degw, ctx->d_core, d_grp + kInitOff);
    peel_ctrl_init(s, d_ctrl, d_grp, (uint32_t)nv, tail_limit);
    // LDS tail: the live vertices are those of `list` (or all nv when list is null) whose core[] is still an alive marker
    auto run_tail = [&](const int32_t *list, uint32_t n_in) -> int {
        KOMB_HIP(ctx, hipMemsetAsync(T.livebits, 0, live_words * sizeof(unsigned long long), s));
        KOMB_HIP(ctx, hipMemsetAsync(T.cnt, 0, 4 * sizeof(uint32_t), s));
        KOMB_HIP(ctx, hipMemsetAsync(T.rows, 0, (size_t)kCoreTailV * kCoreTailWords * sizeof(unsigned long long), s));
        int64_t gm = ((int64_t)n_in + kBlock - 1) / kBlock;
        k_ctail_mark<<<(int)(gm < 1 ? 1 : (gm > 1024 ? 1024 : gm)), kBlock, 0, s>>>(list, n_in, ctx->d_core, T);
        k_ctail_rows<<<dim3(kCoreTailV, 8), kBlock, 0, s>>>(ctx->d_o_rowptr, ctx->d_o_col, T);
        k_core_tail<<<1, 1024, 0, s>>>(d_ctrl, T, d_degw, ctx->d_core);
        KOMB_HIP(ctx, d2h(ctx, &ctx->h_ctrl[0], d_ctrl, sizeof(PeelCtrl)));
        return KOMB_OK;
    };
    // local finish: compact the live subgraph, sweep the h-index operator to its fixed point (local_dev.h)
    auto run_local = [&]() -> int {
        const PeelCtrl hc = ctx->h_ctrl[0];
        Range r_local("core: local finish");
        EventSet evs;
        hipEvent_t ev[2] = {nullptr, nullptr};
        for (auto &e : ev) KOMB_HIP(ctx, evs.make(&e));
        (void)hipEventRecord(ev[0], s);
        KOMB_HIP(ctx, hipMemsetAsync(d_livebits, 0, live_words * sizeof(unsigned long long), s));
        LocalStats ls;
        const int rc = local_finish(ctx, bufs, hc, d_ctrl, (uint32_t)nv, ctx->d_core, d_degw, Q.live[hc.live_sel],
            (uint32_t)kWave * CoreLocal::kU, sizeof(uint32_t), local_item_limit(ctx, std::max<uint64_t>(kCoreLocalItems, (uint64_t)ctx->ne / 3)), 0u, false, 0, ctx->d_core,
            [&](const LocalGraph &lg, const int32_t *num, void *items, PeelCtrl *d_cctrl, int32_t launch) {
                CoreCollect C{(uint32_t)nv, ctx->d_o_rowptr, ctx->d_o_col, ctx->d_core, d_livebits, num, lg.off, lg.cur, (uint32_t *)items};
                k_peel_step<CoreCollect><<<grid, kPeelBlock, 0, s>>>(d_cctrl, d_grp, Q, C, launch);
            },
            [&](const LocalGraph &lg, void *items, uint64_t total, LocalCtrl *d_lctrl, uint32_t *d_cnt, int *launches) -> int {
                return local_fixpoint(ctx, d_lctrl, d_cnt, lg, CoreLocal{(const uint32_t *)items}, total, launches);
            },
            &ls,
            [&](const LocalGraph &lg) { k_live_bits<<<(int)((lg.n + 255) / 256 > 1024 ? 1024 : (lg.n + 255) / 256), 256, 0, s>>>(lg.gid, lg.n, d_livebits); });
        (void)hipEventRecord(ev[1], s);
        (void)hipEventSynchronize(ev[1]);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
        KOMB_TRY(rc);
        if (ls.refused) { ctx->h_ctrl[0].done = 0; ctx->h_ctrl[0].tail_limit = ls.new_limit; return KOMB_OK; }
        stt.core_local_units = (int32_t)ls.units; stt.core_local_sweeps = ls.sweeps; stt.core_local_items = (int64_t)ls.items;
        stt.ms_core_local = (double)ms;
        PeelCtrl done = hc;
        done.done = 1; done.remaining = 0;
        done.n_levels += (int32_t)ls.levels;
        done.max_level = ls.max_val > done.max_level ? ls.max_val : done.max_level;
        ctx->h_ctrl[0] = done;
        return KOMB_OK;
    };
    int launches = 0, st = KOMB_OK;
    if (sharded) {
        // live degrees owned by vertex range, the frontier exchanged every sub-round; the remainder goes to the replicated
        // local finish under the replicated peel's rule, after one exchange of the live degrees (shard_dev.h)
        uint32_t iw[2] = {0u, 0u};
        KOMB_HIP(ctx, d2h(ctx, iw, d_grp + kInitOff, sizeof(iw)));     // isolated vertices; the smallest positive degree
        ShardCore SP{(uint32_t)nv, ctx->d_o_rowptr, ctx->d_o_col, d_degw, ctx->d_core, 0u, 0u};
        shard_bounds((uint64_t)nv, rank, world, &SP.lo, &SP.hi);
        ShardStats ss;
        st = shard_peel(ctx, bufs, SP, d_degw, (uint32_t)nv, iw[0], (int32_t)iw[1], rank, world, fn, user, Q, d_ctrl,
                        [&](int32_t launch) { k_peel_step<ShardCore><<<grid, kPeelBlock, 0, s>>>(d_ctrl, d_grp, Q, SP, launch); },
                        fin == FIN_LOCAL ? tail_limit : 0u, [&]() -> int { return run_local(); }, &ss);
        stt.shard_exchanges = (int32_t)ss.exchanges; stt.ms_exchange = ss.ms_exchange; stt.exchange_words = ss.words;
        launches = ss.launches;
        PeelCtrl fc{};
        fc.done = st == KOMB_OK ? 1 : 2; fc.n_levels = ss.levels; fc.n_rounds = ss.rounds; fc.n_scans = ss.scans; fc.max_level = ss.max_level;
        ctx->h_ctrl[0] = fc;
    } else if (tail_limit && (uint64_t)nv <= tail_limit) {
        // small graph: the finish takes the whole peel (unless nothing is left to peel)
        st = d2h(ctx, &ctx->h_ctrl[0], d_ctrl, sizeof(PeelCtrl)) == hipSuccess ? KOMB_OK : KOMB_ERR_DEVICE;
        if (st == KOMB_OK && !ctx->h_ctrl[0].done) st = (fin == FIN_LOCAL) ? run_local() : run_tail(nullptr, (uint32_t)nv);
    } else {
        ctx->h_ctrl[0].done = 0;
    }
    for (int guard = 0; st == KOMB_OK && ctx->h_ctrl[0].done != 1 && ctx->h_ctrl[0].done != 2 && guard < 64; ++guard) {
        if (ctx->h_ctrl[0].done == 3) {
            const PeelCtrl &c = ctx->h_ctrl[0];
            if (fin == FIN_LOCAL) st = run_local();
            else st = c.live_mode ? run_tail(Q.live[c.live_sel], c.live_count) : run_tail(nullptr, (uint32_t)nv);
            continue;
        }
        int batch = 0;
        st = drive_peel(ctx, d_ctrl, nv, [&](int32_t launch) {
            k_peel_step<CoreProblem><<<grid, kPeelBlock, 0, s>>>(d_ctrl, d_grp, Q, P, launch);
        }, &batch);
        launches += batch;
    }
    stt.ms_core = ctx->timer.stop(s);
#ifdef KOMB_STEP_TIMERS
    {
        const PeelCtrl &c = ctx->h_ctrl[0];
        const double n = c.pad1[6] ? (double)c.pad1[6] : 1.0;
        fprintf(stderr, "komb core step timers (block 0, %u small multi-workgroup PROCESS steps), us per step: ctrl %.2f queue+slice %.2f items %.2f flush %.2f barrier %.2f ticket %.2f\n",
                c.pad1[6], c.pad1[0] / n / 100.0, c.pad1[1] / n / 100.0, c.pad1[2] / n / 100.0, c.pad1[3] / n / 100.0, c.pad1[4] / n / 100.0, c.pad1[5] / n / 100.0);
    }
#endif
    KOMB_TRY(st);
    if (ctx->h_ctrl[0].done != 1) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "k-core peel ended in an inconsistent state");
    if (stt.core_local_units) stt.engine_flags |= KOMB_ENGINE_LOCAL_FINISH;
    else if (fin == FIN_LDS && tail_limit) stt.engine_flags |= KOMB_ENGINE_LDS_TAIL;
    stt.core_levels = ctx->h_ctrl[0].n_levels;
    stt.core_subrounds = ctx->h_ctrl[0].n_rounds;
    stt.core_launches = launches;
    stt.max_coreness = ctx->h_ctrl[0].max_level;
    ctx->core_done = true;
    return KOMB_OK;
}

} // namespace komb
