// kcore.hip -- rows a2 + a3 of the hot-path table: per-vertex degree and
// coreness, replacing igraph_degree(ALL,NO_LOOPS) + igraph_coreness(ALL)
// (reference src/graph.cpp:462-463).
//
// Algorithm: level-synchronous peel.  degw[] holds the live degree, core[] is
// kAlive until the vertex is peeled.  Each launch of k_core_step reads the
// device control block and either
//   SCAN    : sweeps all vertices, moves the live ones with degw <= level into
//             the frontier queue (wave ballot + prefix popcount, one atomic per
//             wave) and records the smallest live degree above the level, or
//   PROCESS : gives every frontier vertex to one wavefront, which streams the
//             CSR row (coalesced col reads) and atomically decrements the live
//             neighbours; the lane whose decrement lands a neighbour exactly
//             on the level owns that neighbour and the wave peels it on the
//             spot from an LDS stack, so a whole cascade is followed inside one
//             launch without any cross-workgroup hand-off.
// Coreness is a unique integer per vertex, so the order of peeling inside a
// level does not matter; results equal Batagelj-Zaversnik's.
#include "peel_dev.h"

namespace komb {

namespace {

constexpr int kStack = 1024;                  // chained vertices a wave may hold (LDS, per wave)

__global__ __launch_bounds__(kBlock) void k_core_init(const uint32_t *__restrict__ rowptr, int64_t nv,
                                                      int32_t *__restrict__ deg, int32_t *__restrict__ degw,
                                                      int32_t *__restrict__ core)
{
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += (int64_t)gridDim.x * kBlock) {
        const int32_t d = (int32_t)(rowptr[v + 1] - rowptr[v]);
        deg[v] = d; degw[v] = d; core[v] = kAlive;
    }
}

__global__ __launch_bounds__(kBlock) void k_core_step(PeelCtrl *ctrl, const uint32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ col, int32_t *degw,
                                                      int32_t *core, int32_t *q0, int32_t *q1, int64_t nv)
{
    __shared__ CtrlView sh_cv;
    __shared__ int32_t sh_stack[kBlock / kWave][kStack];
    const CtrlView cv = load_ctrl(ctrl, &sh_cv);
    if (cv.done) return;
    const int k = cv.level;
    const int lane = lane_id();
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;

    if (cv.mode == MODE_SCAN) {
        int32_t *q = cv.cur_sel ? q1 : q0;
        int32_t lmin = 0x7FFFFFFF;
        // wave-uniform trip count: every lane reaches the ballot in wave_append
        for (int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) - lane; base < nv; base += nthreads) {
            const int64_t v = base + lane;
            bool hit = false;
            if (v < nv && core[v] == kAlive) {
                const int32_t d = degw[v];
                if (d <= k) { hit = true; core[v] = k; }
                else lmin = min(lmin, d);
            }
            wave_append(hit, (int32_t)v, q, &ctrl->tail[cv.cur_sel]);
        }
        lmin = wave_min(lmin);
        if (lane == 0 && lmin != 0x7FFFFFFF) atomicMin(&ctrl->next_min, lmin);
    } else {
        const int32_t *q = cv.cur_sel ? q1 : q0;
        int32_t *qn = cv.cur_sel ? q0 : q1;
        uint32_t *tail_n = &ctrl->tail[cv.cur_sel ^ 1];
        int32_t *stack = sh_stack[threadIdx.x >> 6];
        const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
        const int64_t nwaves = nthreads >> 6;
        uint32_t chained = 0;                                   // wave-uniform
        for (int64_t i = wave; i < (int64_t)cv.cur_count; i += nwaves) {
            int32_t cur = q[i];
            int sp = 0;                                         // wave-uniform
            for (;;) {
                const uint32_t b = rowptr[cur], e = rowptr[cur + 1];
                for (uint32_t j0 = b; j0 < e; j0 += kWave) {
                    const uint32_t j = j0 + (uint32_t)lane;
                    bool trig = false;
                    int32_t u = -1;
                    if (j < e) {
                        u = col[j];
                        if (core[u] == kAlive) {                // a stale "alive" only costs a no-op decrement
                            const int32_t old = atomicSub(&degw[u], 1);
                            if (old == k + 1) { trig = true; core[u] = k; }
                        }
                    }
                    const uint64_t m = __ballot(trig);
                    if (m) {
                        const int cnt = __popcll(m);
                        if (sp + cnt <= kStack) {
                            if (trig) stack[sp + __popcll(m & lanemask_lt())] = u;
                            sp += cnt;
                            chained += (uint32_t)cnt;
                        } else {
                            wave_append(trig, u, qn, tail_n);   // spill: next launch peels them
                        }
                    }
                }
                if (sp == 0) break;
                __builtin_amdgcn_wave_barrier();
                cur = stack[--sp];
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0 && chained) atomicAdd(&ctrl->acc, chained);
    }
    finalize_launch(ctrl, cv);
}

__global__ void k_ctrl_init(PeelCtrl *ctrl, uint32_t units)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        PeelCtrl c{};
        c.mode = MODE_SCAN; c.level = 0; c.round = 1; c.done = (units == 0) ? 1 : 0;
        c.cur_count = 0; c.cur_sel = 0; c.remaining = units; c.next_min = 0x7FFFFFFF;
        *ctrl = c;
    }
}

} // namespace

void peel_ctrl_init(hipStream_t s, PeelCtrl *d_ctrl, uint32_t units)
{
    k_ctrl_init<<<1, 64, 0, s>>>(d_ctrl, units);
}

int core_run(komb_ctx *ctx)
{
    if (ctx->nv < 0) KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_core_run: no graph loaded");
    const int64_t nv = ctx->nv;
    hipStream_t s = ctx->stream;
    ctx->core_done = false;
    if (!ctx->d_deg) {
        KOMB_HIP(ctx, hipMalloc(&ctx->d_deg, (size_t)(nv > 0 ? nv : 1) * sizeof(int32_t)));
        KOMB_HIP(ctx, hipMalloc(&ctx->d_core, (size_t)(nv > 0 ? nv : 1) * sizeof(int32_t)));
    }
    ctx->stats.core_levels = ctx->stats.core_launches = 0;
    ctx->stats.max_coreness = 0; ctx->stats.ms_core = 0.0;
    if (nv == 0) { ctx->core_done = true; return KOMB_OK; }

    int32_t *d_degw = nullptr, *d_q0 = nullptr, *d_q1 = nullptr; PeelCtrl *d_ctrl = nullptr;
    auto cleanup = [&]() {
        if (d_degw) (void)hipFree(d_degw); if (d_q0) (void)hipFree(d_q0);
        if (d_q1) (void)hipFree(d_q1); if (d_ctrl) (void)hipFree(d_ctrl);
    };
    hipError_t e = hipMalloc(&d_degw, (size_t)nv * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&d_q0, (size_t)nv * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&d_q1, (size_t)nv * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&d_ctrl, sizeof(PeelCtrl));
    if (e != hipSuccess) { cleanup(); KOMB_HIP(ctx, e); }

    int64_t g = (nv + kBlock - 1) / kBlock;
    const int grid = (int)(g < 256 ? 256 : (g > 2048 ? 2048 : g));
    ctx->timer.start(s);
    k_core_init<<<grid, kBlock, 0, s>>>(ctx->d_rowptr, nv, ctx->d_deg, d_degw, ctx->d_core);
    peel_ctrl_init(s, d_ctrl, (uint32_t)nv);
    int launches = 0;
    int st = drive_peel(ctx, d_ctrl, nv, [&]() {
        k_core_step<<<grid, kBlock, 0, s>>>(d_ctrl, ctx->d_rowptr, ctx->d_col, d_degw, ctx->d_core, d_q0, d_q1, nv);
    }, &launches);
    ctx->stats.ms_core = ctx->timer.stop(s);
    cleanup();
    KOMB_TRY(st);
    if (ctx->h_ctrl[0].done != 1) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "k-core peel ended in an inconsistent state");
    ctx->stats.core_levels = ctx->h_ctrl[0].n_levels;
    ctx->stats.core_launches = launches;
    ctx->stats.max_coreness = ctx->h_ctrl[0].max_level;
    ctx->core_done = true;
    return KOMB_OK;
}

} // namespace komb
