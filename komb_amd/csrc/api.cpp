// api.cpp -- the extern "C" surface declared in include/komb_accel.h.
// Host-only glue: argument checks, device selection, copies in and out.  All
// arithmetic of the path runs in the HIP kernels of the sibling .hip files;
// there is no CPU fallback (a missing device is KOMB_ERR_DEVICE).
#include "common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

using namespace komb;

namespace {

int require_device(komb_ctx *ctx)
{
    if (!ctx) return KOMB_ERR_ARG;
    if (!ctx->device_ok) {
        if (ctx->err.empty()) ctx->err = "no usable HIP device";
        return KOMB_ERR_DEVICE;
    }
    KOMB_HIP(ctx, hipSetDevice(ctx->device));
    ctx->err.clear();
    return KOMB_OK;
}

} // namespace

extern "C" {

int komb_abi_version(void) { return KOMB_ACCEL_ABI_VERSION; }

komb_ctx *komb_create(const komb_opts *opts)
{
    komb_ctx *ctx = new (std::nothrow) komb_ctx();
    if (!ctx) return nullptr;
    if (opts) ctx->opts = *opts;
    ctx->device = ctx->opts.device;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        ctx->err = std::string("no usable HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        (void)hipGetLastError();
        return ctx;
    }
    if (ctx->device < 0 || ctx->device >= ndev) {
        ctx->err = "device ordinal out of range";
        return ctx;
    }
    e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_ctrl, 2 * sizeof(PeelCtrl), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_local, 2 * sizeof(LocalCtrl), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&ctx->h_stage, kStageBytes, hipHostMallocDefault);
    // Every context works on a stream of its own: a BLOCKING one (hipStreamCreate, not hipStreamNonBlocking), so that the
    // legacy default stream -- the synchronous hipMemcpy calls of the fetch entry points, a host that launches work of its own
    // there -- still orders with it, while two contexts (two host threads, or a host's RCCL stream beside the library) no
    // longer serialise on one queue.  (Round 2 tried a non-blocking stream before the peel's launches carried their sequence
    // word and saw stale control blocks with two processes on one GPU; with the sequence word the engine no longer depends
    // on launch-order visibility)  KOMB_CREATE_NULL_STREAM in komb_opts.reserved[0] puts the context
    // back on the default stream.
    ctx->stream = nullptr;
    if (e == hipSuccess && !(ctx->opts.reserved[0] & KOMB_CREATE_NULL_STREAM)) { e = hipStreamCreate(&ctx->stream); ctx->own_stream = e == hipSuccess; }
    if (e == hipSuccess && !ctx->timer.init()) e = hipErrorUnknown;
    if (e != hipSuccess) {
        ctx->err = std::string("device initialisation failed: ") + hipGetErrorString(e);
        return ctx;
    }
    ctx->device_ok = true;
    // One-time costs of a process' first use of the library -- loading its code object onto the device (the first kernel
    // launch) and, for a caller that asks (KOMB_CREATE_WARM_UPLOAD), the pinned staging buffers of the graph upload -- are
    // paid here, not by the first graph build: komb2 creates its context on a second thread beside the SAM parse, so they
    // leave its critical path (komb_amd/host/komb2.cpp).  KOMB_CREATE_NO_WARMUP skips it.
    if (!(ctx->opts.reserved[0] & KOMB_CREATE_NO_WARMUP)) warm_up(ctx);
    return ctx;
}

void komb_destroy(komb_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->device_ok) {
        (void)hipSetDevice(ctx->device);
        graph_free(ctx);
        stager_free(ctx);
        ctx->pool.clear();
        ctx->timer.destroy();
        if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
        if (ctx->h_ctrl) (void)hipHostFree(ctx->h_ctrl);
        if (ctx->h_local) (void)hipHostFree(ctx->h_local);
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    }
    delete ctx;
}

const char *komb_last_error(const komb_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int komb_graph_from_edges(komb_ctx *ctx, int64_t nv, int64_t n_raw, const int64_t *uv_pairs)
{
    KOMB_TRY(require_device(ctx));
    return graph_from_edges(ctx, nv, n_raw, uv_pairs);
}

int komb_graph_from_csr(komb_ctx *ctx, int64_t nv, const int64_t *rowptr, const int32_t *col)
{
    KOMB_TRY(require_device(ctx));
    return graph_from_csr(ctx, nv, rowptr, col);
}

int komb_graph_info(komb_ctx *ctx, int64_t *nv, int64_t *ne)
{
    if (!ctx) return KOMB_ERR_ARG;
    if (ctx->nv < 0) KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_graph_info: no graph loaded");
    if (nv) *nv = ctx->nv;
    if (ne) *ne = ctx->ne;
    return KOMB_OK;
}

int komb_graph_get_csr(komb_ctx *ctx, int64_t *rowptr, int32_t *col)
{
    KOMB_TRY(require_device(ctx));
    if (ctx->nv < 0) KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_graph_get_csr: no graph loaded");
    if (!rowptr || (ctx->ne > 0 && !col)) KOMB_FAIL(ctx, KOMB_ERR_ARG, "komb_graph_get_csr: null output");
    std::vector<uint32_t> rp((size_t)ctx->nv + 1);
    KOMB_HIP(ctx, hipMemcpy(rp.data(), ctx->d_o_rowptr, rp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < rp.size(); ++i) rowptr[i] = (int64_t)rp[i];
    if (ctx->ne > 0)
        KOMB_HIP(ctx, hipMemcpy(col, ctx->d_o_col, (size_t)(2 * ctx->ne) * sizeof(int32_t), hipMemcpyDeviceToHost));
    return KOMB_OK;
}

int komb_core_run(komb_ctx *ctx)
{
    KOMB_TRY(require_device(ctx));
    return core_run(ctx);
}

int komb_core_run_sharded(komb_ctx *ctx, int32_t rank, int32_t world, komb_allreduce_fn allreduce, void *user)
{
    KOMB_TRY(require_device(ctx));
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !allreduce))
        KOMB_FAIL(ctx, KOMB_ERR_ARG, "komb_core_run_sharded: bad rank %d / world %d / callback", rank, world);
    return core_run(ctx, rank, world, allreduce, user, true);
}

int komb_set_shard_peel(komb_ctx *ctx, int32_t on)
{
    if (!ctx) return KOMB_ERR_ARG;
    ctx->shard_peel = on != 0;
    return KOMB_OK;
}

int komb_core_fetch(komb_ctx *ctx, int32_t *degree, int32_t *coreness)
{
    KOMB_TRY(require_device(ctx));
    if (!ctx->core_done) KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_core_fetch: komb_core_run has not completed");
    if (ctx->nv == 0) return KOMB_OK;
    if (degree) KOMB_HIP(ctx, staged_copy(ctx, degree, ctx->d_deg, (size_t)ctx->nv * sizeof(int32_t), false));
    if (coreness) KOMB_HIP(ctx, staged_copy(ctx, coreness, ctx->d_core, (size_t)ctx->nv * sizeof(int32_t), false));
    return KOMB_OK;
}

int komb_degree_coreness(komb_ctx *ctx, int32_t *degree, int32_t *coreness)
{
    KOMB_TRY(komb_core_run(ctx));
    return komb_core_fetch(ctx, degree, coreness);
}

int komb_set_option(komb_ctx *ctx, const char *name, const char *value)
{
    if (!ctx || !name || !*name) return KOMB_ERR_ARG;
    if (value) ctx->options[name] = value; else ctx->options.erase(name);
    return KOMB_OK;
}

int komb_truss_prepare(komb_ctx *ctx)
{
    KOMB_TRY(require_device(ctx));
    if (ctx->nv < 0) KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_truss_prepare: no graph loaded");
    if (ctx->prep.valid || ctx->nv == 0 || ctx->ne == 0) return KOMB_OK;
    KOMB_TRY(prep_ensure(ctx));
    ctx->stats.ms_prepare = ctx->prep.ms;
    ctx->stats.ms_prep_vertex = ctx->prep.ms_part[0]; ctx->stats.ms_prep_edges = ctx->prep.ms_part[1]; ctx->stats.ms_prep_rows = ctx->prep.ms_part[2];
    return KOMB_OK;
}

int komb_truss_unprepare(komb_ctx *ctx)
{
    KOMB_TRY(require_device(ctx));
    truss_free(ctx);
    prep_free(ctx, &ctx->prep);
    return KOMB_OK;
}

int komb_graph_moments(komb_ctx *ctx)
{
    KOMB_TRY(require_device(ctx));
    if (ctx->nv < 0) KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_graph_moments: no graph loaded");
    int64_t mom[5] = {0, 0, 0, 0, 0};
    if (ctx->nv > 0 && ctx->ne > 0) KOMB_TRY(graph_moments(ctx, mom));
    ctx->stats.sum_deg_sq = mom[0]; ctx->stats.wedge_items = mom[1];
    ctx->stats.max_degree = (int32_t)mom[2]; ctx->stats.oriented_items = mom[4];
    return KOMB_OK;
}

int komb_truss_run(komb_ctx *ctx, const uint8_t *vmask)
{
    KOMB_TRY(require_device(ctx));
    return truss_run(ctx, vmask, 0, 1, nullptr, nullptr);
}

int komb_truss_run_sharded(komb_ctx *ctx, const uint8_t *vmask, int32_t rank, int32_t world,
                           komb_allreduce_fn allreduce, void *user)
{
    KOMB_TRY(require_device(ctx));
    return truss_run(ctx, vmask, rank, world, allreduce, user);
}

int komb_truss_run_slice(komb_ctx *ctx, const uint8_t *vmask, int32_t rank, int32_t world)
{
    KOMB_TRY(require_device(ctx));
    if (world < 1 || rank < 0 || rank >= world) KOMB_FAIL(ctx, KOMB_ERR_ARG, "komb_truss_run_slice: bad rank %d / world %d", rank, world);
    ctx->slice_rank = rank; ctx->slice_world = world;
    const int rc = truss_run(ctx, vmask, 0, 1, nullptr, nullptr);
    ctx->slice_rank = 0; ctx->slice_world = 1;
    return rc;
}

int komb_truss_count(komb_ctx *ctx, int64_t *ne_sub)
{
    if (!ctx) return KOMB_ERR_ARG;
    if (!ctx->truss_done) KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_truss_count: komb_truss_run has not completed");
    if (ne_sub) *ne_sub = ctx->t_ne;
    return KOMB_OK;
}

int komb_truss_fetch(komb_ctx *ctx, int32_t *eu, int32_t *ev, int32_t *truss)
{
    KOMB_TRY(require_device(ctx));
    if (!ctx->truss_done) KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_truss_fetch: komb_truss_run has not completed");
    const size_t bytes = (size_t)ctx->t_ne * sizeof(int32_t);
    if (bytes == 0) return KOMB_OK;
    if (eu || ev) KOMB_TRY(truss_edges_canonical(ctx));  // (the endpoints: igraph_edge after igraph_trussness, src/graph.cpp:529-532 -- not part of the timed call)
    if (eu) KOMB_HIP(ctx, staged_copy(ctx, eu, ctx->d_t_eu, bytes, false));
    if (ev) KOMB_HIP(ctx, staged_copy(ctx, ev, ctx->d_t_ev, bytes, false));
    if (truss) KOMB_HIP(ctx, staged_copy(ctx, truss, ctx->d_t_truss, bytes, false));
    return KOMB_OK;
}

int komb_truss_fetch_support(komb_ctx *ctx, int32_t *support)
{
    KOMB_TRY(require_device(ctx));
    if (!ctx->truss_done) KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_truss_fetch_support: komb_truss_run has not completed");
    KOMB_TRY(truss_support_canonical(ctx));              // (made on the first request: the timed step delivers trussness, as igraph_trussness does)
    const size_t bytes = (size_t)ctx->t_ne * sizeof(int32_t);
    if (bytes && support) KOMB_HIP(ctx, staged_copy(ctx, support, ctx->d_t_sup, bytes, false));
    return KOMB_OK;
}

int komb_trussness(komb_ctx *ctx, const uint8_t *vmask, int64_t *ne_out, int32_t *eu, int32_t *ev, int32_t *truss)
{
    KOMB_TRY(komb_truss_run(ctx, vmask));
    if (ne_out) *ne_out = ctx->t_ne;
    return komb_truss_fetch(ctx, eu, ev, truss);
}

int komb_corea_ranks(komb_ctx *ctx, const int32_t *degree, const int32_t *coreness, int64_t nv,
                     double *rank_degree, double *rank_key)
{
    KOMB_TRY(require_device(ctx));
    return corea_ranks(ctx, degree, coreness, nv, rank_degree, rank_key);
}

int komb_corea_scores(komb_ctx *ctx, const int32_t *degree, const int32_t *coreness, int64_t nv, double *score)
{
    KOMB_TRY(require_device(ctx));
    if (nv < 0 || (nv > 0 && !score)) KOMB_FAIL(ctx, KOMB_ERR_ARG, "komb_corea_scores: bad arguments");
    // (no std::vector: value-initialising 2 x 8 bytes x nv is 40 ms of page faults at nv = 10M before the first rank arrives)
    std::unique_ptr<double[]> rd(new (std::nothrow) double[(size_t)nv + 1]), rc(new (std::nothrow) double[(size_t)nv + 1]);
    if (!rd || !rc) KOMB_FAIL(ctx, KOMB_ERR_NOMEM, "komb_corea_scores: host allocation failed");
    KOMB_TRY(corea_ranks(ctx, degree, coreness, nv, rd.get(), rc.get()));
    // host libm on purpose: |ln r_deg - ln r_key| as src/CoreA.h:131, same "%f" text downstream (element-wise: the host's
    // threads share the loop, every element is the same libm call it would be on one)
    // (a GPU box shows every CPU of its host and grants a share of them: 16 threads at most)
#pragma omp parallel for schedule(static) num_threads(16) if (nv > 100000)
    for (int64_t i = 0; i < nv; ++i) score[i] = std::fabs(std::log(rd[(size_t)i]) - std::log(rc[(size_t)i]));
    return KOMB_OK;
}

int komb_densest_block(komb_ctx *ctx, const double *suspiciousness, int32_t *order, int32_t *side, int64_t *n_block, double *max_density)
{
    KOMB_TRY(require_device(ctx));
    return merge_run(ctx, suspiciousness, order, side, n_block, max_density);
}

int komb_get_stats(komb_ctx *ctx, komb_stats *out)
{
    if (!ctx || !out) return KOMB_ERR_ARG;
    *out = ctx->stats;
    return KOMB_OK;
}

} // extern "C"
