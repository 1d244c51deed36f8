// graph_build.hip -- row a1 of the hot-path table: the simple undirected graph
// igraph_create + igraph_simplify(multiple=true, loops=true) would hold
// (reference src/graph.cpp:418, src/graph.cpp:438), built on the device and
// left resident in HBM as a symmetric CSR with ascending rows.
//
// Layout: rowptr uint32[nv+1] (slot offsets; the design limit is 2^32-1 slots),
// col int32[2*ne].  Both directions of every edge are materialised as 64-bit
// keys (src<<32 | dst), radix-sorted and uniqued; col is the low word of the
// surviving keys and rowptr the positions where the high word changes.
#include "common.h"

namespace komb {

namespace {

constexpr uint64_t kDropKey = ~0ull;          // loops / padding sort to the end

__global__ __launch_bounds__(kBlock) void k_make_keys(const int64_t *__restrict__ uv, int64_t n_raw, int64_t nv,
                                                      uint64_t *__restrict__ keys, int *__restrict__ bad)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_raw; i += (int64_t)gridDim.x * kBlock) {
        const int64_t u = uv[2 * i], v = uv[2 * i + 1];
        uint64_t k0 = kDropKey, k1 = kDropKey;
        if (u < 0 || v < 0 || u >= nv || v >= nv) *bad = 1;
        else if (u != v) {
            k0 = ((uint64_t)u << 32) | (uint64_t)v;
            k1 = ((uint64_t)v << 32) | (uint64_t)u;
        }
        keys[2 * i] = k0;
        keys[2 * i + 1] = k1;
    }
}

// uniq[0..ns) sorted by (src,dst): col[j] = dst, rowptr[v] = first slot with src >= v
__global__ __launch_bounds__(kBlock) void k_keys_to_csr(const uint64_t *__restrict__ uniq, int64_t ns, int64_t nv,
                                                        uint32_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                        int32_t *__restrict__ src)
{
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < ns; j += (int64_t)gridDim.x * kBlock) {
        const uint64_t k = uniq[j];
        const int64_t s = (int64_t)(k >> 32);
        col[j] = (int32_t)(k & 0xFFFFFFFFu);
        src[j] = (int32_t)s;
        const int64_t p = (j == 0) ? -1 : (int64_t)(uniq[j - 1] >> 32);
        for (int64_t v = p + 1; v <= s; ++v) rowptr[v] = (uint32_t)j;
        if (j == ns - 1)
            for (int64_t v = s + 1; v <= nv; ++v) rowptr[v] = (uint32_t)ns;
    }
}

__global__ __launch_bounds__(kBlock) void k_fill_u32(uint32_t *p, int64_t n, uint32_t val)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = val;
}

__global__ __launch_bounds__(kBlock) void k_rowptr_narrow(const int64_t *__restrict__ in, int64_t n,
                                                          uint32_t *__restrict__ out, int *__restrict__ bad)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t x = in[i];
        if (x < 0 || x > 0xFFFFFFFFll || (i > 0 && in[i - 1] > x)) *bad = 1;
        out[i] = (uint32_t)x;
    }
}

// simple + sorted + symmetric check of a caller-supplied CSR (one thread per slot's row)
__global__ __launch_bounds__(kBlock) void k_validate_csr(const uint32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                         int64_t nv, int *__restrict__ bad)
{
    for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < nv; u += (int64_t)gridDim.x * kBlock) {
        const uint32_t b = rowptr[u], e = rowptr[u + 1];
        for (uint32_t j = b; j < e; ++j) {
            const int32_t v = col[j];
            if (v < 0 || v >= nv || v == (int32_t)u || (j > b && col[j - 1] >= v)) { *bad = 1; continue; }
            uint32_t lo = rowptr[v], hi = rowptr[v + 1];          // u must be in row v
            bool found = false;
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                const int32_t c = col[mid];
                if (c < (int32_t)u) lo = mid + 1; else if (c > (int32_t)u) hi = mid; else { found = true; break; }
            }
            if (!found) *bad = 1;
        }
    }
}

// src[j] = row of slot j, for a caller-supplied CSR (one wavefront per row, once per graph)
__global__ __launch_bounds__(kBlock) void k_fill_src(const uint32_t *__restrict__ rowptr, int64_t nv, int32_t *__restrict__ src)
{
    const int lane = (int)(threadIdx.x & 63);
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
    for (int64_t v = wave; v < nv; v += nwaves)
        for (uint32_t j = rowptr[v] + (uint32_t)lane; j < rowptr[v + 1]; j += 64) src[j] = (int32_t)v;
}

inline int grid_for(int64_t n)
{
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (int)g;
}

} // namespace

void graph_free(komb_ctx *ctx)
{
    if (ctx->d_rowptr) (void)hipFree(ctx->d_rowptr);
    if (ctx->d_col) (void)hipFree(ctx->d_col);
    if (ctx->d_src) (void)hipFree(ctx->d_src);
    if (ctx->d_deg) (void)hipFree(ctx->d_deg);
    if (ctx->d_core) (void)hipFree(ctx->d_core);
    ctx->d_rowptr = nullptr; ctx->d_col = nullptr; ctx->d_src = nullptr; ctx->d_deg = nullptr; ctx->d_core = nullptr;
    ctx->nv = -1; ctx->ne = 0; ctx->core_done = false; ctx->moments_valid = false;
    truss_free(ctx);
    ctx->pool.clear();                                   // scratch sized for the old graph
}

int graph_from_edges(komb_ctx *ctx, int64_t nv, int64_t n_raw, const int64_t *uv)
{
    graph_free(ctx);
    if (nv < 0 || nv > INT32_MAX - 1 || n_raw < 0 || (n_raw > 0 && !uv))
        KOMB_FAIL(ctx, KOMB_ERR_ARG, "graph_from_edges: bad nv=%lld n_raw=%lld", (long long)nv, (long long)n_raw);
    if (n_raw > (int64_t)1 << 36)                        // (what bounds a graph is its simple form: 2^32-16 slots, checked below)
        KOMB_FAIL(ctx, KOMB_ERR_LIMIT, "graph_from_edges: %lld raw pairs", (long long)n_raw);
    hipStream_t s = ctx->stream;
    Range r_all("komb_graph_from_edges");
    ctx->timer.start(s);

    KOMB_HIP(ctx, hipMalloc(&ctx->d_rowptr, (size_t)(nv + 1) * sizeof(uint32_t)));
    int64_t ns = 0;
    if (n_raw > 0) {
        int64_t *d_uv = nullptr; uint64_t *d_k0 = nullptr, *d_k1 = nullptr; int *d_bad = nullptr;
        auto cleanup = [&]() { if (d_uv) (void)hipFree(d_uv); if (d_k0) (void)hipFree(d_k0); if (d_k1) (void)hipFree(d_k1); if (d_bad) (void)hipFree(d_bad); };
        const int64_t nk = 2 * n_raw;
        hipError_t e = hipMalloc(&d_uv, (size_t)nk * sizeof(int64_t));
        if (e == hipSuccess) e = hipMalloc(&d_k0, (size_t)nk * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc(&d_k1, (size_t)nk * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc(&d_bad, sizeof(int));
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(int), s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_uv, uv, (size_t)nk * sizeof(int64_t), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) { cleanup(); KOMB_HIP(ctx, e); }
        k_make_keys<<<grid_for(n_raw), kBlock, 0, s>>>(d_uv, n_raw, nv, d_k0, d_bad);
        int bad = 0;
        e = hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { cleanup(); KOMB_HIP(ctx, e); }
        if (bad) { cleanup(); KOMB_FAIL(ctx, KOMB_ERR_ARG, "graph_from_edges: vertex id outside [0,%lld)", (long long)nv); }
        (void)hipFree(d_uv); d_uv = nullptr;

        uint64_t *sorted = nullptr;
        int st = prim_sort_u64(ctx, d_k0, d_k1, nk, 64, &sorted);
        uint64_t *other = (sorted == d_k0) ? d_k1 : d_k0;
        int64_t nu = 0;
        if (st == KOMB_OK) st = prim_unique_u64(ctx, sorted, other, nk, &nu);
        if (st != KOMB_OK) { cleanup(); return st; }
        // the drop key, if present, is the last unique key
        if (nu > 0) {
            uint64_t last = 0;
            e = d2h(ctx, &last, other + (nu - 1), sizeof(uint64_t));
            if (e != hipSuccess) { cleanup(); KOMB_HIP(ctx, e); }
            if (last == kDropKey) --nu;
        }
        ns = nu;
        if (ns > 0xFFFFFFF0ll) { cleanup(); KOMB_FAIL(ctx, KOMB_ERR_LIMIT, "graph has %lld slots; limit is 2^32-16", (long long)ns); }
        if (ns / 2 > INT32_MAX - 16) { cleanup(); KOMB_FAIL(ctx, KOMB_ERR_LIMIT, "graph has %lld edges; limit is 2^31-16", (long long)(ns / 2)); }
        e = hipMalloc(&ctx->d_col, (size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc(&ctx->d_src, (size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
        if (e != hipSuccess) { cleanup(); KOMB_HIP(ctx, e); }
        if (ns > 0) k_keys_to_csr<<<grid_for(ns), kBlock, 0, s>>>(other, ns, nv, ctx->d_rowptr, ctx->d_col, ctx->d_src);
        else k_fill_u32<<<grid_for(nv + 1), kBlock, 0, s>>>(ctx->d_rowptr, nv + 1, 0u);
        e = hipStreamSynchronize(s);
        cleanup();
        KOMB_HIP(ctx, e);
    } else {
        KOMB_HIP(ctx, hipMalloc(&ctx->d_col, sizeof(int32_t)));
        KOMB_HIP(ctx, hipMalloc(&ctx->d_src, sizeof(int32_t)));
        k_fill_u32<<<grid_for(nv + 1), kBlock, 0, s>>>(ctx->d_rowptr, nv + 1, 0u);
        KOMB_HIP(ctx, hipStreamSynchronize(s));
    }
    ctx->nv = nv;
    ctx->ne = ns / 2;
    ctx->stats = komb_stats{};
    ctx->stats.nv = nv; ctx->stats.ne = ctx->ne;
    ctx->stats.ms_build = ctx->timer.stop(s);
    return KOMB_OK;
}

int graph_from_csr(komb_ctx *ctx, int64_t nv, const int64_t *rowptr, const int32_t *col)
{
    graph_free(ctx);
    if (nv < 0 || nv > INT32_MAX - 1 || !rowptr)
        KOMB_FAIL(ctx, KOMB_ERR_ARG, "graph_from_csr: bad nv=%lld", (long long)nv);
    const int64_t ns = rowptr[nv];
    if (rowptr[0] != 0 || ns < 0 || (ns & 1) || (ns > 0 && !col))
        KOMB_FAIL(ctx, KOMB_ERR_ARG, "graph_from_csr: rowptr[0]=%lld rowptr[nv]=%lld is not a symmetric CSR", (long long)rowptr[0], (long long)ns);
    if (ns > 0xFFFFFFF0ll || ns / 2 > INT32_MAX - 16)
        KOMB_FAIL(ctx, KOMB_ERR_LIMIT, "graph has %lld slots; limit is 2^32-16", (long long)ns);
    hipStream_t s = ctx->stream;
    ctx->timer.start(s);
    int64_t *d_rp64 = nullptr; int *d_bad = nullptr;
    hipError_t e = hipMalloc(&ctx->d_rowptr, (size_t)(nv + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&ctx->d_col, (size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&ctx->d_src, (size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&d_rp64, (size_t)(nv + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&d_bad, sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(int), s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rp64, rowptr, (size_t)(nv + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && ns > 0) e = hipMemcpyAsync(ctx->d_col, col, (size_t)ns * sizeof(int32_t), hipMemcpyHostToDevice, s);
    int bad = 0;
    if (e == hipSuccess) {
        k_rowptr_narrow<<<grid_for(nv + 1), kBlock, 0, s>>>(d_rp64, nv + 1, ctx->d_rowptr, d_bad);
        e = hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (e == hipSuccess && !bad && nv > 0) {                     // rowptr is sane: rows can be walked safely
        k_fill_src<<<grid_for(nv * 16), kBlock, 0, s>>>(ctx->d_rowptr, nv, ctx->d_src);
        k_validate_csr<<<grid_for(nv), kBlock, 0, s>>>(ctx->d_rowptr, ctx->d_col, nv, d_bad);
        e = hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (d_rp64) (void)hipFree(d_rp64);
    if (d_bad) (void)hipFree(d_bad);
    if (e != hipSuccess) { graph_free(ctx); KOMB_HIP(ctx, e); }
    if (bad) { graph_free(ctx); KOMB_FAIL(ctx, KOMB_ERR_ARG, "graph_from_csr: CSR is not simple, symmetric and row-sorted"); }
    ctx->nv = nv;
    ctx->ne = ns / 2;
    ctx->stats = komb_stats{};
    ctx->stats.nv = nv; ctx->stats.ne = ctx->ne;
    ctx->stats.ms_build = ctx->timer.stop(s);
    return KOMB_OK;
}

} // namespace komb
