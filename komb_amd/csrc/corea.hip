// corea.hip -- rows a9 + a10 of the hot-path table: the two fractional-rank
// vectors behind CoreA's anomaly score (reference src/CoreA.h:109-140 calls
// fractionalRank, src/CoreA.h:142-187, on coreness*n+degree and on degree).
//
// The reference ranks in O(U*n) (two full passes per distinct value).  Here:
// radix-sort (key, index) pairs descending, then every sorted position finds
// the bounds [first,last] of its run of equal keys by binary search and
// scatters rank = ((first+1)+(last+1))/2 -- the exact half-integer the
// reference's sum-then-divide produces (sums stay below 2^53).
// Keys are 64-bit: src/CoreA.h:122 computes them in `int`, which is undefined
// once max_coreness*n + degree reaches 2^31.
#include "common.h"

namespace komb {

namespace {

__global__ __launch_bounds__(kBlock) void k_corea_keys(const int32_t *__restrict__ deg, const int32_t *__restrict__ core,
                                                       int64_t n, int64_t *__restrict__ kdeg, int64_t *__restrict__ kcore,
                                                       uint32_t *__restrict__ idx0, uint32_t *__restrict__ idx1)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t d = deg[i];
        kdeg[i] = d;
        kcore[i] = (int64_t)core[i] * n + d;
        idx0[i] = (uint32_t)i; idx1[i] = (uint32_t)i;
    }
}

// sk descending.  first = smallest p with sk[p] <= key; last = (smallest p with sk[p] < key) - 1
__global__ __launch_bounds__(kBlock) void k_run_ranks(const int64_t *__restrict__ sk, const uint32_t *__restrict__ sidx,
                                                      int64_t n, double *__restrict__ rank)
{
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        const int64_t key = sk[p];
        int64_t lo = 0, hi = p;                       // first is in [0,p]
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (sk[mid] <= key) hi = mid; else lo = mid + 1;
        }
        const int64_t first = lo;
        lo = p + 1; hi = n;                           // first position with a smaller key is in (p,n]
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (sk[mid] < key) hi = mid; else lo = mid + 1;
        }
        const int64_t last = lo - 1;
        rank[sidx[p]] = (double)((first + 1) + (last + 1)) / 2.0;
    }
}

inline int bits_for(int64_t maxval)
{
    int b = 1;
    while (b < 63 && (maxval >> b) != 0) ++b;
    return b;
}

} // namespace

int corea_ranks(komb_ctx *ctx, const int32_t *deg, const int32_t *core, int64_t n, double *rank_deg, double *rank_key)
{
    if (n < 0 || (n > 0 && (!deg || !core || !rank_deg || !rank_key)))
        KOMB_FAIL(ctx, KOMB_ERR_ARG, "komb_corea: bad arguments");
    ctx->stats.ms_corea = 0.0;
    if (n == 0) return KOMB_OK;
    if (n > INT32_MAX - 1) KOMB_FAIL(ctx, KOMB_ERR_LIMIT, "komb_corea: n exceeds 2^31-2");
    int32_t maxd = 0, maxc = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (deg[i] < 0 || core[i] < 0) KOMB_FAIL(ctx, KOMB_ERR_ARG, "komb_corea: negative degree/coreness at %lld", (long long)i);
        if (deg[i] > maxd) maxd = deg[i];
        if (core[i] > maxc) maxc = core[i];
    }
    hipStream_t s = ctx->stream;
    DevBufs bufs(ctx);                                   // scratch from the context's pool, returned on every exit
    int32_t *d_deg = nullptr, *d_core = nullptr; int64_t *d_k[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t *d_i[4] = {nullptr, nullptr, nullptr, nullptr}; double *d_r[2] = {nullptr, nullptr};
    KOMB_HIP(ctx, bufs.alloc(&d_deg, (size_t)n));
    KOMB_HIP(ctx, bufs.alloc(&d_core, (size_t)n));
    for (int t = 0; t < 4; ++t) KOMB_HIP(ctx, bufs.alloc(&d_k[t], (size_t)n));
    for (int t = 0; t < 4; ++t) KOMB_HIP(ctx, bufs.alloc(&d_i[t], (size_t)n));
    for (int t = 0; t < 2; ++t) KOMB_HIP(ctx, bufs.alloc(&d_r[t], (size_t)n));
    KOMB_HIP(ctx, staged_copy(ctx, d_deg, deg, (size_t)n * 4, true));
    KOMB_HIP(ctx, staged_copy(ctx, d_core, core, (size_t)n * 4, true));
    int64_t g64 = (n + kBlock - 1) / kBlock;
    const int grid = (int)(g64 > 4096 ? 4096 : g64);
    ctx->timer.start(s);
    k_corea_keys<<<grid, kBlock, 0, s>>>(d_deg, d_core, n, d_k[0], d_k[2], d_i[0], d_i[2]);
    int64_t *sk = nullptr; uint32_t *si = nullptr;
    int st = prim_sort_pairs_desc_i64(ctx, d_k[0], d_k[1], d_i[0], d_i[1], n, bits_for(maxd), &sk, &si);
    if (st == KOMB_OK) {
        k_run_ranks<<<grid, kBlock, 0, s>>>(sk, si, n, d_r[0]);
        st = prim_sort_pairs_desc_i64(ctx, d_k[2], d_k[3], d_i[2], d_i[3], n, bits_for((int64_t)maxc * n + maxd), &sk, &si);
    }
    if (st == KOMB_OK) k_run_ranks<<<grid, kBlock, 0, s>>>(sk, si, n, d_r[1]);
    ctx->stats.ms_corea = ctx->timer.stop(s);
    KOMB_TRY(st);
    KOMB_HIP(ctx, staged_copy(ctx, rank_deg, d_r[0], (size_t)n * 8, false));
    KOMB_HIP(ctx, staged_copy(ctx, rank_key, d_r[1], (size_t)n * 8, false));
    return KOMB_OK;
}

} // namespace komb
