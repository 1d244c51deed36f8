// prims.hip -- device-wide sort / select / scan used by graph construction and
// CoreA ranking.  These are rocPRIM (via hipCUB) library primitives kept
// behind plain signatures so the hand-written kernels' translation units do
// not pull the library headers in.  None of them is on the timed peel path.
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace komb {

namespace {
struct TempBuf {                             // scratch from the context's caching pool
    komb_ctx *ctx;
    void *p = nullptr;
    explicit TempBuf(komb_ctx *c) : ctx(c) {}
    hipError_t get(size_t bytes) { return ctx->pool.get(&p, bytes ? bytes : 16); }
    ~TempBuf() { if (p) ctx->pool.put(p); }
};
} // namespace

int prim_sort_u64(komb_ctx *ctx, uint64_t *keys, uint64_t *tmp_keys, int64_t n, int end_bit, uint64_t **sorted)
{
    // (64-bit item counts: the raw pair list of a graph near the 2^32-slot limit has more than 2^31 keys)
    hipcub::DoubleBuffer<uint64_t> db(keys, tmp_keys);
    size_t bytes = 0;
    KOMB_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, db, (long long)n, 0, end_bit, ctx->stream));
    TempBuf t(ctx);
    KOMB_HIP(ctx, t.get(bytes));
    KOMB_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(t.p, bytes, db, (long long)n, 0, end_bit, ctx->stream));
    KOMB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *sorted = db.Current();
    return KOMB_OK;
}

int prim_unique_u64(komb_ctx *ctx, const uint64_t *in, uint64_t *out, int64_t n, int64_t *n_out)
{
    long long *d_num = nullptr;
    KOMB_HIP(ctx, hipMalloc(&d_num, sizeof(long long)));
    size_t bytes = 0;
    hipError_t e = hipcub::DeviceSelect::Unique(nullptr, bytes, in, out, d_num, (int64_t)n, ctx->stream);
    TempBuf t(ctx);
    if (e == hipSuccess) e = t.get(bytes);
    if (e == hipSuccess) e = hipcub::DeviceSelect::Unique(t.p, bytes, in, out, d_num, (int64_t)n, ctx->stream);
    long long h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d_num, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_num);
    KOMB_HIP(ctx, e);
    *n_out = h;
    return KOMB_OK;
}

int prim_exclusive_sum_u32(komb_ctx *ctx, const uint32_t *in, uint32_t *out, int64_t n)
{
    if (n > INT32_MAX) KOMB_FAIL(ctx, KOMB_ERR_LIMIT, "scan: %lld items exceed the 2^31-1 primitive limit", (long long)n);
    if (n == 0) return KOMB_OK;
    size_t bytes = 0;
    KOMB_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n, ctx->stream));
    TempBuf t(ctx);
    KOMB_HIP(ctx, t.get(bytes));
    KOMB_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(t.p, bytes, in, out, (int)n, ctx->stream));
    return KOMB_OK;                              // asynchronous on the context's stream (the scratch goes back to the pool, which the same stream uses)
}

// exclusive sum of 32-bit counts into 64-bit offsets (totals beyond 2^32)
struct WidenU32 { __host__ __device__ unsigned long long operator()(uint32_t x) const { return (unsigned long long)x; } };
int prim_exclusive_sum_u32_u64(komb_ctx *ctx, const uint32_t *in, unsigned long long *out, int64_t n)
{
    if (n > INT32_MAX) KOMB_FAIL(ctx, KOMB_ERR_LIMIT, "scan: %lld items exceed the 2^31-1 primitive limit", (long long)n);
    if (n == 0) return KOMB_OK;
    hipcub::TransformInputIterator<unsigned long long, WidenU32, const uint32_t *> wide(in, WidenU32());
    size_t bytes = 0;
    KOMB_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, wide, out, (int)n, ctx->stream));
    TempBuf t(ctx);
    KOMB_HIP(ctx, t.get(bytes));
    KOMB_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(t.p, bytes, wide, out, (int)n, ctx->stream));
    return KOMB_OK;
}

int prim_sort_pairs_desc_i64(komb_ctx *ctx, int64_t *keys, int64_t *keys_tmp, uint32_t *vals, uint32_t *vals_tmp,
                             int64_t n, int end_bit, int64_t **sorted_keys, uint32_t **sorted_vals)
{
    if (n > INT32_MAX) KOMB_FAIL(ctx, KOMB_ERR_LIMIT, "sort: %lld keys exceed the 2^31-1 primitive limit", (long long)n);
    // keys are non-negative, so the unsigned bit order of [0,end_bit) is the value order
    hipcub::DoubleBuffer<uint64_t> dk((uint64_t *)keys, (uint64_t *)keys_tmp);
    hipcub::DoubleBuffer<uint32_t> dv(vals, vals_tmp);
    size_t bytes = 0;
    KOMB_HIP(ctx, hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, dk, dv, (int)n, 0, end_bit, ctx->stream));
    TempBuf t(ctx);
    KOMB_HIP(ctx, t.get(bytes));
    KOMB_HIP(ctx, hipcub::DeviceRadixSort::SortPairsDescending(t.p, bytes, dk, dv, (int)n, 0, end_bit, ctx->stream));
    KOMB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *sorted_keys = (int64_t *)dk.Current();
    *sorted_vals = dv.Current();
    return KOMB_OK;
}

// records of the k-truss index build: 32-bit keys (an edge id), 8-byte values (the other two edges of a triangle), sorted
// by the key bits [begin_bit, end_bit).  Asynchronous on the context's stream.
int prim_sort_pairs_u32_u64(komb_ctx *ctx, uint32_t *keys, uint32_t *keys_alt, unsigned long long *vals, unsigned long long *vals_alt,
                            int64_t n, int begin_bit, int end_bit, uint32_t **sorted_keys, unsigned long long **sorted_vals)
{
    hipcub::DoubleBuffer<uint32_t> dk(keys, keys_alt);
    hipcub::DoubleBuffer<unsigned long long> dv(vals, vals_alt);
    if (n > 0) {
        size_t bytes = 0;
        KOMB_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dk, dv, (long long)n, begin_bit, end_bit, ctx->stream));
        TempBuf t(ctx);
        KOMB_HIP(ctx, t.get(bytes));
        KOMB_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(t.p, bytes, dk, dv, (long long)n, begin_bit, end_bit, ctx->stream));
    }
    *sorted_keys = dk.Current();
    *sorted_vals = dv.Current();
    return KOMB_OK;
}

// stable sort of (32-bit key, 32-bit value) pairs by the key bits [0, end_bit).  Asynchronous on the context's stream.
int prim_sort_pairs_u32_u32(komb_ctx *ctx, uint32_t *keys, uint32_t *keys_alt, uint32_t *vals, uint32_t *vals_alt,
                            int64_t n, int end_bit, uint32_t **sorted_keys, uint32_t **sorted_vals)
{
    hipcub::DoubleBuffer<uint32_t> dk(keys, keys_alt);
    hipcub::DoubleBuffer<uint32_t> dv(vals, vals_alt);
    if (n > 0) {
        size_t bytes = 0;
        KOMB_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dk, dv, (long long)n, 0, end_bit, ctx->stream));
        TempBuf t(ctx);
        KOMB_HIP(ctx, t.get(bytes));
        KOMB_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(t.p, bytes, dk, dv, (long long)n, 0, end_bit, ctx->stream));
    }
    *sorted_keys = dk.Current();
    *sorted_vals = dv.Current();
    return KOMB_OK;
}

// records of the graph build: 64-bit keys (internal source, internal target), 32-bit values (canonical edge id), sorted by
// the key bits [0, end_bit).  Asynchronous on the context's stream.
int prim_sort_pairs_u64_u32(komb_ctx *ctx, uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt,
                            int64_t n, int end_bit, uint64_t **sorted_keys, uint32_t **sorted_vals)
{
    hipcub::DoubleBuffer<uint64_t> dk(keys, keys_alt);
    hipcub::DoubleBuffer<uint32_t> dv(vals, vals_alt);
    if (n > 0) {
        size_t bytes = 0;
        KOMB_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dk, dv, (long long)n, 0, end_bit, ctx->stream));
        TempBuf t(ctx);
        KOMB_HIP(ctx, t.get(bytes));
        KOMB_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(t.p, bytes, dk, dv, (long long)n, 0, end_bit, ctx->stream));
    }
    *sorted_keys = dk.Current();
    *sorted_vals = dv.Current();
    return KOMB_OK;
}

} // namespace komb
