// truss_orient.h -- step 1 of the k-truss path (ktruss.hip): slot-parallel row filters.  The vertex-induced subgraph (a5,
// igraph_induced_subgraph_map, reference src/graph.cpp:502) is an ordered stream compaction of the resident oriented
// CSR's slots (the orientation itself comes with the graph: graph_build.hip).  Included by ktruss.hip only.
#pragma once

#include "peel_dev.h"

namespace komb {

namespace {

inline int grid_for(int64_t n, int per_block = kBlock, int cap = 256 * 16)
{
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ------------------------------------------------------------------ row filters
struct PredMask {                       // keep slot (a,b) when both endpoints are selected
    const uint8_t *mask;
    __device__ bool operator()(int32_t a, int32_t b) const { return mask[a] && mask[b]; }
};
// ------------------------------------------------------- slot-parallel filters
// A row filter (induced subgraph) keeps a subset of the CSR slots
// in slot order: a global ordered stream compaction.  Every slot knows its row
// through src[], so work is split by SLOTS, not rows -- a 134k-slot hub row is
// shared by dozens of workgroups instead of serialising one wavefront.
// Pass 1 counts the kept slots of each workgroup's chunk; an exclusive scan of
// the per-chunk counts gives chunk bases; pass 2 recomputes the predicate and
// writes (col, src) at base + block-local ordered prefix (ballot + popcount per
// wave, wave totals through LDS).  Row pointers of the result follow from the
// (sorted) src of the kept slots.
constexpr int kSlotsPerThread = 16;
constexpr int kChunkSlots = kBlock * kSlotsPerThread;          // slots per workgroup

template <class Pred, bool FILL>
__global__ __launch_bounds__(kBlock) void k_slot_filter(const int32_t *__restrict__ src, const int32_t *__restrict__ col,
                                                        int64_t ns, Pred pred, uint32_t *__restrict__ chunk_count,
                                                        const uint32_t *__restrict__ chunk_base,
                                                        int32_t *__restrict__ out_col, int32_t *__restrict__ out_src,
                                                        unsigned long long *__restrict__ keep_bits)
{
    // keep_bits: one bit per slot (64-slot words = one wavefront ballot).  Pass 1 evaluates the predicate
    // and records it; pass 2 only replays the bits (no second gather of the predicate's operands).
    __shared__ uint32_t sh_wave[kBlock / kWave];
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    const int64_t nchunks = (ns + kChunkSlots - 1) / kChunkSlots;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t c0 = chunk * kChunkSlots;
        uint32_t run = FILL ? chunk_base[chunk] : 0u;          // kept slots before the current 256-slot row of the chunk
        uint32_t total = 0;
        for (int r = 0; r < kSlotsPerThread; ++r) {
            const int64_t j = c0 + (int64_t)r * kBlock + threadIdx.x;
            const int64_t jw = j - lane;                        // first slot of this wave's 64 (multiple of 64)
            bool keep = false;
            if (!FILL) {
                if (j < ns) keep = pred(src[j], col[j]);
                const uint64_t m = __ballot(keep);
                if (lane == 0 && jw < ns) keep_bits[jw >> 6] = m;
                total += (uint32_t)__popcll(m);
                continue;
            }
            uint64_t m = 0;
            if (jw < ns) m = keep_bits[jw >> 6];
            keep = (m >> lane) & 1ull;
            const uint32_t wcnt = (uint32_t)__popcll(m);
            __syncthreads();
            if (lane == 0) sh_wave[w] = wcnt;
            __syncthreads();
            uint32_t before = 0, all = 0;
#pragma unroll
            for (int i = 0; i < kBlock / kWave; ++i) { const uint32_t x = sh_wave[i]; if (i < w) before += x; all += x; }
            if (keep) {
                const uint32_t o = run + before + (uint32_t)__popcll(m & lanemask_lt());
                out_col[o] = col[j];
                if (out_src) out_src[o] = src[j];
            }
            run += all;
        }
        if (!FILL) {                                            // per-wave totals -> chunk count
            __syncthreads();
            if (lane == 0) sh_wave[w] = total;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t t = 0;
                for (int i = 0; i < kBlock / kWave; ++i) t += sh_wave[i];
                chunk_count[chunk] = t;
            }
        }
    }
}

// kept slots before every 64-slot word of the keep bitmask (chunk bases + the words of the chunk before it):
// output position of a kept slot j = word_rank[j >> 6] + popcount(bits[j >> 6] below j)
__global__ __launch_bounds__(kBlock) void k_word_rank(const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ chunk_base,
                                                      int64_t nwords, uint32_t *__restrict__ word_rank)
{
    constexpr int kWordsPerChunk = kChunkSlots / 64;
    const int64_t nchunks = (nwords + kWordsPerChunk - 1) / kWordsPerChunk;
    for (int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * kBlock) {
        uint32_t r = chunk_base[c];
        const int64_t w1 = min(nwords, (c + 1) * kWordsPerChunk);
        for (int64_t w = c * kWordsPerChunk; w < w1; ++w) { word_rank[w] = r; r += (uint32_t)__popcll(bits[w]); }
    }
}

// row pointers of the kept slots straight from the keep bits: row v of the result starts at the number of kept slots before
// the first slot of row v of the input (no src[] of the kept slots needed)
__global__ __launch_bounds__(kBlock) void k_rowptr_from_bits(const uint32_t *__restrict__ rowptr_in, int64_t nv, int64_t ns,
                                                             const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ word_rank,
                                                             uint32_t kept, uint32_t *__restrict__ rowptr_out)
{
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v <= nv; v += (int64_t)gridDim.x * kBlock) {
        const int64_t j = v < nv ? (int64_t)rowptr_in[v] : ns;
        rowptr_out[v] = j < ns ? word_rank[j >> 6] + (uint32_t)__popcll(bits[j >> 6] & ((1ull << (j & 63)) - 1ull)) : kept;
    }
}

// row pointers of a CSR from the ascending src[] of its slots (gaps = empty rows)
__global__ __launch_bounds__(kBlock) void k_rowptr_from_src(const int32_t *__restrict__ src, int64_t ns, int64_t nv,
                                                            uint32_t *__restrict__ rowptr)
{
    if (ns == 0) {
        for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v <= nv; v += (int64_t)gridDim.x * kBlock) rowptr[v] = 0u;
        return;
    }
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < ns; j += (int64_t)gridDim.x * kBlock) {
        const int64_t a = src[j];
        const int64_t p = (j == 0) ? -1 : (int64_t)src[j - 1];
        for (int64_t v = p + 1; v <= a; ++v) rowptr[v] = (uint32_t)j;
        if (j == ns - 1)
            for (int64_t v = a + 1; v <= nv; ++v) rowptr[v] = (uint32_t)ns;
    }
}

// same result, one thread per ROW (binary search in src): used when the kept slots are few compared with
// the vertices, where the gap-filling form above would leave one thread to fill millions of empty rows
__global__ __launch_bounds__(kBlock) void k_rowptr_search(const int32_t *__restrict__ src, int64_t ns, int64_t nv,
                                                          uint32_t *__restrict__ rowptr)
{
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v <= nv; v += (int64_t)gridDim.x * kBlock) {
        int64_t lo = 0, hi = ns;                                // first slot with src >= v
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if ((int64_t)src[mid] < v) lo = mid + 1; else hi = mid;
        }
        rowptr[v] = (uint32_t)lo;
    }
}

} // namespace

// ordered compaction of the CSR slots (src,col)[ns] that satisfy pred -> (out_src,out_col)[n_out] + out_rowptr
template <class Pred>
static int compact_slots(komb_ctx *ctx, DevBufs &bufs, const int32_t *src, const int32_t *col, int64_t ns, int64_t nv, Pred pred,
                         uint32_t *out_rowptr, int32_t **out_col, int32_t **out_src, int64_t *n_out,
                         unsigned long long **keep_bits_out = nullptr, uint32_t **word_rank_out = nullptr,
                         const uint32_t *rowptr_in = nullptr)
{
    // out_src == nullptr (with rowptr_in = the input's row pointers and word_rank_out): the kept slots' rows are not written
    // out, the result's row pointers come from the keep bits (the orientation: 0.4 GB of stores and a pass over them less)
    hipStream_t s = ctx->stream;
    const int64_t nchunks = (ns + kChunkSlots - 1) / kChunkSlots;
    uint32_t *d_cc = nullptr, *d_cb = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_cc, (size_t)nchunks + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_cb, (size_t)nchunks + 1));
    unsigned long long *d_bits = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_bits, (size_t)(ns + 63) / 64 + 1));
    KOMB_HIP(ctx, hipMemsetAsync(d_cc, 0, ((size_t)nchunks + 1) * sizeof(uint32_t), s));
    const int g = grid_for(nchunks, 1, 256 * 32);
    k_slot_filter<Pred, false><<<g, kBlock, 0, s>>>(src, col, ns, pred, d_cc, nullptr, nullptr, nullptr, d_bits);
    KOMB_TRY(prim_exclusive_sum_u32(ctx, d_cc, d_cb, nchunks + 1));
    uint32_t kept = 0;
    KOMB_HIP(ctx, d2h(ctx, &kept, d_cb + nchunks, sizeof(uint32_t)));
    KOMB_HIP(ctx, bufs.alloc(out_col, (size_t)kept + 8));        // + 8: the triangle enumeration reads 16 bytes at a time, past the end of the last row
    if (out_src) KOMB_HIP(ctx, bufs.alloc(out_src, (size_t)kept));
    else if (!rowptr_in || !word_rank_out) KOMB_FAIL(ctx, KOMB_ERR_ARG, "compact_slots: no rows of the kept slots and no row pointers to derive them from");
    if (word_rank_out) {
        const int64_t nwords = (ns + 63) / 64;
        KOMB_HIP(ctx, bufs.alloc(word_rank_out, (size_t)nwords + 1));
        k_word_rank<<<grid_for((nwords + kChunkSlots / 64 - 1) / (kChunkSlots / 64)), kBlock, 0, s>>>(d_bits, d_cb, nwords, *word_rank_out);
    }
    k_slot_filter<Pred, true><<<g, kBlock, 0, s>>>(src, col, ns, pred, nullptr, d_cb, *out_col, out_src ? *out_src : nullptr, d_bits);
    if (!out_src) k_rowptr_from_bits<<<grid_for(nv + 1), kBlock, 0, s>>>(rowptr_in, nv, ns, d_bits, *word_rank_out, kept, out_rowptr);
    else if ((int64_t)kept * 4 < nv) k_rowptr_search<<<grid_for(nv + 1), kBlock, 0, s>>>(*out_src, (int64_t)kept, nv, out_rowptr);
    else k_rowptr_from_src<<<grid_for(kept), kBlock, 0, s>>>(*out_src, (int64_t)kept, nv, out_rowptr);
    bufs.release(d_cc); bufs.release(d_cb);
    if (keep_bits_out) *keep_bits_out = d_bits; else bufs.release(d_bits);
    *n_out = (int64_t)kept;
    return KOMB_OK;
}

} // namespace komb
