// truss_orient.h -- step 1 of the k-truss path (ktruss.hip): slot-parallel row filters.  The vertex-induced subgraph (a5,
// igraph_induced_subgraph_map, reference src/graph.cpp:502) and the (degree,id) orientation are both ordered stream
// compactions of the CSR slots; the orientation's predicate reads degrees through small tables that stay in the L2s.
// Included by ktruss.hip only.
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
struct PredOrient {                     // keep slot (a,b) when a precedes b in (degree,id) order
    const int32_t *deg;
    const uint8_t *deg8;                // min(deg, 255): a 1-byte-per-vertex table that the L2s (small graphs) or the Infinity
                                        // Cache hold; exact degrees only when both endpoints saturate
    __device__ bool operator()(int32_t a, int32_t b) const
    {
        int32_t da = deg8[a], db = deg8[b];
        if (da == 255 && db == 255) { da = deg[a]; db = deg[b]; }
        return da < db || (da == db && a < b);
    }
};
struct PredOrientClass {                // the same order, for graphs whose 1-byte table does not fit the L2s
    const int32_t *deg;
    const uint8_t *deg8;
    const uint32_t *deg2;               // a 2-bit degree class per vertex, 16 vertices per word: |V| / 4 bytes (2.5 MB for 10 M
                                        // vertices) stay in every XCD's L2.  The class is a monotone function of the degree
                                        // (thresholds t1 <= t2 <= t3 at the quartiles of the slots' endpoint degrees), so two
                                        // different classes decide the order and only equal classes -- about a third of the
                                        // slots -- go on to the 1-byte table: the one random gather per slot mostly ends in L2.
    int32_t t1, t2, t3;
    __device__ bool operator()(int32_t a, int32_t b) const
    {
        int32_t da = deg8[a];                                    // (row-local: the wavefront's slots share a few rows)
        const int32_t ca = (da > t1) + (da > t2) + (da > t3);
        const int32_t cb = (int32_t)((deg2[(uint32_t)b >> 4] >> (((uint32_t)b & 15u) * 2u)) & 3u);
        if (ca != cb) return ca < cb;
        int32_t db = deg8[b];
        if (da == 255 && db == 255) { da = deg[a]; db = deg[b]; }
        return da < db || (da == db && a < b);
    }
};

// ------------------------------------------------------- slot-parallel filters
// A row filter (induced subgraph, orientation) keeps a subset of the CSR slots
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
                                                        unsigned long long *__restrict__ keep_bits,
                                                        unsigned long long *__restrict__ keep_upper_bits,
                                                        uint32_t *__restrict__ upper_cnt)
{
    // upper_cnt (nullable, pass 1, with keep_upper_bits): upper slots (column above row) per 64-slot word, kept or not: their
    // prefix sum is the canonical edge id of a word's first upper slot
    // keep_upper_bits (nullable, pass 1): the kept slots whose column is above their row -- the canonical (u < v) copies
    // that are also the oriented copies; the result gather ranks the others through them
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
                bool up = false;
                bool upper = false;
                if (j < ns) { const int32_t a = src[j], b = col[j]; keep = pred(a, b); upper = b > a; up = keep && upper; }
                const uint64_t m = __ballot(keep);
                if (keep_upper_bits) {
                    const uint64_t mu = __ballot(up), ma = __ballot(upper);
                    if (lane == 0 && jw < ns) { keep_upper_bits[jw >> 6] = mu; upper_cnt[jw >> 6] = (uint32_t)__popcll(ma); }
                }
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

// degrees, their 1-byte copies, and hist[d] = slots whose row has degree min(d, 255) (the distribution of the slots'
// endpoint degrees: the orientation's class thresholds are its quartiles)
__global__ __launch_bounds__(kBlock) void k_degree(const uint32_t *__restrict__ rowptr, int64_t nv, int32_t *__restrict__ deg,
                                                   uint8_t *__restrict__ deg8, unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t sh_h[256];
    sh_h[threadIdx.x] = 0u;                                     // kBlock == 256
    __syncthreads();
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += (int64_t)gridDim.x * kBlock) {
        const int32_t d = (int32_t)(rowptr[v + 1] - rowptr[v]);
        deg[v] = d;
        deg8[v] = (uint8_t)min(d, 255);
        if (d) atomicAdd(&sh_h[min(d, 255)], (uint32_t)d);      // (a workgroup's rows hold fewer than 2^32 slots: the CSR does)
    }
    __syncthreads();
    if (sh_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)sh_h[threadIdx.x]);
}

__global__ __launch_bounds__(kBlock) void k_degree_classes(const uint8_t *__restrict__ deg8, int64_t nv, int32_t t1, int32_t t2, int32_t t3,
                                                           uint32_t *__restrict__ deg2)
{
    const int64_t nw = (nv + 15) / 16;
    for (int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x; w < nw; w += (int64_t)gridDim.x * kBlock) {
        uint32_t word = 0;
        for (int k = 0; k < 16; ++k) {
            const int64_t v = w * 16 + k;
            if (v < nv) { const int32_t d = deg8[v]; word |= (uint32_t)((d > t1) + (d > t2) + (d > t3)) << (2 * k); }
        }
        deg2[w] = word;
    }
}

} // namespace

// ordered compaction of the CSR slots (src,col)[ns] that satisfy pred -> (out_src,out_col)[n_out] + out_rowptr
template <class Pred>
static int compact_slots(komb_ctx *ctx, DevBufs &bufs, const int32_t *src, const int32_t *col, int64_t ns, int64_t nv, Pred pred,
                         uint32_t *out_rowptr, int32_t **out_col, int32_t **out_src, int64_t *n_out,
                         unsigned long long **keep_bits_out = nullptr, uint32_t **word_rank_out = nullptr,
                         unsigned long long **keep_upper_out = nullptr, uint32_t **upper_cnt_out = nullptr,
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
    unsigned long long *d_kub = nullptr;
    uint32_t *d_ucw = nullptr;
    if (keep_upper_out) {
        KOMB_HIP(ctx, bufs.alloc(&d_kub, (size_t)(ns + 63) / 64 + 1));
        KOMB_HIP(ctx, bufs.alloc(&d_ucw, (size_t)(ns + 63) / 64 + 2));
        KOMB_HIP(ctx, hipMemsetAsync(d_ucw + (ns + 63) / 64, 0, 2 * sizeof(uint32_t), s));
    }
    k_slot_filter<Pred, false><<<g, kBlock, 0, s>>>(src, col, ns, pred, d_cc, nullptr, nullptr, nullptr, d_bits, d_kub, d_ucw);
    if (keep_upper_out) { *keep_upper_out = d_kub; *upper_cnt_out = d_ucw; }
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
    k_slot_filter<Pred, true><<<g, kBlock, 0, s>>>(src, col, ns, pred, nullptr, d_cb, *out_col, out_src ? *out_src : nullptr, d_bits, nullptr, nullptr);
    if (!out_src) k_rowptr_from_bits<<<grid_for(nv + 1), kBlock, 0, s>>>(rowptr_in, nv, ns, d_bits, *word_rank_out, kept, out_rowptr);
    else if ((int64_t)kept * 4 < nv) k_rowptr_search<<<grid_for(nv + 1), kBlock, 0, s>>>(*out_src, (int64_t)kept, nv, out_rowptr);
    else k_rowptr_from_src<<<grid_for(kept), kBlock, 0, s>>>(*out_src, (int64_t)kept, nv, out_rowptr);
    bufs.release(d_cc); bufs.release(d_cb);
    if (keep_bits_out) *keep_bits_out = d_bits; else bufs.release(d_bits);
    *n_out = (int64_t)kept;
    return KOMB_OK;
}

} // namespace komb
