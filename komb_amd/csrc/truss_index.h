// truss_index.h -- step 2b of the k-truss path (ktruss.hip): the incidence index built from the enumeration's record stream
// (DESIGN.md section 4.2a): records sorted by bin, one workgroup per bin counts (k_bin_count) and finishes (k_bin_finish)
// its up to 2048 edges out of LDS -- supports, slice offsets, the index window, the peel's initial state and the
// first level's frontier.  Included by ktruss.hip only, after truss_tri.h.
#pragma once

#include "peel_dev.h"

namespace komb {

namespace {

// ---- the record stream of the single pass (k_triangles, STREAM): destination-binned build of the index
// A BIN is a set of up to 2^kBinBits edges whose slices form one contiguous WINDOW of the index.  Internal edge ids follow
// the (degree,id) order of their source vertices (graph_build.hip), so the edges with the longest slices -- the hubs' -- are
// neighbours: 2048 CONSECUTIVE edges per bin would leave the last bins with hundreds of times the mean (C2: 391 k entries
// against 5.9 k) and one workgroup with all of it.  A bin therefore takes CHUNKS of 2^kChunkBits consecutive edges, dealt
// round-robin: chunk c goes to bin c mod NB (NB a power of two), so every bin holds a sample of the whole degree range,
//     bin(e)   = (e >> kChunkBits) & (NB - 1)
//     local(e) = (e >> (kChunkBits + log2 NB)) << kChunkBits | (e & (2^kChunkBits - 1))          < 2^kBinBits
// and the index is laid out bin after bin (an edge's slice is found through its (start, length) pair, not through its
// neighbour's start).  The records are radix-sorted by the bin field of their key only -- 16 of C3's 27 bits, two radix
// passes --: records of one bin become contiguous, in no particular order inside it.  One workgroup then finishes a bin
// out of LDS: a histogram of the bin's keys gives every edge its record count (k_bin_count); after the scan of the
// supports, per-edge write cursors in LDS place every record value in its edge's slice (k_bin_finish) -- LDS atomics and
// stores inside one window of the index, instead of one global atomic and one scattered HBM line per triangle.  Reads and
// writes of per-edge arrays stay coalesced: a chunk is 64 consecutive edges.
constexpr int kBinBits = 11;
constexpr uint32_t kBinEdges = 1u << kBinBits;
#ifndef KOMB_CHUNK_BITS
#define KOMB_CHUNK_BITS 6
#endif
constexpr int kChunkBits = KOMB_CHUNK_BITS;
constexpr uint32_t kChunkMask = (1u << kChunkBits) - 1u;
struct BinGeom {
    int nb_bits;                          // log2 of the number of bins
    uint32_t nb;                          // bins (a power of two: >= ceil(m / 2^kBinBits))
    __host__ __device__ uint32_t bin_of(uint32_t e) const { return (e >> kChunkBits) & (nb - 1u); }
    __host__ __device__ uint32_t local_of(uint32_t e) const { return ((e >> (kChunkBits + nb_bits)) << kChunkBits) | (e & kChunkMask); }
    __host__ __device__ uint64_t edge_of(uint32_t b, uint32_t i) const
    {
        return ((((uint64_t)(i >> kChunkBits) << nb_bits) | b) << kChunkBits) | (i & kChunkMask);
    }
    // keys of the unused record positions: above every edge id (local index 2^kBinBits), spread over the bins
    __host__ __device__ uint32_t sentinel_base() const { return (uint32_t)(((uint64_t)kBinEdges >> kChunkBits) << (kChunkBits + nb_bits)); }
};
inline BinGeom bin_geom(int64_t m)
{
    BinGeom g{0, 1u};
    while (((int64_t)g.nb << kBinBits) < m) { ++g.nb_bits; g.nb <<= 1; }
    return g;
}

// boff[b] = first sorted record whose bin is >= b, for b = 0 .. nb.  One thread per bin, binary search: no pass over the keys.
__global__ __launch_bounds__(kBlock) void k_bin_offsets(const uint32_t *__restrict__ key, int64_t n, BinGeom g, uint32_t *__restrict__ boff)
{
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b <= (int64_t)g.nb; b += (int64_t)gridDim.x * kBlock) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if ((int64_t)g.bin_of(key[mid]) < b) lo = mid + 1; else hi = mid;
        }
        boff[b] = (uint32_t)lo;
    }
}

__global__ __launch_bounds__(kBlock) void k_count_mismatch(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, int64_t n,
                                                           unsigned long long *__restrict__ bad)
{
    unsigned long long t = 0;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) t += a[e] != b[e] ? 1ull : 0ull;
    block_add_u64(t, bad);
}

// supports of a bin's edges: dense own-role entries + records of the bin with that key (LDS histogram); per-bin totals
// (their scan gives every bin its window of the index) and the 64-bit grand total on the side.
__global__ __launch_bounds__(kBlock) void k_bin_count(const uint32_t *__restrict__ key, const uint32_t *__restrict__ boff, BinGeom g,
                                                      const uint32_t *__restrict__ own, int64_t m, uint32_t *__restrict__ sum,
                                                      uint32_t *__restrict__ bin_total, unsigned long long *__restrict__ total,
                                                      int32_t *__restrict__ min_pos)
{
    // min_pos: the smallest positive support = the peel's first level (k_bin_finish queues its frontier)
    __shared__ uint32_t sh_cnt[kBinEdges];
    __shared__ uint32_t sh_part[kBlock / kWave];
    unsigned long long t = 0;
    int32_t lmin = 0x7FFFFFFF;
    for (uint32_t b = blockIdx.x; b < g.nb; b += gridDim.x) {
        for (uint32_t i = threadIdx.x; i < kBinEdges; i += kBlock) sh_cnt[i] = 0u;
        __syncthreads();
        const uint32_t r0 = boff[b], r1 = boff[b + 1];
        for (uint32_t r = r0 + threadIdx.x; r < r1; r += kBlock) {
            const uint32_t k = key[r];
            if ((int64_t)k < m) atomicAdd(&sh_cnt[g.local_of(k)], 1u);      // (the rest: keys of unused record positions)
        }
        __syncthreads();
        uint32_t tb = 0;
        for (uint32_t i = threadIdx.x; i < kBinEdges; i += kBlock) {
            const uint64_t e = g.edge_of(b, i);
            if (e >= (uint64_t)m) continue;
            const uint32_t c = own[e] + sh_cnt[i];
            sum[e] = c;
            tb += c;
            if (c) lmin = min(lmin, (int32_t)c);
        }
        tb = wave_sum(tb);
        if (lane_id() == 0) sh_part[threadIdx.x >> 6] = tb;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t all = 0;
            for (int i = 0; i < kBlock / kWave; ++i) all += sh_part[i];
            bin_total[b] = all;                                  // (a bin holds 2048 edges: their supports sum to far less than 2^32 ... unless the graph is beyond the index limit, which the 64-bit total reports)
            t += all;
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { sum[m] = 0u; bin_total[g.nb] = 0u; }
    if (threadIdx.x == 0 && t) atomicAdd(total, t);
    lmin = wave_min(lmin);
    if (lane_id() == 0 && lmin != 0x7FFFFFFF) atomicMin(min_pos, lmin);
}

// Dense index of a bin's edges.  An edge's slice is [its records' values | its own-role entries]; the slices of a
// bin's edges (in the order of their local indices) are one contiguous WINDOW of the index (~40 KB).  One workgroup per bin assembles
// the window in LDS -- every record takes its position from a per-edge cursor (LDS atomic), the own-role entries are
// copied out of the tasks' dense blocks -- and then writes it as one coalesced stream: no global atomic, no scattered
// store, every line of the index written whole, once.  A window that does not fit the LDS buffer (hub edges) is written
// in place instead.  All loads of a phase are issued before the first is used: two workgroups per CU, and the kernel
// lives on memory-level parallelism.
constexpr int kFinBlock = 512;
#ifndef KOMB_FIN_PER_CU
#define KOMB_FIN_PER_CU 2
#endif
#ifndef KOMB_WIN_CAP
#define KOMB_WIN_CAP 7168
#endif
constexpr int kFinPerCu = KOMB_FIN_PER_CU;          // finishing workgroups per CU
constexpr uint32_t kWinCap = KOMB_WIN_CAP;          // entries of the LDS window (7168: 56 KB; with the two 8 KB tables: 2 workgroups per CU)
constexpr int kFinE = (int)(kBinEdges / kFinBlock);          // consecutive edges per thread
constexpr int kFinU = 4;                           // records per thread per trip
static_assert((kWinCap * 8 + 2 * kBinEdges * 4 + 512) * kFinPerCu <= 160 * 1024, "the finishing workgroups of a CU: windows and tables are sized for gfx950's 160 KB of LDS");
static_assert(kWinCap * 8 >= (2 * kBinEdges + 4) * 4 + kBinEdges * 8, "the window buffer doubles as the table of the in-place copy");
static_assert(kFinE == 4 && kFinE <= (1 << kChunkBits), "a thread's edges are 4 consecutive ones of a chunk: their supports and own-role counts are one 16-byte vector each");
// The kernel also does what followed the index build: the slices' (start, length) pairs off2[] (a workgroup scan of the bin's supports on
// top of the bin's base -- the 100 M-element device scan is gone) and the peel's initial state (support, alive marker or
// "gone" for a triangle-free edge, the count of those and the smallest positive support for the first level).
__global__ __launch_bounds__(kFinBlock, (kFinBlock / 64) * kFinPerCu / 4) void k_bin_finish(const uint32_t *__restrict__ key, const int2 *__restrict__ val,
                                                         const uint32_t *__restrict__ boff, BinGeom g,
                                                         const uint32_t *__restrict__ own, const uint32_t *__restrict__ cnt,
                                                         const uint32_t *__restrict__ bin_base,
                                                         const int2 *__restrict__ own_dense, const unsigned long long *__restrict__ ownoff,
                                                         int2 *__restrict__ dense, int64_t m,
                                                         uint2 *__restrict__ off2, int32_t *__restrict__ sup, int32_t *__restrict__ stamp, uint8_t *__restrict__ st8,
                                                         uint32_t *__restrict__ init, int32_t *__restrict__ light0)
{
    // init[0] += triangle-free edges; init[1] = the smallest positive support (from k_bin_count) = the peel's first level L1.
    // The edges with support L1 ARE that level's first frontier (nothing has been decremented yet): they are stamped with
    // round 1 (its level, hence their trussness, is recorded when that sub-round ends) and appended to light queue 0 here (one reservation per bin on init[2]), so the peel starts
    // with a PROCESS step instead of a dense SCAN of every edge (0.55 ms at C3).  Only when they are light units (L1 <= kLight).
    __shared__ uint32_t sh_off[kBinEdges + 4];     // slice offsets relative to the window
    __shared__ uint32_t sh_cur[kBinEdges];
    __shared__ uint32_t sh_wsum[kFinBlock / kWave];
    __shared__ int2 sh_win[kWinCap];
    __shared__ uint32_t sh_hsum[kFinBlock / kWave];
    __shared__ uint32_t sh_qbase;
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    uint32_t zeros = 0;
    const int32_t L1 = (int32_t)init[1];
    const bool queue_first = light0 != nullptr && L1 <= kLight;
    for (uint32_t b = blockIdx.x; b < g.nb; b += gridDim.x) {
        const uint64_t r0 = boff[b], r1 = boff[b + 1];
        const uint32_t base = bin_base[b];
        // ---- the thread's 4 consecutive edges (one chunk holds them all): supports and own-role counts, then (for the edges that have some) where their blocks are
        const uint32_t i0 = threadIdx.x * (uint32_t)kFinE;
        const uint64_t e0 = g.edge_of(b, i0);
        const uint32_t nin = e0 >= (uint64_t)m ? 0u : (uint32_t)min((uint64_t)kFinE, (uint64_t)m - e0);     // the thread's edges that exist
        uint32_t c[kFinE], ow[kFinE];
        unsigned long long oo[kFinE];
        if (nin == (uint32_t)kFinE) {
            const uint4 v = *reinterpret_cast<const uint4 *>(cnt + e0);
            const uint4 q = *reinterpret_cast<const uint4 *>(own + e0);
            c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
            ow[0] = q.x; ow[1] = q.y; ow[2] = q.z; ow[3] = q.w;
        } else {
#pragma unroll
            for (int u = 0; u < kFinE; ++u) { const bool in = (uint32_t)u < nin; c[u] = in ? cnt[e0 + u] : 0u; ow[u] = in ? own[e0 + u] : 0u; }
        }
#pragma unroll
        for (int u = 0; u < kFinE; ++u) oo[u] = ow[u] ? ownoff[e0 + u] : 0ull;
        // slice offsets: exclusive scan of the supports over the workgroup
        const uint32_t mine = c[0] + c[1] + c[2] + c[3];
        const uint32_t incl = wave_incl_scan(mine);
        uint32_t hits = 0;
#pragma unroll
        for (int u = 0; u < kFinE; ++u) hits += (queue_first && (int32_t)c[u] == L1) ? 1u : 0u;
        const uint32_t hincl = wave_incl_scan(hits);
        if (lane == kWave - 1) { sh_wsum[w] = incl; sh_hsum[w] = hincl; }
        __syncthreads();
        uint32_t before = 0, hbefore = 0, hall = 0;
#pragma unroll
        for (int i = 0; i < kFinBlock / kWave; ++i) { before += i < w ? sh_wsum[i] : 0u; hbefore += i < w ? sh_hsum[i] : 0u; hall += sh_hsum[i]; }
        if (threadIdx.x == 0 && hall) sh_qbase = atomicAdd(&init[2], hall);
        uint32_t o[kFinE + 1];                                      // relative to the window
        o[0] = before + incl - mine;
#pragma unroll
        for (int u = 0; u < kFinE; ++u) o[u + 1] = o[u] + c[u];
#pragma unroll
        for (int u = 0; u < kFinE; ++u) { sh_off[i0 + u] = o[u]; sh_cur[i0 + u] = o[u]; }
        if (threadIdx.x == kFinBlock - 1) sh_off[kBinEdges] = o[kFinE];      // the bin's total (edges that do not exist count 0)
        // ... written out, with the peel's initial state
        if (nin == (uint32_t)kFinE) {
            reinterpret_cast<uint4 *>(off2 + e0)[0] = make_uint4(base + o[0], c[0], base + o[1], c[1]);
            reinterpret_cast<uint4 *>(off2 + e0)[1] = make_uint4(base + o[2], c[2], base + o[3], c[3]);
            int4 sv, mv;
            int32_t *svp = &sv.x, *mvp = &mv.x;
#pragma unroll
            for (int u = 0; u < kFinE; ++u) {
                const bool first = queue_first && (int32_t)c[u] == L1;
                svp[u] = (int32_t)c[u];
                mvp[u] = first ? 1 : (c[u] ? alive_marker(c[u]) : 0);   // round 1: the first frontier; round 0 (trussness 2): gone before the first sub-round
                if (!c[u]) ++zeros;
            }
            *reinterpret_cast<int4 *>(sup + e0) = sv;
            *reinterpret_cast<int4 *>(stamp + e0) = mv;
            *reinterpret_cast<uchar4 *>(st8 + e0) = make_uchar4(state_of_stamp(mv.x), state_of_stamp(mv.y), state_of_stamp(mv.z), state_of_stamp(mv.w));
        } else {
#pragma unroll
            for (int u = 0; u < kFinE; ++u) if ((uint32_t)u < nin) {
                const uint64_t e = e0 + u;
                const bool first = queue_first && (int32_t)c[u] == L1;
                off2[e] = make_uint2(base + o[u], c[u]);
                sup[e] = (int32_t)c[u];
                stamp[e] = first ? 1 : (c[u] ? alive_marker(c[u]) : 0);
                st8[e] = state_of_stamp(stamp[e]);
                if (!c[u]) ++zeros;
            }
        }
        __syncthreads();
        if (hall) {                                                 // the bin's part of the first frontier, in edge order
            uint32_t q = sh_qbase + hbefore + hincl - hits;
#pragma unroll
            for (int u = 0; u < kFinE; ++u) if (queue_first && (int32_t)c[u] == L1 && (uint32_t)u < nin) light0[q++] = (int32_t)(e0 + u);
        }
        const uint32_t W = sh_off[kBinEdges];
        const bool inwin = W <= kWinCap;                            // (workgroup-uniform)
        // ---- records
        for (uint64_t r = r0 + threadIdx.x; r < r1; r += (uint64_t)kFinBlock * kFinU) {
            uint32_t k[kFinU];
            int2 v[kFinU];
#pragma unroll
            for (int u = 0; u < kFinU; ++u) {
                const uint64_t rr = r + (uint64_t)u * kFinBlock;
                if (rr < r1) { k[u] = key[rr]; v[u] = val[rr]; }
            }
#pragma unroll
            for (int u = 0; u < kFinU; ++u) {
                const uint64_t rr = r + (uint64_t)u * kFinBlock;
                if (rr < r1 && (int64_t)k[u] < m) {                   // (the rest: keys of unused record positions)
                    const uint32_t p = atomicAdd(&sh_cur[g.local_of(k[u])], 1u);
                    if (inwin) sh_win[p] = v[u]; else dense[base + p] = v[u];
                }
            }
        }
        // ---- own-role entries of the thread's edges: they end the edges' slices
        if (inwin) {
            // the 4 edges' copies advance together (4 independent loads per trip), into the LDS window
            uint32_t most = 0;
#pragma unroll
            for (int u = 0; u < kFinE; ++u) most = max(most, ow[u]);
            for (uint32_t kk = 0; kk < most; ++kk) {
                int2 t[kFinE];
#pragma unroll
                for (int u = 0; u < kFinE; ++u) if (kk < ow[u]) t[u] = own_dense[oo[u] + kk];
#pragma unroll
                for (int u = 0; u < kFinE; ++u) if (kk < ow[u]) sh_win[o[u + 1] - ow[u] + kk] = t[u];
            }
        } else {
            // A window beyond the LDS buffer holds a chunk of hub edges (runs of hundreds of entries per edge).  A thread
            // copying its own edges' runs would store 8 bytes at 512 unrelated addresses per trip; instead the workgroup
            // flattens all (edge, k) pairs: consecutive threads copy consecutive entries of one run -- reads from the task's
            // block and writes into the window are both whole lines.  The (unused) window buffer holds the table: per edge the
            // entries before it, where its run starts in the window, and where its block is.
            uint32_t *tab_pre = reinterpret_cast<uint32_t *>(sh_win), *tab_dst = tab_pre + kBinEdges + 4;
            unsigned long long *tab_src = reinterpret_cast<unsigned long long *>(tab_dst + kBinEdges);
            const uint32_t mine_ow = ow[0] + ow[1] + ow[2] + ow[3];
            const uint32_t incl_ow = wave_incl_scan(mine_ow);
            __syncthreads();                                        // (sh_wsum is free again: every thread has read its offsets)
            if (lane == kWave - 1) sh_wsum[w] = incl_ow;
            __syncthreads();
            uint32_t before_ow = 0, total_ow = 0;
#pragma unroll
            for (int i = 0; i < kFinBlock / kWave; ++i) { before_ow += i < w ? sh_wsum[i] : 0u; total_ow += sh_wsum[i]; }
            uint32_t run = before_ow + incl_ow - mine_ow;
#pragma unroll
            for (int u = 0; u < kFinE; ++u) {
                tab_pre[i0 + u] = run; tab_dst[i0 + u] = o[u + 1] - ow[u]; tab_src[i0 + u] = oo[u];
                run += ow[u];
            }
            if (threadIdx.x == kFinBlock - 1) tab_pre[kBinEdges] = run;
            __syncthreads();
            for (uint32_t x = threadIdx.x; x < total_ow; x += kFinBlock) {
                uint32_t lo = 0;                                    // the edge whose run holds entry x: largest i with tab_pre[i] <= x
#pragma unroll
                for (uint32_t st = kBinEdges / 2; st > 0; st >>= 1) lo += (tab_pre[lo + st] <= x) ? st : 0u;
                const uint32_t kk = x - tab_pre[lo];
                dense[base + tab_dst[lo] + kk] = own_dense[tab_src[lo] + kk];
            }
        }
        __syncthreads();
        // ---- the window, as a stream
        if (inwin) for (uint32_t j = threadIdx.x; j < W; j += kFinBlock) dense[base + j] = sh_win[j];
        __syncthreads();
    }
    block_add_min(zeros, 0x7FFFFFFF, &init[0], (int32_t *)&init[1]);
}

// (start, length) pairs from the monotone offsets of the other index layouts (bounded slices, exact two-pass), and the peel
// state from the slice lengths.  Triangle-free edges are peeled here (stamp 0 = sub-round 0 = trussness 2); init[0] counts
// them and init[1] receives the smallest positive support = the first populated level.
__global__ __launch_bounds__(kBlock) void k_peel_init(int64_t m, const uint32_t *__restrict__ off, uint2 *__restrict__ off2,
                                                      int32_t *__restrict__ sup, int32_t *__restrict__ stamp, uint8_t *__restrict__ st8,
                                                      uint32_t *__restrict__ init)
{
    uint32_t zeros = 0;
    int32_t lmin = 0x7FFFFFFF;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < m; e += (int64_t)gridDim.x * kBlock) {
        const uint32_t b0 = off[e];
        const int32_t s0 = (int32_t)(off[e + 1] - b0);
        off2[e] = make_uint2(b0, (uint32_t)s0);
        sup[e] = s0;
        if (s0 == 0) { stamp[e] = 0; st8[e] = (uint8_t)ST_GONE; ++zeros; }                     // round 0 (level 0, trussness 2): gone before the first sub-round
        else { const int32_t a = alive_marker((uint32_t)s0); stamp[e] = a; st8[e] = state_of_stamp(a); lmin = min(lmin, s0); }
    }
    block_add_min(zeros, lmin, &init[0], (int32_t *)&init[1]);
}

} // namespace

} // namespace komb
