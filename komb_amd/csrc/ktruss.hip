// ktruss.hip -- rows a5 + a6 of the hot-path table: per-edge trussness of the
// (optionally vertex-induced) simple graph, replacing
// igraph_induced_subgraph_map + igraph_trussness (reference src/graph.cpp:502,
// src/graph.cpp:508).  Trussness(e) = 2 + the support level at which e is
// peeled; a triangle-free edge has trussness 2.
//
// MI355X-first design (no intersections inside the peel loop):
//   1. orient every edge from the lower to the higher (degree,id) endpoint: an
//      ordered stream compaction of the CSR slots (k_slot_filter) gives the
//      oriented CSR (orow/ocol, rows still ascending by id).  The internal edge
//      id is the oriented slot.  On power-law unitig graphs the oriented rows
//      are tiny (max ~10^2), whatever the hub degrees are.
//   2. enumerate every triangle once (k_triangles: edge a->b, every element of
//      N+(b) looked up in the LDS-staged N+(a)) and build the incidence index:
//      for every edge, the pairs of the other two edges of its triangles
//      (24 bytes per triangle).  Default (round 3): ONE enumeration; the entries
//      of a task's own edges leave it as a dense block, every other entry as a
//      12-byte record of one stream -- no atomic, no scattered store per
//      triangle; the records are radix-sorted by BIN (2048 consecutive edges)
//      and one workgroup per bin assembles its window of the index in LDS
//      (k_bin_count, k_bin_finish), which also writes the slice offsets, the
//      peel's initial state and the first level's frontier.  Round 2's bounded
//      slices (KOMB_INDEX=slices) and the exact count-scan-fill two-pass layout
//      (KOMB_INDEX=two_pass, the fallback) are kept and tested.
//   3. peel: level-synchronous sub-rounds driven by the device control block
//      (peel_dev.h).  A frontier edge walks its incidence slice; a triangle
//      whose other two edges are both still present loses one support on each
//      (ties between two frontier edges are broken by edge id so every
//      triangle is destroyed exactly once); the decrement that lands an edge
//      exactly on the level enqueues it for the next sub-round.  The peel
//      never touches the adjacency again.
//   4. gather results into canonical (min,max)-lexicographic edge order with
//      ORIGINAL vertex ids -- the identity the C ABI promises.  No search: the
//      oriented copies that sit in the other endpoint's row travel through a
//      stable radix sort by target id, after which every access is a stream.
#include "peel_dev.h"
#include "truss_tail.h"
#include "local_dev.h"
#include "shard_dev.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

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
                out_src[o] = src[j];
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

// ------------------------------------------------------ triangle enumeration
// Every triangle {a,b,w}, a -> b -> w in (degree,id) order, is found exactly
// once, from its oriented edge e = (a->b), as a common out-neighbour w of a and
// b.  With i = slot of w in row a and j = slot of w in row b the triangle is
// {e, i, j} in internal edge ids.
//
// One wavefront owns kTriV consecutive source vertices.  Their oriented rows
// are one contiguous range [S0,S1) of ocol, staged in LDS together with one
// counter per slot.  The probe items -- every element of N+(b) for every owned
// edge (a->b) -- are cut into chunks of 4 consecutive elements of one row, the
// chunks are flattened over the 64 lanes (prefix sum + binary search in LDS);
// each item is looked up in the staged row of a (row signature first, then a
// binary search in LDS for the survivors).
// Of a triangle's three edges, e and i belong to the owned rows, so their
// counts / write cursors are LDS atomics private to the wave; only j needs a
// global atomic.  Hits are rare (~4% of the probes): they are parked in an LDS
// buffer and handled densely, 64 triangles at a time.
//   TRI_COUNT  counts supports (own[] by plain stores, other[] by atomics).
//   TRI_SINGLE writes each edge's incidence pairs into its slice
//              [off[x], off[x+1]): own-role entries from the front (LDS
//              cursor), third-role entries from the back (global counter).  The
//              slices are either capacity-bounded (no counting pass at all;
//              k_compact_inc then packs them) or exact (after TRI_COUNT + scan:
//              the two ends meet precisely).
// Tasks whose rows exceed the LDS budget fall back to global binary search and
// global atomics for all three roles.
#ifndef KOMB_TRI_CAP
#define KOMB_TRI_CAP 256
#endif
#ifndef KOMB_TRI_EU
#define KOMB_TRI_EU 4
#endif
#ifndef KOMB_TRI_CAND
#define KOMB_TRI_CAND 128
#endif
#ifndef KOMB_TRI_V
#define KOMB_TRI_V 16
#endif
constexpr int kTriV = KOMB_TRI_V;               // consecutive source vertices per task (<= 63: lane l holds orow[v0 + l])
constexpr int kTriCap = KOMB_TRI_CAP;
constexpr int kTriR = 4;                       // consecutive elements of one row N+(b) a lane probes per trip (one 16-byte load)
struct __attribute__((packed, aligned(4))) Int4U { int32_t x, y, z, w; };      // 16 bytes at a 4-byte aligned address
struct __attribute__((packed, aligned(4))) UInt2U { uint32_t x, y; };
constexpr int kTriBuf = 128;                   // parked triangles per wave on the unstaged path (handled once >= 64 are waiting)
#ifndef KOMB_TRI_REC
#define KOMB_TRI_REC 384
#endif
constexpr int kTriRec = KOMB_TRI_REC;           // triangle records a staged task keeps until it is done (own-role entries, 8 bytes each)
constexpr int kTriCand = KOMB_TRI_CAND;        // parked lookup candidates per wave (searched once >= 64 are waiting)
constexpr int kTriWaves = kBlock / kWave;
#ifndef KOMB_TRI_SIGW
#define KOMB_TRI_SIGW 8
#endif
constexpr int kTriSigW = KOMB_TRI_SIGW;          // 32-bit words of a source row's Bloom signature (a power of two)
constexpr int kTriSigShift = 32 - 5 - (kTriSigW == 2 ? 1 : kTriSigW == 4 ? 2 : kTriSigW == 8 ? 3 : 4);

enum : int { TRI_COUNT = 0, TRI_SINGLE = 2 };

#ifdef KOMB_TRI_PROFILE
// debug: when every wavefront of the enumeration started and ended (100 MHz clock), to see its tail
__device__ unsigned long long g_tri_prof[2 * 16384];
#endif

// BACK: other_or_cursor[x] starts at off[x+1]-1, the last position of x's slice, and is counted DOWN: the
// returning atomic is the third-role write position itself (no load of off[x+1] from a second random line)
// DENSE (TRI_SINGLE over capacity-bounded slices): the own-role entries do not go to the slices at all.  A staged
// task keeps one 8-byte record per triangle in LDS and, when it is done, writes the own-role entries of all its edges
// as ONE compact block of `dense` (claimed from `dense_cursor` in chunks, see below), edge after edge:
// ownoff[e] = where edge e's entries start.  176 M scattered 8-byte stores (one HBM line each) become a coalesced
// stream.  A task with more triangles than the record buffer holds, and a row too long to stage, fall back to the
// slices (ownoff[e] = kOwnSpill); k_compact_inc reads either.
constexpr unsigned long long kOwnSpill = ~0ull;
constexpr uint32_t kOwnChunk = 4096;                // entries a wavefront claims from dense_cursor at a time (one atomic per ~15 tasks)

// STREAM (TRI_SINGLE, with DENSE): no slices at all.  Every incidence entry that does not go into a dense own-role block
// -- the third-role entry of every triangle, and all three entries of a triangle whose task has no block (a spilled or
// unstaged sub-range, a region that has run out) -- is appended as a record (key = the edge the entry belongs to, value =
// the other two edges) to ONE stream, 64 records per store instruction, no atomic and no scattered store per triangle.
// A wavefront claims kRecChunk positions of the stream at a time; what it leaves unused gets the sentinel key (larger
// than every edge id).  The host then sorts the records by key (the destination-binned build of the index: DESIGN.md
// section 4.2) and merges them with the dense blocks.  A claim beyond `cap` writes nothing: the host sees the cursor
// pass the capacity and falls back to the exact two-pass build.
constexpr uint32_t kRecChunk = 1024;
struct TriStream {
    uint32_t *key;                       // [cap]
    int2 *val;                           // [cap]
    unsigned long long *cursor;          // positions claimed so far
    unsigned long long cap;
    uint32_t sentinel;
};

template <int MODE, class OffT = uint32_t, bool BACK = false, bool DENSE = false, bool STREAM = false>     // OffT: 64-bit when the bounded slices exceed 2^32 entries
__global__ __launch_bounds__(kBlock, KOMB_TRI_EU) void k_triangles(const uint32_t *__restrict__ orow, const int32_t *__restrict__ ocol,
                                                      int64_t nv, int64_t task_lo, int64_t task_hi,
                                                      uint32_t *own, uint32_t *other_or_cursor,
                                                      const OffT *__restrict__ off, int2 *__restrict__ inc,
                                                      int2 *__restrict__ dense, unsigned long long *dense_cursor, unsigned long long dense_cap,
                                                      unsigned long long *__restrict__ ownoff, int ablate, TriStream ts, int tv)
{
    // tv: consecutive source vertices per task (<= kTriV).  Fewer than kTriV when the graph has few vertices for its work
    // (a 20 000-vertex graph with 4 M edges is 1 250 tasks of 16 vertices: not even one per SIMD)
    static_assert(!STREAM || (MODE == TRI_SINGLE && DENSE), "the record stream replaces the slices of the single pass");
    // ablate (debug, KOMB_TRI_ABLATE): 1 = no gather of w, 2 = no row lookup, 4 = no stores/atomics, 8 = no probes at all,
    // 16 = no own-role stores, 32 = no third-role atomic + store
    static_assert(kTriCap <= 256, "edge indices and cursors of a staged task are kept in 8 bits");
    static_assert(kTriRec * sizeof(uint2) >= kTriBuf * sizeof(uint3), "the unstaged path parks its triangles in the record buffer");
    __shared__ int32_t sh_col[kTriWaves][kTriCap];
    __shared__ uint32_t sh_cnt[kTriWaves][kTriCap];
    __shared__ uint32_t sh_orow[kTriWaves][kTriV + 1];
    __shared__ uint32_t sh_pref[kTriWaves][kWave];
    __shared__ uint32_t sh_rb0[kTriWaves][kWave];
    __shared__ uint32_t sh_ra0[kTriWaves][kWave];
    __shared__ uint32_t sh_ra1[kTriWaves][kWave];
    __shared__ uint2 sh_rec[kTriWaves][kTriRec];
    __shared__ uint3 sh_cand[kTriWaves][kTriCand];
    __shared__ uint32_t sh_sig[kTriWaves][kTriSigW * kTriV];
    __shared__ uint32_t sh_ri[kTriWaves][kWave];
    __shared__ uint32_t sh_len[kTriWaves][kWave];
    __shared__ uint8_t sh_rid[kTriWaves][kTriCap];
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    int32_t *s_col = sh_col[w];
    uint32_t *s_cnt = sh_cnt[w], *s_orow = sh_orow[w], *s_pref = sh_pref[w];
    uint32_t *s_rb0 = sh_rb0[w], *s_ra0 = sh_ra0[w], *s_ra1 = sh_ra1[w];
    uint2 *s_rec = sh_rec[w];
    uint3 *s_tri = reinterpret_cast<uint3 *>(sh_rec[w]), *s_cand = sh_cand[w];
    uint32_t *s_sig = sh_sig[w], *s_ri = sh_ri[w], *s_len = sh_len[w];
    uint8_t *s_rid = sh_rid[w];
    const int64_t gw = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * kBlock) >> 6;
    unsigned long long chunk_pos = 0, chunk_end = 0;             // this wavefront's claim on `dense` (wave-uniform)
    unsigned long long rec_pos = 0, rec_end = 0;                 // this wavefront's claim on the record stream (wave-uniform)
    // all 64 lanes call: the lanes with `has` append (key, val) at consecutive positions of the wavefront's claim
    auto rec_append = [&](bool has, uint32_t key, int2 val) {
        const uint64_t m = __ballot(has);
        if (!m) return;
        const uint32_t c = (uint32_t)__popcll(m);
        const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt());
        const uint32_t left = (uint32_t)min((unsigned long long)c, rec_end - rec_pos);    // (wave-uniform) positions left in the current claim
        unsigned long long q = rec_pos + rank;
        if (left < c) {                                          // the claim runs out inside this append: the rest goes to a new one
            unsigned long long got = 0;
            if (lane == 0) got = atomicAdd(ts.cursor, (unsigned long long)kRecChunk);
            const uint32_t glo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)got);
            const uint32_t ghi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(got >> 32));
            const unsigned long long fresh = ((unsigned long long)ghi << 32) | glo;
            if (rank >= left) q = fresh + (rank - left);
            rec_pos = fresh + (c - left);
            rec_end = fresh + kRecChunk;
        } else rec_pos += c;
        if (has && q < ts.cap) { ts.key[q] = key; ts.val[q] = val; }
    };

#ifdef KOMB_TRI_PROFILE
    if (lane == 0 && gw < 16384) g_tri_prof[2 * gw] = wall_clock64();
#endif
    for (int64_t task = task_lo + gw; task < task_hi; task += nw) {
      const int64_t v0t = task * tv;
      const int nvt_all = (int)min((int64_t)tv, nv - v0t);
      const uint32_t myrow = (lane <= nvt_all) ? orow[v0t + lane] : 0u;       // lane l holds orow[v0t + l]
      // A task whose rows exceed the LDS budget is cut into sub-ranges of consecutive vertices that fit;
      // only a single row longer than the budget runs unstaged (global binary search, global atomics).
      for (int sub = 0; sub < nvt_all;) {
        const uint32_t sub_base = (uint32_t)__shfl((int)myrow, sub);
        const bool fits = lane > sub && lane <= nvt_all && myrow - sub_base <= (uint32_t)kTriCap;
        const int nfit = __popcll(__ballot(fits));                            // rows are cumulative: a prefix of lanes fits
        const int nvt = nfit > 0 ? nfit : 1;
        __builtin_amdgcn_wave_barrier();
        {
            const uint32_t val = (uint32_t)__shfl((int)myrow, (lane + sub) & (kWave - 1));
            if (lane <= nvt) s_orow[lane] = val;
        }
        __builtin_amdgcn_wave_barrier();
        sub += nvt;
        const uint32_t S0 = s_orow[0], S1 = s_orow[nvt];
        const uint32_t E = S1 - S0;
        if (E == 0) continue;
        const bool staged = E <= (uint32_t)kTriCap;
        // Signature width of this sub-range: its rows share the wave's kTriSigW * kTriV words -- 256 bits each when all kTriV
        // rows are staged together, up to 4096 bits when a single long row is (a 200-slot row fills 256 bits to 54%, which
        // rejects next to nothing: dense graphs ran the lookup for most of their probes)
        int sig_lw = 0;                                         // (wave-uniform) log2 of the words per row
        while ((kTriSigW << (sig_lw + 1)) * nvt <= kTriSigW * kTriV) ++sig_lw;
        const int sig_w = kTriSigW << sig_lw;
        const int sig_shift = kTriSigShift - sig_lw;
        if (staged) {
            for (uint32_t k = (uint32_t)lane; k < E; k += kWave) { s_col[k] = ocol[S0 + k]; s_cnt[k] = 0u; }
            for (int x = lane; x < kTriSigW * kTriV; x += kWave) s_sig[x] = 0u;
            __builtin_amdgcn_wave_barrier();
            // Bloom signature (32 * kTriSigW bits) of every owned row: a probe whose bit is clear cannot be in the row
            for (uint32_t k = (uint32_t)lane; k < E; k += kWave) {
                int lo = 0, hi = nvt - 1;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_orow[mid] <= S0 + k) lo = mid; else hi = mid - 1; }
                const uint32_t hb = ((uint32_t)s_col[k] * 0x9E3779B1u) >> sig_shift;
                atomicOr(&s_sig[sig_w * lo + (int)(hb >> 5)], 1u << (hb & 31u));
                s_rid[k] = (uint8_t)lo;                                   // the edge's source row, for the batches below
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- unstaged sub-range (one row longer than the LDS budget): triangles are parked and handled 64 at a time,
        // all three roles through global cursors
        uint32_t n_tri = 0;                                     // parked triangles (wave-uniform)
        auto flush_tris = [&]() {
            __builtin_amdgcn_wave_barrier();
            for (uint32_t b0 = 0; b0 < n_tri; b0 += kWave) {
                const uint32_t x = b0 + (uint32_t)lane;
                if (x < n_tri) {
                    const uint3 tr = s_tri[x];
                    const uint32_t e = S0 + tr.x, i = S0 + tr.y, jj = tr.z;
                    if (MODE == TRI_COUNT) {
                        atomicAdd(&other_or_cursor[e], 1u); atomicAdd(&other_or_cursor[i], 1u); atomicAdd(&other_or_cursor[jj], 1u);
                    } else if (!STREAM) {
                        const OffT pe = off[e] + atomicAdd(&own[e], 1u);
                        const OffT pi = off[i] + atomicAdd(&own[i], 1u);
                        const OffT pj = BACK ? (OffT)atomicSub(&other_or_cursor[jj], 1u) : off[jj + 1] - 1u - atomicAdd(&other_or_cursor[jj], 1u);
                        inc[pe] = make_int2((int)i, (int)jj);
                        inc[pi] = make_int2((int)e, (int)jj);
                        inc[pj] = make_int2((int)e, (int)i);
                    }
                }
                if (STREAM) {                                    // all three entries of these triangles are records
                    const bool has = x < n_tri;
                    const uint3 tr = has ? s_tri[x] : make_uint3(0u, 0u, 0u);
                    const uint32_t e = S0 + tr.x, i = S0 + tr.y, jj = tr.z;
                    rec_append(has, e, make_int2((int)i, (int)jj));
                    rec_append(has, i, make_int2((int)e, (int)jj));
                    rec_append(has, jj, make_int2((int)e, (int)i));
                }
            }
            __builtin_amdgcn_wave_barrier();
            n_tri = 0;
        };

        // ---- staged sub-range.  A triangle's own-role cursors are LDS atomics taken when it is found; the rest waits in
        // the record buffer: record = (e_rel | i_rel << 8 | cursor_e << 16 | cursor_i << 24, j).  The records are worked off
        // densely, 64 per pass (`drain`): the third-role entry (one returning global atomic + one store) and -- when the
        // own-role entries go to the slices (exact slices, or a DENSE sub-range that has spilled) -- the two own-role stores.
        // A DENSE sub-range keeps its records until it is done.
        uint32_t n_rec = 0, n_done = 0;                         // wave-uniform: records, records whose third role is written
        bool spilled = !DENSE;                                  // wave-uniform: own-role entries go to inc[off[edge] + cursor]
        auto drain = [&](uint32_t lo, uint32_t hi, bool third, bool own_role) {
            __builtin_amdgcn_wave_barrier();
            if (MODE == TRI_SINGLE && STREAM) for (uint32_t b0 = lo; b0 < hi; b0 += kWave) {
                const uint32_t x = b0 + (uint32_t)lane;
                const bool has = x < hi;
                const uint2 rc = has ? s_rec[x] : make_uint2(0u, 0u);
                const uint32_t e = S0 + (rc.x & 0xFFu), i = S0 + ((rc.x >> 8) & 0xFFu), jj = rc.y;
                if (third && !(ablate & 32)) rec_append(has, jj, make_int2((int)e, (int)i));
                if (own_role && !(ablate & 16)) {
                    rec_append(has, e, make_int2((int)i, (int)jj));
                    rec_append(has, i, make_int2((int)e, (int)jj));
                }
            }
            if (MODE == TRI_SINGLE && !STREAM) for (uint32_t b0 = lo; b0 < hi; b0 += kWave) {
                const uint32_t x = b0 + (uint32_t)lane;
                if (x < hi) {
                    const uint2 rc = s_rec[x];
                    const uint32_t e = S0 + (rc.x & 0xFFu), i = S0 + ((rc.x >> 8) & 0xFFu), jj = rc.y;
                    if (third && !(ablate & 32)) {
                        const OffT pj = BACK ? (OffT)atomicSub(&other_or_cursor[jj], 1u) : off[jj + 1] - 1u - atomicAdd(&other_or_cursor[jj], 1u);
                        inc[pj] = make_int2((int)e, (int)i);
                    }
                    if (own_role && !(ablate & 16)) {
                        inc[off[e] + ((rc.x >> 16) & 0xFFu)] = make_int2((int)i, (int)jj);
                        inc[off[i] + (rc.x >> 24)] = make_int2((int)e, (int)jj);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        };

        // probes that pass the signature test are parked and looked up densely, 64 at a time
        uint32_t n_cand = 0;                                    // wave-uniform
        auto search_cands = [&]() {
            __builtin_amdgcn_wave_barrier();
            for (uint32_t b0 = 0; b0 < n_cand; b0 += kWave) {
                const uint32_t x = b0 + (uint32_t)lane;
                uint32_t e_rel = 0, l = 0, jj = 0, rend = 0, n = 0;
                int32_t wvv = 0;
                if (x < n_cand) {
                    const uint3 c = s_cand[x];
                    e_rel = c.x & 0xFFFFu; jj = c.y;
                    const uint32_t r = c.x >> 16;
                    wvv = (int32_t)c.z;
                    l = s_orow[r] - S0;
                    rend = s_orow[r + 1] - S0;
                    n = rend - l;
                }
                // branchless lower_bound; the trip count is that of the longest row among these 64 candidates
                while (__ballot(n > 0)) {
                    const uint32_t half = n >> 1;
                    const uint32_t probe = min(l + half, (uint32_t)kTriCap - 1u);
                    const bool go = n > 0 && s_col[probe] < wvv;
                    l = go ? l + half + 1u : l;
                    n = go ? n - half - 1u : half;
                }
                const bool hit = x < n_cand && l < rend && s_col[min(l, (uint32_t)kTriCap - 1u)] == wvv && !(ablate & 4);
                const uint64_t hm = __ballot(hit);
                if (!hm) continue;
                if (hit) {
                    const uint32_t ce = atomicAdd(&s_cnt[e_rel], 1u), ci = atomicAdd(&s_cnt[l], 1u);
                    if (MODE == TRI_COUNT) atomicAdd(&other_or_cursor[jj], 1u);
                    else s_rec[n_rec + (uint32_t)__popcll(hm & lanemask_lt())] = make_uint2(e_rel | (l << 8) | (ce << 16) | (ci << 24), jj);
                }
                if (MODE == TRI_SINGLE) {
                    n_rec += (uint32_t)__popcll(hm);
                    if (n_rec - n_done >= (uint32_t)kWave) {                 // 64 or more are waiting: one dense pass over all of them
                        drain(n_done, n_rec, true, spilled);
                        n_done = n_rec;
                    }
                    if (n_rec > (uint32_t)kTriRec - kWave) {
                        // the buffer is full.  A DENSE sub-range gives up its block: everything kept so far, and what follows,
                        // goes to the slices
                        if (DENSE && !spilled) {
                            drain(n_done, n_rec, true, false);
                            drain(0, n_rec, false, true);
                            spilled = true;
                            if (lane == 0) atomicAdd(dense_cursor + 1, 1ull);     // statistics
                        } else drain(n_done, n_rec, true, true);
                        n_rec = 0; n_done = 0;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            n_cand = 0;
        };

        for (uint32_t p0 = 0; p0 < E; p0 += kWave) {
            // lane <-> owned edge e = S0 + p0 + lane
            const uint32_t rel = p0 + (uint32_t)lane;
            const bool valid = rel < E;
            uint32_t rb0 = 0, lenb = 0, ra0 = 0, ra1 = 0, ri = 0;
            if (valid) {
                int lo = 0, hi = nvt - 1;                     // source vertex: last idx with s_orow[idx] <= S0+rel
                if (staged) lo = (int)s_rid[rel];
                else while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (s_orow[mid] <= S0 + rel) lo = mid; else hi = mid - 1;
                }
                ra0 = s_orow[lo] - S0; ra1 = s_orow[lo + 1] - S0;
                ri = (uint32_t)lo;
                const int32_t b = staged ? s_col[rel] : ocol[S0 + rel];
                const UInt2U rb = *reinterpret_cast<const UInt2U *>(orow + b);     // orow[b], orow[b + 1] in one load
                rb0 = rb.x;
                lenb = rb.y - rb0;
            }
            // The probe items of the 64 edges are cut into chunks of kTriR consecutive elements of ONE row N+(b) and the
            // chunks are flattened over the lanes: one owner search and one 16-byte load per kTriR probes (a probe per lane
            // cost an owner search -- 8 LDS reads -- each; the kernel is bound by its LDS traffic)
            const uint32_t incl = wave_incl_scan((lenb + (uint32_t)kTriR - 1u) / (uint32_t)kTriR);
            const uint32_t total = (uint32_t)__shfl((int)incl, kWave - 1);
            __builtin_amdgcn_wave_barrier();
            s_pref[lane] = incl; s_rb0[lane] = rb0; s_ra0[lane] = ra0; s_ra1[lane] = ra1; s_ri[lane] = ri; s_len[lane] = lenb;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t it0 = 0; it0 < ((ablate & 8) ? 0u : total); it0 += kWave) {
                const uint32_t it = it0 + (uint32_t)lane;
                const bool live = it < total;
                int t = 0;                                    // owner: smallest t with s_pref[t] > it (branchless, 6 fixed steps)
#pragma unroll
                for (int st = kWave / 2; st > 0; st >>= 1) t += (s_pref[t + st - 1] <= it) ? st : 0;
                const uint32_t first = t ? s_pref[t - 1] : 0u;
                const uint32_t c0 = (it - first) * (uint32_t)kTriR;           // first element of the chunk, relative to the row
                const uint32_t j0 = s_rb0[t] + c0;                            // its slot in row b
                const uint32_t nin = live ? min((uint32_t)kTriR, s_len[t] - c0) : 0u;    // elements of the chunk inside the row
                if (staged) {
                    int32_t wv[kTriR];
                    if (live && !(ablate & 1)) {
                        // the rows are 4-byte aligned only; the array is padded so that the last chunk may read past its row
                        const Int4U q = *reinterpret_cast<const Int4U *>(ocol + j0);
                        wv[0] = q.x; wv[1] = q.y; wv[2] = q.z; wv[3] = q.w;
                    } else {
#pragma unroll
                        for (int k = 0; k < kTriR; ++k) wv[k] = (int32_t)(j0 + (uint32_t)k);
                    }
                    const uint32_t r = s_ri[t];
#pragma unroll
                    for (int k = 0; k < kTriR; ++k) {
                        bool cand = false;
                        if ((uint32_t)k < nin && !(ablate & 2)) {
                            const uint32_t hb = ((uint32_t)wv[k] * 0x9E3779B1u) >> sig_shift;
                            cand = (s_sig[(uint32_t)sig_w * r + (hb >> 5)] >> (hb & 31u)) & 1u;
                        }
                        const uint64_t cm = __ballot(cand);
                        if (cm) {
                            if (cand) s_cand[n_cand + (uint32_t)__popcll(cm & lanemask_lt())] =
                                make_uint3((p0 + (uint32_t)t) | (r << 16), j0 + (uint32_t)k, (uint32_t)wv[k]);
                            n_cand += (uint32_t)__popcll(cm);
                            if (n_cand >= (uint32_t)kWave) search_cands();     // at most 63 are waiting when the next 64 arrive
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < kTriR; ++k) {
                        const bool look = (uint32_t)k < nin && !(ablate & 2);
                        const int32_t wvk = look ? ocol[j0 + (uint32_t)k] : 0;
                        uint32_t l = look ? s_ra0[t] : 0u, h = look ? s_ra1[t] : 0u;
                        const uint32_t rend = h;
                        while (l < h) { const uint32_t mid = (l + h) >> 1; if (ocol[S0 + mid] < wvk) l = mid + 1; else h = mid; }
                        const bool hit = l < rend && ocol[S0 + l] == wvk && !(ablate & 4);
                        const uint64_t hm = __ballot(hit);
                        if (hm) {
                            if (hit) s_tri[n_tri + (uint32_t)__popcll(hm & lanemask_lt())] = make_uint3(p0 + (uint32_t)t, l, j0 + (uint32_t)k);
                            n_tri += (uint32_t)__popcll(hm);
                            if (n_tri >= (uint32_t)kTriBuf - kWave) flush_tris();
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (!staged) {
            flush_tris();
            if (DENSE) for (uint32_t k = (uint32_t)lane; k < E; k += kWave) ownoff[S0 + k] = kOwnSpill;
            continue;
        }
        search_cands();
        if (MODE == TRI_SINGLE) { drain(n_done, n_rec, true, spilled); n_done = n_rec; }
        bool to_dense = MODE == TRI_SINGLE && DENSE && !spilled;     // wave-uniform
        unsigned long long base = 0;
        if (to_dense) {
            // the own-role entries of this sub-range as one block of `dense`: exclusive prefix of the cursors (into s_col,
            // which is done with), a claim on the wavefront's chunk, the offsets, the entries
            __builtin_amdgcn_wave_barrier();
            uint32_t run = 0;
            for (uint32_t k0 = 0; k0 < E; k0 += kWave) {
                const uint32_t k = k0 + (uint32_t)lane;
                const uint32_t c = k < E ? s_cnt[k] : 0u;
                const uint32_t ic = wave_incl_scan(c);
                if (k < E) s_col[k] = (int32_t)(run + ic - c);
                run += (uint32_t)__shfl((int)ic, kWave - 1);
            }
            if (run) {
                if (chunk_pos + run > chunk_end) {               // (wave-uniform) the block does not fit what is left of the chunk
                    const uint32_t want = run > kOwnChunk ? run : kOwnChunk;
                    unsigned long long got = 0;
                    if (lane == 0) got = atomicAdd(dense_cursor, (unsigned long long)want);
                    const uint32_t glo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)got);
                    const uint32_t ghi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(got >> 32));
                    chunk_pos = ((unsigned long long)ghi << 32) | glo;
                    chunk_end = chunk_pos + want;
                    if (chunk_end > dense_cap) {
                        // the region is sized by a bound on the entries plus a share for the chunks' unused tails; blocks of
                        // an unlucky size in a graph that meets the bound can exceed it: this sub-range goes to the slices
                        to_dense = false;
                        chunk_pos = 0; chunk_end = 0;
                    }
                }
                if (to_dense) { base = chunk_pos; chunk_pos += run; }
            }
        }
        if (to_dense) {
            __builtin_amdgcn_wave_barrier();
            for (uint32_t k = (uint32_t)lane; k < E; k += kWave) { own[S0 + k] = s_cnt[k]; ownoff[S0 + k] = base + (uint32_t)s_col[k]; }
            if (!(ablate & 16)) for (uint32_t b0 = 0; b0 < n_rec; b0 += kWave) {
                const uint32_t x = b0 + (uint32_t)lane;
                if (x < n_rec) {
                    const uint2 rc = s_rec[x];
                    const uint32_t er = rc.x & 0xFFu, ir = (rc.x >> 8) & 0xFFu;
                    dense[base + (uint32_t)s_col[er] + ((rc.x >> 16) & 0xFFu)] = make_int2((int)(S0 + ir), (int)rc.y);
                    dense[base + (uint32_t)s_col[ir] + (rc.x >> 24)] = make_int2((int)(S0 + er), (int)rc.y);
                }
            }
        } else {
            if (MODE == TRI_SINGLE && DENSE && !spilled) drain(0, n_rec, false, true);      // no room in the region: own-role entries of every record to the slices
            __builtin_amdgcn_wave_barrier();
            for (uint32_t k = (uint32_t)lane; k < E; k += kWave) {
                own[S0 + k] = STREAM ? 0u : s_cnt[k];            // (STREAM: those entries are records, counted with the sorted stream)
                if (MODE == TRI_SINGLE && DENSE) ownoff[S0 + k] = kOwnSpill;
            }
        }
      }   // sub-ranges
    }
    if (STREAM) for (unsigned long long q = rec_pos + (unsigned long long)lane; q < rec_end && q < ts.cap; q += kWave) ts.key[q] = ts.sentinel;
#ifdef KOMB_TRI_PROFILE
    if (lane == 0 && gw < 16384) g_tri_prof[2 * gw + 1] = wall_clock64();
#endif
}

__global__ __launch_bounds__(kBlock) void k_total_u32(const uint32_t *__restrict__ v, int64_t n, unsigned long long *__restrict__ total)
{
    unsigned long long t = 0;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) t += v[e];
    block_add_u64(t, total);
}

__global__ __launch_bounds__(kBlock) void k_back_cursors(const uint32_t *__restrict__ off, int64_t m, uint32_t *__restrict__ cursor)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e <= m; e += (int64_t)gridDim.x * kBlock)
        cursor[e] = e < m ? off[e + 1] - 1u : 0u;
}

// sup = own + other (64-bit total on the side).  back_off: other[] holds back cursors that started at
// back_off[e+1]-1 (k_back_cursors), so the third-role count is how far they moved.
__global__ __launch_bounds__(kBlock) void k_sum_counts(const uint32_t *__restrict__ own, const uint32_t *__restrict__ other,
                                                       const uint32_t *__restrict__ back_off,
                                                       int64_t m1, uint32_t *__restrict__ sum, unsigned long long *__restrict__ total)
{
    unsigned long long t = 0;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < m1; e += (int64_t)gridDim.x * kBlock) {
        const uint32_t oth = back_off == nullptr ? other[e] : (e + 1 < m1 ? back_off[e + 1] - 1u - other[e] : 0u);
        const uint32_t c = own[e] + oth;
        sum[e] = c;
        t += c;
    }
    block_add_u64(t, total);                                    // 64-bit: the 32-bit slice offsets must not wrap
}
// capacity of edge (a->b)'s slice in the single-pass layout: |N(a) & N(b)| <= d(a) - 1, a being the
// lower-(degree,id) endpoint.  total accumulates the 64-bit sum (the 32-bit offsets must not wrap).
// total[1]: bound on the OWN-role entries alone (the triangles an edge a->x closes with the other out-neighbours of a:
// at most d+(a) - 1), the capacity of the dense own-role region.
__global__ __launch_bounds__(kBlock) void k_slice_caps(const int32_t *__restrict__ osrc, const int32_t *__restrict__ deg,
                                                       const uint32_t *__restrict__ orow, int64_t m,
                                                       uint32_t *__restrict__ cap, unsigned long long *__restrict__ total)
{
    unsigned long long t = 0, to = 0;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < m; e += (int64_t)gridDim.x * kBlock) {
        const int32_t a = osrc[e];
        const uint32_t c = (uint32_t)(deg[a] - 1);
        cap[e] = c;
        t += c;
        to += orow[a + 1] - orow[a] - 1u;
    }
    block_add_u64(t, total);
    block_add_u64(to, total + 1);
}

// Dense index from the bounded slices: 64 consecutive edges per wavefront, their entries flattened
// over the lanes; the dense slices of consecutive edges are contiguous, so the writes are one
// coalesced stream.  Entry k of edge x sits at offc[x]+k (k < own[x]) or offc[x+1]-1-(k-own[x]).
// With a dense own-role region (own_dense / ownoff, see k_triangles) the own-role entries of edge x are
// own_dense[ownoff[x] + k] unless ownoff[x] is kOwnSpill.
template <class OffT>
__global__ __launch_bounds__(kBlock) void k_compact_inc(const OffT *__restrict__ offc, const uint32_t *__restrict__ own,
                                                        const uint32_t *__restrict__ off, const int2 *__restrict__ sparse,
                                                        const int2 *__restrict__ own_dense, const unsigned long long *__restrict__ ownoff,
                                                        int2 *__restrict__ dense, int64_t m)
{
    __shared__ uint32_t sh_end[kBlock / kWave][kWave];
    const int lane = lane_id();
    uint32_t *s_end = sh_end[threadIdx.x >> 6];
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
    const int64_t nbatches = (m + kWave - 1) / kWave;
    for (int64_t bt = wave; bt < nbatches; bt += nwaves) {
        const int64_t e = bt * kWave + lane;
        uint32_t d0 = 0, len = 0, ow = 0;
        OffT c0 = 0, c1 = 0;
        unsigned long long oo = kOwnSpill;
        if (e < m) { d0 = off[e]; len = off[e + 1] - d0; c0 = offc[e]; c1 = offc[e + 1]; ow = own[e]; if (ownoff) oo = ownoff[e]; }
        const uint32_t incl = wave_incl_scan(len);
        const uint32_t total = (uint32_t)__shfl((int)incl, kWave - 1);
        const uint32_t dbase = (uint32_t)__shfl((int)d0, 0);
        __builtin_amdgcn_wave_barrier();
        s_end[lane] = incl;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t it0 = 0; it0 < total; it0 += kWave) {
            const uint32_t it = it0 + (uint32_t)lane;
            int t = 0;
            {
                int lo = 0;
#pragma unroll
                for (int st = kWave / 2; st > 0; st >>= 1) lo += (s_end[lo + st - 1] <= it) ? st : 0;
                t = lo;
            }
            const uint32_t first = t ? s_end[t - 1] : 0u;
            const OffT tc0 = (OffT)__shfl((unsigned long long)c0, t), tc1 = (OffT)__shfl((unsigned long long)c1, t);
            const uint32_t tow = (uint32_t)__shfl((int)ow, t);
            const unsigned long long too = __shfl(oo, t);
            if (it < total) {
                const uint32_t k = it - first;
                if (k < tow && too != kOwnSpill) dense[dbase + it] = own_dense[too + k];
                else {
                    const OffT sp = k < tow ? tc0 + k : tc1 - 1u - (k - tow);
                    dense[dbase + it] = sparse[sp];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the record stream of the single pass (k_triangles, STREAM): destination-binned build of the index
// The records are radix-sorted by the key bits above kBinBits only: records of one BIN -- 2^kBinBits consecutive edge
// ids -- become contiguous, in no particular order inside the bin.  One workgroup then finishes a bin out of LDS: a
// histogram of the bin's keys gives every edge its record count (k_bin_count); after the scan of the supports, per-edge
// write cursors in LDS place every record value in its edge's slice (k_bin_fill) -- LDS atomics and stores inside one
// ~100 KB window of the index, instead of one global atomic and one scattered HBM line per triangle.
constexpr int kBinBits = 11;
constexpr uint32_t kBinEdges = 1u << kBinBits;

// boff[b] = first sorted record whose bin is >= b, for b = 0 .. nb (the sentinel keys lie above every bin).  One thread
// per bin, binary search: 25 k threads x 27 probes, no pass over the keys.
__global__ __launch_bounds__(kBlock) void k_bin_offsets(const uint32_t *__restrict__ key, int64_t n, int64_t nb, uint32_t *__restrict__ boff)
{
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b <= nb; b += (int64_t)gridDim.x * kBlock) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if ((int64_t)(key[mid] >> kBinBits) < b) lo = mid + 1; else hi = mid;
        }
        boff[b] = (uint32_t)lo;
    }
}

// sum over the vertices of d+(a) (d+(a) - 1): bound on the own-role entries (every edge a->x closes at most d+(a) - 1
// triangles with the other out-neighbours of a); half of it bounds the triangles
__global__ __launch_bounds__(kBlock) void k_own_bound(const uint32_t *__restrict__ orow, int64_t nv, unsigned long long *__restrict__ total)
{
    unsigned long long t = 0;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += (int64_t)gridDim.x * kBlock) {
        const unsigned long long d = orow[v + 1] - orow[v];
        t += d ? d * (d - 1ull) : 0ull;
    }
    block_add_u64(t, total);
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
__global__ __launch_bounds__(kBlock) void k_bin_count(const uint32_t *__restrict__ key, const uint32_t *__restrict__ boff, int64_t nb,
                                                      const uint32_t *__restrict__ own, int64_t m, uint32_t *__restrict__ sum,
                                                      uint32_t *__restrict__ bin_total, unsigned long long *__restrict__ total,
                                                      int32_t *__restrict__ min_pos)
{
    // min_pos: the smallest positive support = the peel's first level (k_bin_finish queues its frontier)
    __shared__ uint32_t sh_cnt[kBinEdges];
    __shared__ uint32_t sh_part[kBlock / kWave];
    unsigned long long t = 0;
    int32_t lmin = 0x7FFFFFFF;
    for (int64_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const int64_t x0 = b << kBinBits;
        const uint32_t nx = (uint32_t)min((int64_t)kBinEdges, m - x0);
        for (uint32_t i = threadIdx.x; i < kBinEdges; i += kBlock) sh_cnt[i] = 0u;
        __syncthreads();
        const uint32_t r0 = boff[b], r1 = boff[b + 1];
        for (uint32_t r = r0 + threadIdx.x; r < r1; r += kBlock) atomicAdd(&sh_cnt[key[r] - (uint32_t)x0], 1u);
        __syncthreads();
        uint32_t tb = 0;
        for (uint32_t i = threadIdx.x; i < nx; i += kBlock) {
            const uint32_t c = own[x0 + i] + sh_cnt[i];
            sum[x0 + i] = c;
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
    if (blockIdx.x == 0 && threadIdx.x == 0) { sum[m] = 0u; bin_total[nb] = 0u; }
    if (threadIdx.x == 0 && t) atomicAdd(total, t);
    lmin = wave_min(lmin);
    if (lane_id() == 0 && lmin != 0x7FFFFFFF) atomicMin(min_pos, lmin);
}

// Dense index of a bin's edges.  An edge's slice is [its records' values | its own-role entries]; the slices of a
// bin's 2^kBinBits consecutive edges are one contiguous WINDOW of the index (~40 KB).  One workgroup per bin assembles
// the window in LDS -- every record takes its position from a per-edge cursor (LDS atomic), the own-role entries are
// copied out of the tasks' dense blocks -- and then writes it as one coalesced stream: no global atomic, no scattered
// store, every line of the index written whole, once.  A window that does not fit the LDS buffer (hub edges) is written
// in place instead.  All loads of a phase are issued before the first is used: two workgroups per CU, and the kernel
// lives on memory-level parallelism.
constexpr int kFinBlock = 512;
constexpr uint32_t kWinCap = 7168;                 // entries of the LDS window (56 KB; with the two 8 KB tables: 2 workgroups per CU)
constexpr int kFinE = (int)(kBinEdges / kFinBlock);          // consecutive edges per thread
constexpr int kFinU = 4;                           // records per thread per trip
static_assert(kFinE == 4, "a thread loads its edges' supports and own-role counts as one 16-byte vector each");
// The kernel also does what followed the index build: the slice offsets off[] (a workgroup scan of the bin's supports on
// top of the bin's base -- the 100 M-element device scan is gone) and the peel's initial state (support, alive marker or
// "gone" for a triangle-free edge, the count of those and the smallest positive support for the first level).
__global__ __launch_bounds__(kFinBlock) void k_bin_finish(const uint32_t *__restrict__ key, const int2 *__restrict__ val,
                                                         const uint32_t *__restrict__ boff, int64_t nb,
                                                         const uint32_t *__restrict__ own, const uint32_t *__restrict__ cnt,
                                                         const uint32_t *__restrict__ bin_base,
                                                         const int2 *__restrict__ own_dense, const unsigned long long *__restrict__ ownoff,
                                                         int2 *__restrict__ dense, int64_t m,
                                                         uint32_t *__restrict__ off, int32_t *__restrict__ sup, int32_t *__restrict__ stamp,
                                                         int32_t *__restrict__ truss, uint32_t *__restrict__ init, int32_t *__restrict__ light0)
{
    // init[0] += triangle-free edges; init[1] = the smallest positive support (from k_bin_count) = the peel's first level L1.
    // The edges with support L1 ARE that level's first frontier (nothing has been decremented yet): they are stamped with
    // round 1 / trussness L1 + 2 and appended to light queue 0 here (one reservation per bin on init[2]), so the peel starts
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
    for (int64_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const int64_t x0 = b << kBinBits;
        const uint32_t nx = (uint32_t)min((int64_t)kBinEdges, m - x0);
        const uint64_t r0 = boff[b], r1 = boff[b + 1];
        const uint32_t base = bin_base[b];
        // ---- the thread's 4 consecutive edges: supports and own-role counts, then (for the edges that have some) where their blocks are
        const uint32_t i0 = threadIdx.x * (uint32_t)kFinE;
        uint32_t c[kFinE], ow[kFinE];
        unsigned long long oo[kFinE];
        if (i0 + kFinE <= nx) {
            const uint4 v = *reinterpret_cast<const uint4 *>(cnt + x0 + i0);
            const uint4 q = *reinterpret_cast<const uint4 *>(own + x0 + i0);
            c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
            ow[0] = q.x; ow[1] = q.y; ow[2] = q.z; ow[3] = q.w;
        } else {
#pragma unroll
            for (int u = 0; u < kFinE; ++u) { const bool in = i0 + (uint32_t)u < nx; c[u] = in ? cnt[x0 + i0 + u] : 0u; ow[u] = in ? own[x0 + i0 + u] : 0u; }
        }
#pragma unroll
        for (int u = 0; u < kFinE; ++u) oo[u] = ow[u] ? ownoff[x0 + i0 + u] : 0ull;
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
        for (int u = 0; u < kFinE; ++u)
            if (i0 + (uint32_t)u < nx) { sh_off[i0 + u] = o[u]; sh_cur[i0 + u] = o[u]; }
        if (threadIdx.x == kFinBlock - 1) sh_off[kBinEdges] = o[kFinE];      // the bin's total (edges beyond nx count 0)
        // ... written out, with the peel's initial state
        if (i0 + kFinE <= nx) {
            *reinterpret_cast<uint4 *>(off + x0 + i0) = make_uint4(base + o[0], base + o[1], base + o[2], base + o[3]);
            int4 sv, mv, tv;
            int32_t *svp = &sv.x, *mvp = &mv.x, *tvp = &tv.x;
#pragma unroll
            for (int u = 0; u < kFinE; ++u) {
                const bool first = queue_first && (int32_t)c[u] == L1;
                svp[u] = (int32_t)c[u];
                mvp[u] = first ? 1 : (c[u] ? alive_marker(c[u]) : 0);   // round 1: the first frontier; round 0: gone before the first sub-round
                tvp[u] = first ? L1 + 2 : 2;
                if (!c[u]) ++zeros;
            }
            *reinterpret_cast<int4 *>(sup + x0 + i0) = sv;
            *reinterpret_cast<int4 *>(stamp + x0 + i0) = mv;
            *reinterpret_cast<int4 *>(truss + x0 + i0) = tv;
        } else {
#pragma unroll
            for (int u = 0; u < kFinE; ++u) if (i0 + (uint32_t)u < nx) {
                const int64_t e = x0 + i0 + u;
                const bool first = queue_first && (int32_t)c[u] == L1;
                off[e] = base + o[u];
                sup[e] = (int32_t)c[u];
                stamp[e] = first ? 1 : (c[u] ? alive_marker(c[u]) : 0);
                truss[e] = first ? L1 + 2 : 2;
                if (!c[u]) ++zeros;
            }
        }
        __syncthreads();
        if (hall) {                                                 // the bin's part of the first frontier, in edge order
            uint32_t q = sh_qbase + hbefore + hincl - hits;
#pragma unroll
            for (int u = 0; u < kFinE; ++u) if (queue_first && (int32_t)c[u] == L1 && i0 + (uint32_t)u < nx) light0[q++] = (int32_t)(x0 + i0 + u);
        }
        const uint32_t W = sh_off[kBinEdges];
        if (b == nb - 1 && threadIdx.x == 0) off[m] = base + W;
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
                if (rr < r1) {
                    const uint32_t p = atomicAdd(&sh_cur[k[u] - (uint32_t)x0], 1u);
                    if (inwin) sh_win[p] = v[u]; else dense[base + p] = v[u];
                }
            }
        }
        // ---- own-role entries of the thread's edges: they end the edges' slices; the 4 edges' copies advance together
        uint32_t most = 0;
#pragma unroll
        for (int u = 0; u < kFinE; ++u) most = max(most, ow[u]);
        for (uint32_t kk = 0; kk < most; ++kk) {
            int2 t[kFinE];
#pragma unroll
            for (int u = 0; u < kFinE; ++u) if (kk < ow[u]) t[u] = own_dense[oo[u] + kk];
#pragma unroll
            for (int u = 0; u < kFinE; ++u) if (kk < ow[u]) {
                const uint32_t p = o[u + 1] - ow[u] + kk;
                if (inwin) sh_win[p] = t[u]; else dense[base + p] = t[u];
            }
        }
        __syncthreads();
        // ---- the window, as a stream
        if (inwin) for (uint32_t j = threadIdx.x; j < W; j += kFinBlock) dense[base + j] = sh_win[j];
        __syncthreads();
    }
    block_add_min(zeros, 0x7FFFFFFF, &init[0], (int32_t *)&init[1]);
}

// peel state from the slice lengths.  Triangle-free edges are peeled here (trussness 2); init[0]
// counts them and init[1] receives the smallest positive support = the first populated level.
__global__ __launch_bounds__(kBlock) void k_peel_init(int64_t m, const uint32_t *__restrict__ off,
                                                      int32_t *__restrict__ sup, int32_t *__restrict__ stamp,
                                                      int32_t *__restrict__ truss, uint32_t *__restrict__ init)
{
    uint32_t zeros = 0;
    int32_t lmin = 0x7FFFFFFF;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < m; e += (int64_t)gridDim.x * kBlock) {
        const int32_t s0 = (int32_t)(off[e + 1] - off[e]);
        sup[e] = s0;
        if (s0 == 0) { stamp[e] = 0; truss[e] = 2; ++zeros; }       // round 0: gone before the first sub-round
        else { stamp[e] = alive_marker((uint32_t)s0); lmin = min(lmin, s0); }
    }
    block_add_min(zeros, lmin, &init[0], (int32_t *)&init[1]);
}

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
    const uint32_t *off;
    const int2 *inc;
    int32_t *sup;
    int32_t *stamp;
    int32_t *truss;

    __device__ __forceinline__ const int32_t *scan_marker() const { return stamp; }
    __device__ __forceinline__ const int32_t *scan_key() const { return sup; }
    __device__ __forceinline__ void mark_scanned(uint32_t e, const CtrlView &cv) const
    {
        stamp[e] = cv.round;
        truss[e] = cv.level + 2;
    }
    __device__ __forceinline__ void slice(uint32_t e, uint32_t &b, uint32_t &len) const
    {
        b = off[e];
        len = off[e + 1] - b;
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
    __device__ __forceinline__ void item_apply(const Loaded &ld, const CtrlView &cv, int32_t &t0, int32_t &t1, uint32_t &c0, uint32_t &c1) const
    {
        const int32_t r = cv.round, L = cv.level;
        if (ld.sx < r || ld.sy < r) return;             // an edge of the triangle is already gone
        const bool xin = (ld.sx == r), yin = (ld.sy == r);
        const bool decx = !xin && (!yin || ld.me < ld.y);
        const bool decy = !yin && (!xin || ld.me < ld.x);
        if (decx && atomicSub(&sup[ld.x], 1) == L + 1) { stamp[ld.x] = r + 1; truss[ld.x] = L + 2; t0 = ld.x; c0 = marker_chunks(ld.sx); }
        if (decy && atomicSub(&sup[ld.y], 1) == L + 1) { stamp[ld.y] = r + 1; truss[ld.y] = L + 2; t1 = ld.y; c1 = marker_chunks(ld.sy); }
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
    const uint32_t *off;
    const int2 *inc;
    int32_t *sup;
    int32_t *stamp;
    int32_t *truss;
    uint32_t lo, hi;                     // internal edge ids this rank owns

    __device__ __forceinline__ const int32_t *scan_marker() const { return stamp; }
    __device__ __forceinline__ const int32_t *scan_key() const { return sup; }
    __device__ __forceinline__ void mark_scanned(uint32_t e, const CtrlView &cv) const
    {
        stamp[e] = cv.round;
        truss[e] = cv.level + 2;
    }
    __device__ __forceinline__ void slice(uint32_t e, uint32_t &b, uint32_t &len) const
    {
        b = off[e];
        len = off[e + 1] - b;
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
        if (decx && atomicSub(&sup[ld.x], 1) == L + 1) { stamp[ld.x] = r + 1; truss[ld.x] = L + 2; t0 = ld.x; }
        if (decy && atomicSub(&sup[ld.y], 1) == L + 1) { stamp[ld.y] = r + 1; truss[ld.y] = L + 2; t1 = ld.y; }
    }
};

// ---- hand-over to the local finish (local_dev.h)
// Collect pass: the peel engine run once over all live edges; a triangle whose other two edges are both live
// becomes an entry of the compact slice (ids of the remainder).
struct TrussCollect {
    static constexpr bool kChain = false;
    static constexpr bool kSingleStep = false;
    uint32_t units;
    const uint32_t *off;                 // the peel's incidence index
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
        b = off[e];
        len = off[e + 1] - b;
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

// -------------------------------------------------------------- result gather
// One thread per slot of the working CSR.  Upper slots (u < v) are the canonical copies of the edges; they are the
// suffix of their (ascending) row, so the canonical id c of slot j is ebase[u] + (j - first upper slot of u) -- which is
// also the number of upper slots before j.  Where trussness and support live is the ORIENTED slot of {u,v}:
//   - u precedes v: the oriented copy is this very slot; its oriented id is its rank among the kept slots (per-word rank
//     array of the orientation's bitmask + a popcount): no search, no gather.
//   - v precedes u: the oriented copy is the slot (v,u) of row v.  Those edges are the "reversed" oriented slots (source
//     id above target id).  Listed in oriented order they are sorted by (v, u); a STABLE sort by their target u puts them
//     in (u, v) order -- exactly the order in which the not-kept upper slots follow each other in the CSR.  So the k-th
//     not-kept upper slot is the k-th entry of that sorted list, and k = c - (kept upper slots before j): a stream on
//     both sides.  (Until round 3 this case binary-searched u in v's oriented row: 3.5 random lines per edge, 4.7 ms of
//     a 31 ms step; the sort of the 50 M (target, oriented id) pairs and this pass take 2.x ms.)
__global__ __launch_bounds__(kBlock) void k_popc_words(const unsigned long long *__restrict__ bits, int64_t nwords, uint32_t *__restrict__ cnt)
{
    for (int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x; w <= nwords; w += (int64_t)gridDim.x * kBlock)
        cnt[w] = w < nwords ? (uint32_t)__popcll(bits[w]) : 0u;
}

// the reversed oriented slots = the kept LOWER slots of the CSR, in slot order: (key = target id, value = the edge's
// (trussness, support)) at its rank among them = oriented id - kept upper slots before it.  Everything is a stream:
// the oriented id of a kept slot grows with the slot index.
__global__ __launch_bounds__(kBlock) void k_rev_emit(const int32_t *__restrict__ col, int64_t ns,
                                                     const unsigned long long *__restrict__ obits, const uint32_t *__restrict__ wrank,
                                                     const unsigned long long *__restrict__ kubits, const uint32_t *__restrict__ kurank,
                                                     const int32_t *__restrict__ truss, const uint32_t *__restrict__ off,
                                                     uint32_t *__restrict__ rkey, unsigned long long *__restrict__ rval)
{
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < ns; j += (int64_t)gridDim.x * kBlock) {
        const unsigned long long word = obits[j >> 6], ku = kubits[j >> 6];
        if (!(((word & ~ku) >> (j & 63)) & 1ull)) continue;            // not a kept lower slot
        const unsigned long long below = (1ull << (j & 63)) - 1ull;
        const uint32_t o = wrank[j >> 6] + (uint32_t)__popcll(word & below);
        const uint32_t rr = o - (kurank[j >> 6] + (uint32_t)__popcll(ku & below));
        rkey[rr] = (uint32_t)col[j];
        rval[rr] = (unsigned long long)(uint32_t)truss[o] | ((unsigned long long)(off[o + 1] - off[o]) << 32);
    }
}

__global__ __launch_bounds__(kBlock) void k_gather_canonical(const int32_t *__restrict__ src,
                                                             const int32_t *__restrict__ col, int64_t ns,
                                                             const unsigned long long *__restrict__ obits,
                                                             const uint32_t *__restrict__ wrank,
                                                             const unsigned long long *__restrict__ kubits,
                                                             const uint32_t *__restrict__ kurank,
                                                             const unsigned long long *__restrict__ rev_sorted,
                                                             const uint32_t *__restrict__ urank,
                                                             const int32_t *__restrict__ truss, const uint32_t *__restrict__ off,
                                                             int32_t *__restrict__ eu, int32_t *__restrict__ ev,
                                                             int32_t *__restrict__ tr_out, int32_t *__restrict__ sup_out)
{
    // a wavefront's 64 lanes hold the 64 slots of one word of the bitmasks: the canonical id of an upper slot is the number of
    // upper slots before it = urank[word] (a prefix sum the orientation's predicate pass prepared) + a popcount of the ballot
    const int64_t nwords = (ns + 63) >> 6;
    const int lane = lane_id();
    for (int64_t w = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6; w < nwords; w += ((int64_t)gridDim.x * kBlock) >> 6) {
        const int64_t j = (w << 6) + lane;
        int32_t u = 0, v = 0;
        if (j < ns) { u = src[j]; v = col[j]; }
        const bool up = j < ns && v > u;
        const unsigned long long um = __ballot(up);
        if (!up) continue;
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint32_t c = urank[w] + (uint32_t)__popcll(um & below);
        const unsigned long long word = obits[w];
        int32_t t, sp;
        if ((word >> lane) & 1ull) {                                   // slot (u,v) is the oriented copy
            const uint32_t o = wrank[w] + (uint32_t)__popcll(word & below);
            t = truss[o]; sp = (int32_t)(off[o + 1] - off[o]);
        } else {
            const unsigned long long r = rev_sorted[c - (kurank[w] + (uint32_t)__popcll(kubits[w] & below))];
            t = (int32_t)(uint32_t)r; sp = (int32_t)(uint32_t)(r >> 32);
        }
        eu[c] = u; ev[c] = v;
        tr_out[c] = t;
        sup_out[c] = sp;
    }
}

// sum_v d(v)^2 and sum_e min(d(u),d(v)) for the roofline's algorithmic bytes
__global__ __launch_bounds__(kBlock) void k_graph_moments(const int32_t *__restrict__ deg, int64_t nv,
                                                          const int32_t *__restrict__ osrc, const int32_t *__restrict__ ocol,
                                                          int64_t m, const uint32_t *__restrict__ cnt,
                                                          const uint32_t *__restrict__ orow,
                                                          unsigned long long *out /*[5]: sum d^2, sum min, max d, sum cnt, sum d+ + d+*/)
{
    unsigned long long s2 = 0, smin = 0, mx = 0, sc = 0, so = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kBlock) {
        const unsigned long long d = (unsigned long long)deg[i];
        s2 += d * d;
        mx = d > mx ? d : mx;
    }
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < m; e += (int64_t)gridDim.x * kBlock)
    {
        const int32_t a = osrc[e], b = ocol[e];
        smin += (unsigned long long)min(deg[a], deg[b]);
        sc += (unsigned long long)cnt[e];
        so += (unsigned long long)(orow[a + 1] - orow[a]) + (unsigned long long)(orow[b + 1] - orow[b]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s2 += __shfl_xor(s2, o); smin += __shfl_xor(smin, o); sc += __shfl_xor(sc, o); so += __shfl_xor(so, o);
        const unsigned long long t = __shfl_xor(mx, o); mx = t > mx ? t : mx;
    }
    if (lane_id() == 0) { atomicAdd(&out[0], s2); atomicAdd(&out[1], smin); atomicMax(&out[2], mx); atomicAdd(&out[3], sc); atomicAdd(&out[4], so); }
}

} // namespace

void truss_free(komb_ctx *ctx)
{
    ctx->pool.put(ctx->d_t_eu);
    ctx->pool.put(ctx->d_t_ev);
    ctx->pool.put(ctx->d_t_truss);
    ctx->pool.put(ctx->d_t_sup);
    ctx->d_t_eu = ctx->d_t_ev = ctx->d_t_truss = ctx->d_t_sup = nullptr;
    ctx->t_ne = -1; ctx->truss_done = false;
}

// ordered compaction of the CSR slots (src,col)[ns] that satisfy pred -> (out_src,out_col)[n_out] + out_rowptr
template <class Pred>
static int compact_slots(komb_ctx *ctx, DevBufs &bufs, const int32_t *src, const int32_t *col, int64_t ns, int64_t nv, Pred pred,
                         uint32_t *out_rowptr, int32_t **out_col, int32_t **out_src, int64_t *n_out,
                         unsigned long long **keep_bits_out = nullptr, uint32_t **word_rank_out = nullptr,
                         unsigned long long **keep_upper_out = nullptr, uint32_t **upper_cnt_out = nullptr)
{
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
    KOMB_HIP(ctx, bufs.alloc(out_col, (size_t)kept + 4));        // + 4: the triangle enumeration reads 16 bytes at a time
    KOMB_HIP(ctx, bufs.alloc(out_src, (size_t)kept));
    k_slot_filter<Pred, true><<<g, kBlock, 0, s>>>(src, col, ns, pred, nullptr, d_cb, *out_col, *out_src, d_bits, nullptr, nullptr);
    if ((int64_t)kept * 4 < nv) k_rowptr_search<<<grid_for(nv + 1), kBlock, 0, s>>>(*out_src, (int64_t)kept, nv, out_rowptr);
    else k_rowptr_from_src<<<grid_for(kept), kBlock, 0, s>>>(*out_src, (int64_t)kept, nv, out_rowptr);
    if (word_rank_out) {
        const int64_t nwords = (ns + 63) / 64;
        KOMB_HIP(ctx, bufs.alloc(word_rank_out, (size_t)nwords + 1));
        k_word_rank<<<grid_for((nwords + kChunkSlots / 64 - 1) / (kChunkSlots / 64)), kBlock, 0, s>>>(d_bits, d_cb, nwords, *word_rank_out);
    }
    bufs.release(d_cc); bufs.release(d_cb);
    if (keep_bits_out) *keep_bits_out = d_bits; else bufs.release(d_bits);
    *n_out = (int64_t)kept;
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
    const int64_t nv = ctx->nv;
    hipStream_t s = ctx->stream;
    komb_stats &st = ctx->stats;
    st.triangles = 0; st.truss_levels = st.truss_subrounds = st.truss_launches = 0;
    st.max_trussness = 0; st.ms_support = st.ms_peel = st.ms_orient = st.ms_tri_count = st.ms_tri_fill = st.ms_gather = st.ms_allreduce = st.ms_compact = 0.0;
    st.truss_scans = 0;
    if (nv == 0 || ctx->ne == 0) {
        KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_eu, 4)); KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_ev, 4));
        KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_truss, 4)); KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_sup, 4));
        ctx->t_ne = 0; ctx->truss_done = true;
        return KOMB_OK;
    }
    Range r_all("komb_truss_run");
    DevBufs bufs(ctx);
    const int gv = grid_for(nv);

    // ---- a5: working CSR = whole graph, or the subgraph induced by vmask (original ids kept)
    const uint32_t *w_rowptr = ctx->d_rowptr;
    const int32_t *w_col = ctx->d_col, *w_src = ctx->d_src;
    int64_t w_ns = 2 * ctx->ne;
    if (vmask_host) {
        uint8_t *d_mask = nullptr; uint32_t *d_rp = nullptr; int32_t *d_c = nullptr, *d_s = nullptr;
        KOMB_HIP(ctx, bufs.alloc(&d_mask, (size_t)nv));
        KOMB_HIP(ctx, bufs.alloc(&d_rp, (size_t)nv + 1));
        KOMB_HIP(ctx, hipMemcpyAsync(d_mask, vmask_host, (size_t)nv, hipMemcpyHostToDevice, s));
        int64_t ns_sub = 0;
        KOMB_TRY(compact_slots(ctx, bufs, ctx->d_src, ctx->d_col, w_ns, nv, PredMask{d_mask}, d_rp, &d_c, &d_s, &ns_sub));
        bufs.release(d_mask);
        w_rowptr = d_rp; w_col = d_c; w_src = d_s; w_ns = ns_sub;
        if (ns_sub == 0) {
            KOMB_HIP(ctx, hipStreamSynchronize(s));
            KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_eu, 4)); KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_ev, 4));
            KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_truss, 4)); KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_sup, 4));
            ctx->t_ne = 0; ctx->truss_done = true;
            return KOMB_OK;
        }
    }

    // ---- orientation: oriented CSR, internal edge id = oriented slot
    Range phase("truss: orientation");
    int32_t *d_deg = nullptr; uint32_t *d_orow = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_deg, (size_t)nv));
    KOMB_HIP(ctx, bufs.alloc(&d_orow, (size_t)nv + 1));
    ctx->timer.start(s);
    uint8_t *d_deg8 = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_deg8, (size_t)nv));
    // graphs whose 1-byte degree table outgrows the L2s (4 MiB per XCD) decide most slots on a 2-bit class table first
    const bool use_classes = (nv > (4ll << 20) || getenv("KOMB_DEG_CLASSES")) && !getenv("KOMB_NO_DEG_CLASSES");   // (no result depends on it)
    unsigned long long *d_dhist = nullptr;
    uint32_t *d_deg2 = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_dhist, 256));
    KOMB_HIP(ctx, hipMemsetAsync(d_dhist, 0, 256 * sizeof(unsigned long long), s));
    k_degree<<<gv, kBlock, 0, s>>>(w_rowptr, nv, d_deg, d_deg8, d_dhist);
    int32_t dth[3] = {255, 255, 255};
    if (use_classes) {
        // class thresholds: the quartiles of the slots' endpoint degrees (capped at 255)
        KOMB_HIP(ctx, bufs.alloc(&d_deg2, (size_t)(nv + 15) / 16 + 1));
        unsigned long long hh[256];
        KOMB_HIP(ctx, d2h(ctx, hh, d_dhist, sizeof(hh)));
        unsigned long long all = 0, run = 0;
        for (int d = 0; d < 256; ++d) all += hh[d];
        int q = 0;
        for (int d = 0; d < 256 && q < 3; ++d) {
            run += hh[d];
            while (q < 3 && run * 4 >= all * (unsigned long long)(q + 1)) dth[q++] = d;
        }
        k_degree_classes<<<grid_for((nv + 15) / 16), kBlock, 0, s>>>(d_deg8, nv, dth[0], dth[1], dth[2], d_deg2);
    }
    bufs.release(d_dhist);
    int32_t *d_ocol = nullptr, *d_osrc = nullptr;
    int64_t m = 0;
    unsigned long long *d_obits = nullptr;                           // bit j: working slot j is the oriented copy of its edge
    uint32_t *d_wrank = nullptr;                                     // oriented slots before each 64-slot word of d_obits
    unsigned long long *d_kubits = nullptr;                          // ... and is an upper slot (its row's id below its column's)
    uint32_t *d_urank = nullptr;                                     // upper slots per 64-slot word (the gather turns it into their prefix sum)
    if (use_classes) KOMB_TRY(compact_slots(ctx, bufs, w_src, w_col, w_ns, nv, PredOrientClass{d_deg, d_deg8, d_deg2, dth[0], dth[1], dth[2]}, d_orow, &d_ocol, &d_osrc, &m, &d_obits, &d_wrank, &d_kubits, &d_urank));
    else KOMB_TRY(compact_slots(ctx, bufs, w_src, w_col, w_ns, nv, PredOrient{d_deg, d_deg8}, d_orow, &d_ocol, &d_osrc, &m, &d_obits, &d_wrank, &d_kubits, &d_urank));
    st.ms_orient = ctx->timer.stop(s);
    phase.next("truss: triangles + incidence index");

    // ---- triangle support + incidence index
    const int ge = grid_for(m);
    // vertices per enumeration task: kTriV, fewer when that leaves the chip without enough tasks (>= 4 per resident wavefront)
    const int tri_tv = (int)std::max<int64_t>(1, std::min<int64_t>(kTriV, nv / (256 * KOMB_TRI_EU * kTriWaves * 4)));
    const int64_t ntasks = (nv + tri_tv - 1) / tri_tv;
    const int gt = grid_for(ntasks, kTriWaves);
    uint32_t *d_own = nullptr, *d_other = nullptr, *d_cnt = nullptr, *d_off = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_own, 2 * ((size_t)m + 1)));         // [own | other] contiguous: one all-reduce
    d_other = d_own + ((size_t)m + 1);
    KOMB_HIP(ctx, bufs.alloc(&d_cnt, (size_t)m + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_off, (size_t)m + 1));
    unsigned long long *d_mom = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_mom, 12));                           // [0..4] graph moments, [5] sum of supports, [6,7] capacity bounds, [8] sharded-check mismatches
    KOMB_HIP(ctx, hipMemsetAsync(d_mom, 0, 12 * sizeof(unsigned long long), s));
    KOMB_HIP(ctx, hipMemsetAsync(d_own, 0, 2 * ((size_t)m + 1) * sizeof(uint32_t), s));
    st.ms_allreduce = 0.0;

    // Three layouts of the index build:
    //   stream   (default) ONE enumeration; own-role entries leave a task as dense blocks, everything else as records of one
    //            stream that is then sorted by destination edge and merged with the blocks -- no atomic and no scattered
    //            store per triangle, no slices sized by a bound.  KOMB_INDEX=stream.
    //   slices   ONE enumeration into slices sized by the bound sup(a->b) <= d(a)-1 (~26 GB at |E|=100M), third-role
    //            entries through one returning atomic + one scattered store each, then a compaction.  KOMB_INDEX=slices.
    //   two_pass count, scan, second enumeration into exact slices: the fallback when the others do not fit in memory.
    //            KOMB_INDEX=two_pass (or KOMB_TWO_PASS=1).
    // A sharded run (world > 1) first counts the supports of its own source-vertex range and sums them over the ranks
    // (fn: the RCCL all-reduce), then builds the index whole with the layout above; the summed supports must equal the
    // supports the build finds.
    // how the peel ends (common.h): local fixed point (default), LDS tail (truss_tail.h), or the general engine alone
    const FinishMode fin = finish_mode(FIN_LOCAL);
    uint32_t tail_limit = 0;
    if (fin == FIN_LDS) {
        tail_limit = kTailEdges;
        if (const char *tl = getenv("KOMB_TAIL")) tail_limit = (uint32_t)strtoul(tl, nullptr, 10);
        if (tail_limit > kTailMaxEdges) tail_limit = kTailMaxEdges;
    } else if (fin == FIN_LOCAL) tail_limit = local_limit((uint64_t)m, 32);
    // (a graph small enough for the finish to take the whole peel is handed over before any step: no frontier may be queued)
    const bool whole_peel_finish = tail_limit && (uint64_t)m <= tail_limit;
    // the peel sharded by edge range, one exchange per sub-round (shard_dev.h): komb_set_shard_peel, or KOMB_SHARD_PEEL=1 (with
    // one rank: the same engine without a collective -- a test of its logic)
    const bool shard_peel_on = (ctx->shard_peel && world > 1) || getenv("KOMB_SHARD_PEEL") != nullptr;
    enum { IDX_STREAM = 0, IDX_SLICES = 1, IDX_TWO_PASS = 2 };
    int layout = IDX_STREAM;
    if (const char *ix = getenv("KOMB_INDEX")) {
        if (!strcmp(ix, "slices")) layout = IDX_SLICES;
        else if (!strcmp(ix, "two_pass")) layout = IDX_TWO_PASS;
        else if (strcmp(ix, "stream")) KOMB_FAIL(ctx, KOMB_ERR_ARG, "KOMB_INDEX=%s: expected stream, slices or two_pass", ix);
    }
    if (getenv("KOMB_TWO_PASS")) layout = IDX_TWO_PASS;
    uint32_t *d_cap = nullptr, *d_offc = nullptr;
    unsigned long long *d_offc64 = nullptr;    // the same offsets in 64 bits when the slices exceed 2^32 entries (KOMB_OFF64=1 forces them)
    int2 *d_sparse = nullptr;
    int2 *d_owndense = nullptr;                // own-role entries as compact per-task blocks (see k_triangles, DENSE)
    unsigned long long *d_ownoff = nullptr, *d_dcur = nullptr;
    uint32_t *d_cnt_ref = nullptr;             // world > 1: the all-reduced supports, kept to check the build against
    uint32_t *d_toff = nullptr;                // stream: first sorted record of every bin
    uint32_t *d_grp = nullptr;                 // the peel's ticket / init words (stream: allocated before the build's last kernels, which fill them)
    int32_t *d_light0 = nullptr;               // stream: light queue 0 of the peel, which k_bin_finish fills with the first frontier
    uint32_t *d_bintot = nullptr;              // stream: supports summed per bin, then their exclusive scan (every bin's window of the index)
    uint32_t *d_reckey = nullptr;              // stream: the sorted records' keys
    int2 *d_recval = nullptr;                  // stream: ... and values
    int64_t n_bins = 0;
    const TriStream no_stream{nullptr, nullptr, nullptr, 0ull, 0u};
#ifdef KOMB_DEBUG_SWITCHES
    const int ablate = getenv("KOMB_TRI_ABLATE") ? atoi(getenv("KOMB_TRI_ABLATE")) : 0;    // breaks results on purpose: debug builds only
#else
    const int ablate = 0;
#endif
    st.ms_sort = 0.0; st.tri_records = 0; st.ms_compact = 0.0; st.ms_tri_count = 0.0; st.ms_tri_fill = 0.0;
    st.index_layout = layout;
    auto zero_counts = [&]() -> hipError_t { return hipMemsetAsync(d_own, 0, 2 * ((size_t)m + 1) * sizeof(uint32_t), s); };
    bool have_counts = false;                  // d_cnt holds the supports (and d_mom[5] their sum)
    if (world > 1) {
        const int64_t task_lo = ntasks * rank / world, task_hi = ntasks * (rank + 1) / world;
        ctx->timer.start(s);
        k_triangles<TRI_COUNT><<<gt, kBlock, 0, s>>>(d_orow, d_ocol, nv, task_lo, task_hi, d_own, d_other, (const uint32_t *)nullptr, nullptr, nullptr, nullptr, 0ull, nullptr, ablate, no_stream, tri_tv);
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
        unsigned long long *d_bound = d_mom + 6;
        k_own_bound<<<gv, kBlock, 0, s>>>(d_orow, nv, d_bound);
        unsigned long long bound = 0;
        KOMB_HIP(ctx, d2h(ctx, &bound, d_bound, sizeof(bound)));
        const int gts = std::min(gt, 256 * KOMB_TRI_EU);             // resident workgroups only: every wavefront ends with one partly used claim
        const unsigned long long slack = (unsigned long long)gts * kTriWaves * kRecChunk + kRecChunk;
        unsigned long long own_cap = bound + bound / 8 + (unsigned long long)gts * kTriWaves * kOwnChunk + kOwnChunk;
        if (const char *oc = getenv("KOMB_OWN_DENSE_CAP")) own_cap = strtoull(oc, nullptr, 10) + 1;      // (tests: a region that runs out)
        if (getenv("KOMB_NO_OWN_DENSE")) own_cap = 1;                // (tests: every entry a record)
        unsigned long long t_bound = bound / 2;
        unsigned long long rec_cap = 3 * t_bound + (3 * t_bound) / 14 + slack;
        if (const char *rc = getenv("KOMB_REC_CAP")) rec_cap = strtoull(rc, nullptr, 10) + 1;            // (tests: a stream that runs out)
        // (what the pool holds unused counts as free: the choice must not depend on what an earlier call left cached)
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const unsigned long long budget = (unsigned long long)((free_b + ctx->pool.unused_bytes()) * 0.7);
        if (own_cap * sizeof(int2) > budget / 2) own_cap = budget / 2 / sizeof(int2);
        if (rec_cap * 12ull > budget / 2) rec_cap = budget / 2 / 12ull;
        if (rec_cap > 0xFFFFFFF0ull) rec_cap = 0xFFFFFFF0ull;        // 32-bit record positions
        int end_bit = 1;
        // the sentinel key (1 << end_bit) - 1 lies at least a whole bin above the last edge id, so the sort by bin puts it last
        while (end_bit < 32 && (1ull << end_bit) <= (unsigned long long)m + kBinEdges) ++end_bit;
        const uint32_t sentinel = end_bit >= 32 ? 0xFFFFFFFFu : (uint32_t)((1ull << end_bit) - 1ull);
        uint32_t *d_key = nullptr; int2 *d_val = nullptr;
        bool ok = bufs.alloc(&d_key, (size_t)rec_cap) == hipSuccess && bufs.alloc(&d_val, (size_t)rec_cap) == hipSuccess &&
                  bufs.alloc(&d_owndense, (size_t)own_cap) == hipSuccess && bufs.alloc(&d_ownoff, (size_t)m + 1) == hipSuccess &&
                  bufs.alloc(&d_dcur, 4) == hipSuccess;
        unsigned long long n_claimed = 0;
        if (ok) {
            KOMB_HIP(ctx, hipMemsetAsync(d_dcur, 0, 4 * sizeof(unsigned long long), s));
            const TriStream ts{d_key, d_val, d_dcur + 2, rec_cap, sentinel};
            ctx->timer.start(s);
            k_triangles<TRI_SINGLE, uint32_t, false, true, true><<<gts, kBlock, 0, s>>>(d_orow, d_ocol, nv, 0, ntasks, d_own, d_other, (const uint32_t *)nullptr, nullptr,
                                                                                      d_owndense, d_dcur, own_cap, d_ownoff, ablate, ts, tri_tv);
            st.ms_tri_fill = ctx->timer.stop(s);
            unsigned long long dc[4] = {0, 0, 0, 0};
            KOMB_HIP(ctx, d2h(ctx, dc, d_dcur, sizeof(dc)));
            n_claimed = dc[2];
            if (getenv("KOMB_TRI_DEBUG"))
                fprintf(stderr, "komb triangles: stream build: %llu record positions claimed of %llu, dense own-role region %llu entries claimed of %llu, %llu task ranges overflowed their record buffer\n",
                        dc[2], rec_cap, dc[0], own_cap, dc[1]);
            if (n_claimed > rec_cap) ok = false;                     // the stream ran out: records were dropped
        } else (void)hipGetLastError();
        uint32_t *d_skey = nullptr;
        if (ok) {
            // sort the records by destination edge
            uint32_t *d_key2 = nullptr; unsigned long long *d_val2 = nullptr;
            ok = bufs.alloc(&d_key2, (size_t)n_claimed + 1) == hipSuccess && bufs.alloc(&d_val2, (size_t)n_claimed + 1) == hipSuccess;
            if (ok) {
                // sort the records by BIN (the key bits above kBinBits): two radix passes instead of four
                ctx->timer.start(s);
                unsigned long long *sv = nullptr;
                if (end_bit > kBinBits) KOMB_TRY(prim_sort_pairs_u32_u64(ctx, d_key, d_key2, (unsigned long long *)d_val, d_val2, (int64_t)n_claimed, kBinBits, end_bit, &d_skey, &sv));
                else { d_skey = d_key; sv = (unsigned long long *)d_val; }      // a single bin: nothing to sort
                st.ms_sort = ctx->timer.stop(s);
                d_recval = (int2 *)sv;
                if (d_skey == d_key) { bufs.release(d_key2); bufs.release(d_val2); } else { bufs.release(d_key); bufs.release(d_val); }
                d_reckey = d_skey;
                n_bins = (m + kBinEdges - 1) >> kBinBits;
                KOMB_HIP(ctx, bufs.alloc(&d_toff, (size_t)n_bins + 2));
                KOMB_HIP(ctx, bufs.alloc(&d_bintot, (size_t)n_bins + 2));
                KOMB_HIP(ctx, bufs.alloc(&d_grp, (size_t)kInitOff + 4));
                if (!whole_peel_finish && !shard_peel_on && !getenv("KOMB_NO_FIRST_QUEUE")) KOMB_HIP(ctx, bufs.alloc(&d_light0, (size_t)m));
                ctx->timer.start(s);
                peel_ctrl_pre(s, d_grp);
                KOMB_HIP(ctx, hipMemsetAsync(d_mom + 5, 0, sizeof(unsigned long long), s));
                k_bin_offsets<<<grid_for(n_bins + 1), kBlock, 0, s>>>(d_skey, (int64_t)n_claimed, n_bins, d_toff);
                k_bin_count<<<grid_for(n_bins, 1, 256 * 8), kBlock, 0, s>>>(d_skey, d_toff, n_bins, d_own, m, d_cnt, d_bintot, d_mom + 5, (int32_t *)(d_grp + kInitOff + 1));
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

    if (layout == IDX_SLICES) {
        // every edge gets a slice sized by the bound sup(a->b) <= d(a)-1; ~26 GB at |E|=100M, which the 288 GB of HBM afford
        bool single = true;
        KOMB_HIP(ctx, bufs.alloc(&d_cap, (size_t)m + 1));
        KOMB_HIP(ctx, bufs.alloc(&d_offc, (size_t)m + 1));
        KOMB_HIP(ctx, hipMemsetAsync(d_cap + m, 0, sizeof(uint32_t), s));
        ctx->timer.start(s);
        k_slice_caps<<<ge, kBlock, 0, s>>>(d_osrc, d_deg, d_orow, m, d_cap, d_mom + 6);
        unsigned long long cap_both[2] = {0, 0};
        KOMB_HIP(ctx, d2h(ctx, cap_both, d_mom + 6, sizeof(cap_both)));
        const unsigned long long cap_total = cap_both[0];
        // dense own-role region: the bound, + what the wavefronts' chunked claims can leave unused
        unsigned long long own_cap = cap_both[1] + cap_both[1] / 8 + (unsigned long long)gt * kTriWaves * kOwnChunk + kOwnChunk;
        if (const char *oc = getenv("KOMB_OWN_DENSE_CAP")) own_cap = strtoull(oc, nullptr, 10) + 1;      // (tests: a region that runs out)
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        if (cap_total * sizeof(int2) > (unsigned long long)(free_b * 0.8)) single = false;
        const bool wide = cap_total > 0xFFFFFFF0ull || getenv("KOMB_OFF64") != nullptr;
        if (single) {
            if (wide) {
                bufs.release(d_offc); d_offc = nullptr;
                if (bufs.alloc(&d_offc64, (size_t)m + 1) != hipSuccess) { (void)hipGetLastError(); single = false; }
                else KOMB_TRY(prim_exclusive_sum_u32_u64(ctx, d_cap, d_offc64, m + 1));
            } else KOMB_TRY(prim_exclusive_sum_u32(ctx, d_cap, d_offc, m + 1));
            if (single && bufs.alloc(&d_sparse, (size_t)cap_total) != hipSuccess) { (void)hipGetLastError(); single = false; }
            // the dense own-role region is an optimisation: without the memory for it the slices take those entries too
            if (single && !getenv("KOMB_NO_OWN_DENSE")) {
                (void)hipMemGetInfo(&free_b, &total_b);
                if (own_cap * sizeof(int2) + (size_t)m * 8 < (unsigned long long)(free_b * 0.8) &&
                    bufs.alloc(&d_owndense, (size_t)own_cap) == hipSuccess && bufs.alloc(&d_ownoff, (size_t)m + 1) == hipSuccess &&
                    bufs.alloc(&d_dcur, 2) == hipSuccess) {
                    KOMB_HIP(ctx, hipMemsetAsync(d_dcur, 0, 2 * sizeof(unsigned long long), s));
                } else {
                    (void)hipGetLastError();
                    bufs.release(d_owndense); bufs.release(d_ownoff); bufs.release(d_dcur);
                    d_owndense = nullptr; d_ownoff = nullptr; d_dcur = nullptr;
                }
            }
        }
        if (single) {
            if (d_offc64) {
                if (d_owndense) k_triangles<TRI_SINGLE, unsigned long long, false, true><<<gt, kBlock, 0, s>>>(d_orow, d_ocol, nv, 0, ntasks, d_own, d_other, d_offc64, d_sparse, d_owndense, d_dcur, own_cap, d_ownoff, ablate, no_stream, tri_tv);
                else k_triangles<TRI_SINGLE, unsigned long long><<<gt, kBlock, 0, s>>>(d_orow, d_ocol, nv, 0, ntasks, d_own, d_other, d_offc64, d_sparse, nullptr, nullptr, 0ull, nullptr, ablate, no_stream, tri_tv);
            } else {
                k_back_cursors<<<ge, kBlock, 0, s>>>(d_offc, m, d_other);
                if (d_owndense) k_triangles<TRI_SINGLE, uint32_t, true, true><<<gt, kBlock, 0, s>>>(d_orow, d_ocol, nv, 0, ntasks, d_own, d_other, d_offc, d_sparse, d_owndense, d_dcur, own_cap, d_ownoff, ablate, no_stream, tri_tv);
                else k_triangles<TRI_SINGLE, uint32_t, true><<<gt, kBlock, 0, s>>>(d_orow, d_ocol, nv, 0, ntasks, d_own, d_other, d_offc, d_sparse, nullptr, nullptr, 0ull, nullptr, ablate, no_stream, tri_tv);
            }
            st.ms_tri_fill = ctx->timer.stop(s);
            KOMB_HIP(ctx, hipMemsetAsync(d_mom + 5, 0, sizeof(unsigned long long), s));
            k_sum_counts<<<ge, kBlock, 0, s>>>(d_own, d_other, d_offc64 ? nullptr : d_offc, m + 1, d_cnt, d_mom + 5);
        } else {
            (void)ctx->timer.stop(s);
            bufs.release(d_cap); bufs.release(d_offc); bufs.release(d_offc64); d_offc64 = nullptr;
            bufs.release(d_owndense); bufs.release(d_ownoff); bufs.release(d_dcur); d_owndense = nullptr; d_ownoff = nullptr; d_dcur = nullptr;
            layout = IDX_TWO_PASS;
        }
    }
#ifdef KOMB_TRI_PROFILE
    if (layout != IDX_TWO_PASS) {
        std::vector<unsigned long long> pr(2 * 16384);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(g_tri_prof), pr.size() * sizeof(unsigned long long));
        unsigned long long t0 = ~0ull, t1 = 0; std::vector<double> dur, endt;
        const int nwv = gt * kTriWaves < 16384 ? gt * kTriWaves : 16384;
        for (int i = 0; i < nwv; ++i) { if (pr[2 * i] < t0) t0 = pr[2 * i]; if (pr[2 * i + 1] > t1) t1 = pr[2 * i + 1]; }
        for (int i = 0; i < nwv; ++i) { dur.push_back((pr[2 * i + 1] - pr[2 * i]) / 100.0); endt.push_back((pr[2 * i + 1] - t0) / 100.0); }
        std::sort(dur.begin(), dur.end()); std::sort(endt.begin(), endt.end());
        auto q = [&](std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
        fprintf(stderr, "komb tri profile: %d waves, span %.0f us; wave busy time us: min %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; end time us: p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f\n",
                nwv, (t1 - t0) / 100.0, dur.front(), q(dur, .1), q(dur, .5), q(dur, .9), q(dur, .99), dur.back(), q(endt, .1), q(endt, .5), q(endt, .9), q(endt, .99), endt.back());
    }
#endif
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
            k_triangles<TRI_COUNT><<<gt, kBlock, 0, s>>>(d_orow, d_ocol, nv, 0, ntasks, d_own, d_other, (const uint32_t *)nullptr, nullptr, nullptr, nullptr, 0ull, nullptr, ablate, no_stream, tri_tv);
            st.ms_tri_count = ctx->timer.stop(s);
            k_sum_counts<<<ge, kBlock, 0, s>>>(d_own, d_other, nullptr, m + 1, d_cnt, d_mom + 5);
        }
    }
    st.index_layout = layout;
    // graph statistics for the roofline model (sum d^2, sum min(d,d), max d, sum d+ + d+): properties of
    // the graph, not results of the path -- computed on the first whole-graph run and on every subgraph run
    const bool want_moments = vmask_host != nullptr || !ctx->moments_valid;
    if (want_moments) k_graph_moments<<<1024, kBlock, 0, s>>>(d_deg, nv, d_osrc, d_ocol, m, d_cnt, d_orow, d_mom);
    {
        unsigned long long mom[9];
        KOMB_HIP(ctx, d2h(ctx, mom, d_mom, sizeof(mom)));
        if (want_moments) {
            st.sum_deg_sq = (int64_t)mom[0]; st.wedge_items = (int64_t)mom[1]; st.max_degree = (int32_t)mom[2];
            st.oriented_items = (int64_t)mom[4];
            ctx->moments_valid = vmask_host == nullptr;
        }
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
    int2 *d_inc = nullptr;
    int32_t *d_sup = nullptr, *d_stamp = nullptr, *d_truss = nullptr;
    bool peel_inited = false;                  // the stream layout's finish also writes the peel's initial state
    const int gc = grid_for((m + kWave - 1) / kWave, kBlock / kWave);
    if (layout == IDX_STREAM) {
        // every bin's window of the index from the scan of the per-bin totals (the slice offsets themselves are a workgroup scan inside k_bin_finish)
        KOMB_TRY(prim_exclusive_sum_u32(ctx, d_bintot, d_bintot, n_bins + 1));
        KOMB_HIP(ctx, d2h(ctx, &total, d_bintot + n_bins, sizeof(uint32_t)));
        KOMB_HIP(ctx, bufs.alloc(&d_inc, (size_t)total));
        KOMB_HIP(ctx, bufs.alloc(&d_sup, (size_t)m));
        KOMB_HIP(ctx, bufs.alloc(&d_stamp, (size_t)m));
        KOMB_HIP(ctx, bufs.alloc(&d_truss, (size_t)m));
        ctx->timer.start(s);
        k_bin_finish<<<grid_for(n_bins, 1, 256 * 2), kFinBlock, 0, s>>>(d_reckey, d_recval, d_toff, n_bins, d_own, d_cnt, d_bintot, d_owndense, d_ownoff, d_inc, m,
                                                                      d_off, d_sup, d_stamp, d_truss, d_grp + kInitOff, d_light0);
        st.ms_compact += ctx->timer.stop(s);
        peel_inited = true;
        bufs.release(d_toff); bufs.release(d_bintot); bufs.release((void *)d_recval); bufs.release(d_reckey);
        bufs.release(d_owndense); bufs.release(d_ownoff); bufs.release(d_dcur);
    } else {
        KOMB_TRY(prim_exclusive_sum_u32(ctx, d_cnt, d_off, m + 1));
        KOMB_HIP(ctx, d2h(ctx, &total, d_off + m, sizeof(uint32_t)));
        KOMB_HIP(ctx, bufs.alloc(&d_inc, (size_t)total));
    }
    if (layout == IDX_SLICES) {
        ctx->timer.start(s);
        if (d_offc64) k_compact_inc<unsigned long long><<<gc, kBlock, 0, s>>>(d_offc64, d_own, d_off, d_sparse, d_owndense, d_ownoff, d_inc, m);
        else k_compact_inc<uint32_t><<<gc, kBlock, 0, s>>>(d_offc, d_own, d_off, d_sparse, d_owndense, d_ownoff, d_inc, m);
        st.ms_compact = ctx->timer.stop(s);
        if (d_dcur && getenv("KOMB_TRI_DEBUG")) {
            unsigned long long dc[2] = {0, 0};
            KOMB_HIP(ctx, d2h(ctx, dc, d_dcur, sizeof(dc)));
            fprintf(stderr, "komb triangles: dense own-role region: %llu entries claimed (%llu would be exact for all), %llu task ranges overflowed their record buffer\n",
                    dc[0], 2ull * (unsigned long long)total / 3ull, dc[1]);
        }
        bufs.release(d_sparse); bufs.release(d_cap); bufs.release(d_offc); bufs.release(d_offc64);
        bufs.release(d_owndense); bufs.release(d_ownoff); bufs.release(d_dcur);
    } else if (layout == IDX_TWO_PASS) {
        // second enumeration, same writer as the single-pass layouts but into the EXACT slices: own-role
        // entries from the front and third-role entries from the back meet precisely -- no compaction
        ctx->timer.start(s);
        KOMB_HIP(ctx, zero_counts());
        k_back_cursors<<<ge, kBlock, 0, s>>>(d_off, m, d_other);
        k_triangles<TRI_SINGLE, uint32_t, true><<<gt, kBlock, 0, s>>>(d_orow, d_ocol, nv, 0, ntasks, d_own, d_other, d_off, d_inc, nullptr, nullptr, 0ull, nullptr, ablate, no_stream, tri_tv);
        st.ms_tri_fill = ctx->timer.stop(s);
    }
    st.ms_support = st.ms_tri_count + st.ms_tri_fill + st.ms_sort + st.ms_compact;
    bufs.release(d_cnt); bufs.release(d_own);

    // ---- peel
    phase.next("truss: peel");
    PeelCtrl *d_ctrl = nullptr;
    PeelQueues Q{nullptr, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, scan_scalar_switch()};
    const size_t heavy_cap = (size_t)total / 32 + 64;             // see kcore.hip
    if (!peel_inited) {
        KOMB_HIP(ctx, bufs.alloc(&d_sup, (size_t)m));
        KOMB_HIP(ctx, bufs.alloc(&d_stamp, (size_t)m));
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
    TrussProblem P{(uint32_t)m, d_off, d_inc, d_sup, d_stamp, d_truss};
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
        const bool dbg = getenv("KOMB_TAIL_DEBUG") != nullptr;
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
            (uint32_t)kWave * TrussLocal::kU, sizeof(uint2), local_item_limit(kTrussLocalItems), local_density_limit(kTrussLocalDensity), true, 2, d_truss,
            [&](const LocalGraph &lg, const int32_t *num, void *items, PeelCtrl *d_cctrl, int32_t launch) {
                TrussCollect C{(uint32_t)m, d_off, d_inc, d_stamp, num, lg.off, lg.cur, (uint2 *)items};
                k_peel_step<TrussCollect><<<gp, kPeelBlock, 0, s>>>(d_cctrl, d_grp, Q, C, launch);
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
        k_peel_init<<<grid_for(m, kBlock, 1024), kBlock, 0, s>>>(m, d_off, d_sup, d_stamp, d_truss, d_grp + kInitOff);
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
        ShardTruss SP{(uint32_t)m, d_off, d_inc, d_sup, d_stamp, d_truss, 0u, 0u};
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
        const double n = c.pad1[7] ? (double)c.pad1[7] : 1.0;
        fprintf(stderr, "komb step timers (block 0, %u small multi-workgroup PROCESS steps), us per step: ctrl %.2f queue+slice %.2f items %.2f flush %.2f barrier %.2f ticket %.2f\n",
                c.pad1[7], c.pad1[0] / n / 100.0, c.pad1[1] / n / 100.0, c.pad1[2] / n / 100.0, c.pad1[3] / n / 100.0, c.pad1[4] / n / 100.0, c.pad1[5] / n / 100.0);
    }
#endif
    if (ctx->h_ctrl[0].done != 1) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "k-truss peel ended in an inconsistent state");
    st.truss_levels = ctx->h_ctrl[0].n_levels;
    st.truss_subrounds = ctx->h_ctrl[0].n_rounds;
    st.truss_launches = launches;
    st.truss_scans = ctx->h_ctrl[0].n_scans;
    st.max_trussness = ctx->h_ctrl[0].max_level + 2;
    for (int i = 0; i < 2; ++i) { bufs.release(Q.light[i]); bufs.release(Q.heavy[i]); bufs.release(Q.live[i]); }
    bufs.release(Q.code);
    bufs.release(d_stamp); bufs.release(d_sup); bufs.release(d_inc);

    // ---- canonical-order results with original vertex ids
    phase.next("truss: canonical gather");
    ctx->timer.start(s);
    const int64_t nwords = (w_ns + 63) / 64;
    KOMB_TRY(prim_exclusive_sum_u32(ctx, d_urank, d_urank, nwords + 1));     // upper slots before every word = canonical id of its first upper slot
    KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_eu, (size_t)m * sizeof(int32_t)));
    KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_ev, (size_t)m * sizeof(int32_t)));
    KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_truss, (size_t)m * sizeof(int32_t)));
    KOMB_HIP(ctx, ctx->pool.get((void **)&ctx->d_t_sup, (size_t)m * sizeof(int32_t)));
    // the reversed oriented slots (source id above target id), stably sorted by target: see k_gather_canonical
    uint32_t *d_kurank = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_kurank, (size_t)nwords + 1));
    k_popc_words<<<grid_for(nwords + 1), kBlock, 0, s>>>(d_kubits, nwords, d_kurank);
    KOMB_TRY(prim_exclusive_sum_u32(ctx, d_kurank, d_kurank, nwords + 1));
    uint32_t kept_upper = 0;
    KOMB_HIP(ctx, d2h(ctx, &kept_upper, d_kurank + nwords, sizeof(uint32_t)));
    const int64_t n_rev = m - (int64_t)kept_upper;
    uint32_t *d_rkey = nullptr, *d_rkey2 = nullptr;
    unsigned long long *d_rval = nullptr, *d_rval2 = nullptr, *d_rev_sorted = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_rkey, (size_t)n_rev + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_rkey2, (size_t)n_rev + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_rval, (size_t)n_rev + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_rval2, (size_t)n_rev + 1));
    d_rev_sorted = d_rval;
    if (n_rev > 0) {
        k_rev_emit<<<grid_for(w_ns), kBlock, 0, s>>>(w_col, w_ns, d_obits, d_wrank, d_kubits, d_kurank, d_truss, d_off, d_rkey, d_rval);
        int vbits = 1;
        while (vbits < 32 && (1ll << vbits) < nv) ++vbits;
        uint32_t *sk = nullptr;
        KOMB_TRY(prim_sort_pairs_u32_u64(ctx, d_rkey, d_rkey2, d_rval, d_rval2, n_rev, 0, vbits, &sk, &d_rev_sorted));
    }
    k_gather_canonical<<<grid_for(w_ns), kBlock, 0, s>>>(w_src, w_col, w_ns, d_obits, d_wrank, d_kubits, d_kurank, d_rev_sorted, d_urank, d_truss, d_off,
                                                        ctx->d_t_eu, ctx->d_t_ev, ctx->d_t_truss, ctx->d_t_sup);
    st.ms_gather = ctx->timer.stop(s);
    if (getenv("KOMB_POOL_DEBUG")) {
        size_t held = 0;
        for (const auto &b : ctx->pool.blocks) held += b.bytes;
        fprintf(stderr, "komb pool: %zu blocks, %.2f GB held, %zu hipMalloc calls so far, %zu trims\n", ctx->pool.blocks.size(), held / 1e9, ctx->pool.n_malloc, ctx->pool.n_trim);
    }
    ctx->t_ne = m;
    ctx->truss_done = true;
    return KOMB_OK;
}

} // namespace komb
