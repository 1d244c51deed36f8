// truss_tail.h -- the end of the k-truss peel in ONE workgroup's LDS.
//
// On power-law unitig graphs the high truss levels live in a small dense core
// (at |E| = 100M: 32k edges on ~600 vertices from trussness 23 on, 18 more
// levels, ~250 sub-rounds).  Down there a sub-round of the general engine
// (peel_dev.h) is a chain of ~10 dependent trips to memory and an arrival ticket
// -- ~20 us however few edges it peels.  Once at most kTailEdges edges are
// left the peel hands over to k_truss_tail: the live sub-graph is renumbered,
// its adjacency becomes a bit matrix in LDS, the live supports 16-bit counters
// in LDS indexed by the rank of the edge in the matrix; a triangle {u,v,x} is a
// set bit of row(u) & row(v), a decrement an LDS atomic, a sub-round two
// __syncthreads().  Same peel rule as TrussProblem (a triangle is destroyed once,
// by its frontier edge, or by the smaller edge key when two of its edges are in
// the frontier), so trussness is identical.
//
// Nothing of the general engine's state is modified until the tail has
// succeeded: when the sub-graph does not fit (too many vertices for the LDS
// layout) the kernel clears `done`, halves the hand-over threshold and the
// general engine simply continues.
#pragma once

#include "peel_dev.h"

namespace komb {
namespace {

constexpr uint32_t kTailEdges = 32768;          // hand over once this few edges are live (default; KOMB_TAIL overrides)
constexpr uint32_t kTailMaxEdges = 65534;       // edge ranks and supports are 16-bit
constexpr uint32_t kTailMaxV = 1024;            // bit rows of <= 16 words: one 16-lane group intersects two rows
constexpr uint32_t kTailRowWords = kTailMaxV / 64;
constexpr uint32_t kTailPoolWords = 40448;      // 161,792 B of LDS for matrix + ranks + supports + queues
constexpr uint32_t kTailPark = 128;              // per-wave buffer of parked triangles (< 64 left + <= 64 new)
constexpr uint32_t kTailMinQueue = 512;         // LDS queue entries below which the layout is refused
constexpr uint32_t kTFrontier = 0xFFFFu, kTDead = 0xFFFEu;   // support values are < kTDead

struct TailBufs {
    int32_t *vmap;                 // [nv]        0, or tail vertex number + 1
    uint32_t *cnt;                 // [64]        0: vertices, 1: live edges, 2: refused, 3: edges written, 4-5: sum of the live
                                   //             supports (64-bit), 8-15: timers, 16-47: which support values occur (1024 bits)
    int32_t *vlist;                // [kTailMaxV] original ids of the tail vertices
    unsigned long long *rows;      // [kTailMaxV * kTailRowWords] adjacency bits, stride kTailRowWords
    uint32_t *pair;                // [E] (lo << 16) | hi, tail vertex numbers
    int32_t *sup;                  // [E] live support
    int32_t *gid;                  // [E] edge id of the general engine
    int32_t *gid_by_rank;          // [E]
    uint32_t *pair_by_rank;        // [E]
    int32_t *truss_by_rank;        // [E]
    uint32_t *spill[2];            // [E] frontier entries beyond the LDS queues
    uint32_t max_edges;            // capacity of the [E] arrays
};

// ---- setup 1: number the endpoints of the live edges, count the edges
__global__ __launch_bounds__(kBlock) void k_tail_mark(const int32_t *__restrict__ list, uint32_t n_in, const int32_t *__restrict__ sup,
                                                      const int32_t *__restrict__ stamp,
                                                      const int32_t *__restrict__ osrc, const int32_t *__restrict__ ocol, TailBufs T)
{
    __shared__ uint32_t seen[32];                       // which support values this workgroup met (1024 bits)
    if (threadIdx.x < 32) seen[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t live = 0;
    unsigned long long work = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_in; i += gridDim.x * kBlock) {
        const int32_t e = list ? list[i] : (int32_t)i;
        if (!marker_alive(stamp[e])) continue;
        ++live;
        const uint32_t sv = (uint32_t)sup[e];
        work += sv;
        atomicOr(&seen[min(sv, 1023u) >> 5], 1u << (min(sv, 1023u) & 31u));
        const int32_t ab[2] = {osrc[e], ocol[e]};
        for (int k = 0; k < 2; ++k)
            if (atomicExch(&T.vmap[ab[k]], 1) == 0) {
                const uint32_t id = atomicAdd(&T.cnt[0], 1u);
                if (id < kTailMaxV) T.vlist[id] = ab[k];
            }
    }
    live = wave_sum(live);
    for (int o = 32; o > 0; o >>= 1) work += __shfl_xor(work, o);
    if (lane_id() == 0 && live) {
        atomicAdd(&T.cnt[1], live);
        atomicAdd(reinterpret_cast<unsigned long long *>(&T.cnt[4]), work);
    }
    __syncthreads();
    if (threadIdx.x < 32 && seen[threadIdx.x]) atomicOr(&T.cnt[16 + threadIdx.x], seen[threadIdx.x]);
}

// ---- setup 2 (one workgroup): vertex numbers in ascending original id (deterministic whatever order
// setup 1's atomics ran in), or refusal.
// Refusal also when one workgroup would be the slower choice.  The tail visits every live triangle from each of
// its edges once (the sum S of the live supports) at ~0.8 M per ms and pays ~5 us per sub-round; the general engine
// pays ~20 us per sub-round but has the whole chip for the triangles (~13 G per s).  With D distinct live support
// values (about the number of levels left, ~12-14 sub-rounds each) the tail wins when S / 800 + 70 D < 240 D + S / 13000
// (microseconds), i.e. S < ~145 000 D: a clique-like remainder (K_250: S = 7.7 M, D = 1 -- 9 ms here, 0.3 ms
// there) stays with the general engine, the many-level dense cores of power-law graphs do not.
constexpr unsigned long long kTailWorkPerLevel = 145000ull;
__global__ __launch_bounds__(kTailMaxV) void k_tail_number(TailBufs T)
{
    __shared__ int32_t ids[kTailMaxV];
    __shared__ uint32_t distinct;
    const uint32_t n = T.cnt[0], ne = T.cnt[1];
    if (threadIdx.x == 0) distinct = 0;
    __syncthreads();
    if (threadIdx.x < 32) atomicAdd(&distinct, (uint32_t)__popc(T.cnt[16 + threadIdx.x]));
    __syncthreads();
    const unsigned long long work = *reinterpret_cast<const unsigned long long *>(&T.cnt[4]);
    if (n > kTailMaxV || ne > kTailMaxEdges || ne > T.max_edges || ne == 0 || work > kTailWorkPerLevel * distinct) {
        if (threadIdx.x == 0) T.cnt[2] = 1u;
        return;
    }
    const uint32_t t = threadIdx.x;
    if (t < n) ids[t] = T.vlist[t];
    __syncthreads();
    if (t < n) {
        const int32_t mine = ids[t];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += ids[j] < mine ? 1u : 0u;
        T.vmap[mine] = (int32_t)rank + 1;
        T.vlist[rank] = mine;
    }
}

// ---- setup 3: adjacency bits and the compact list of live edges
__global__ __launch_bounds__(kBlock) void k_tail_rows(const int32_t *__restrict__ list, uint32_t n_in, const int32_t *__restrict__ sup,
                                                      const int32_t *__restrict__ stamp,
                                                      const int32_t *__restrict__ osrc, const int32_t *__restrict__ ocol, TailBufs T)
{
    if (T.cnt[2]) return;
    for (uint32_t i0 = blockIdx.x * kBlock; i0 < n_in; i0 += gridDim.x * kBlock) {
        const uint32_t i = i0 + threadIdx.x;
        int32_t e = -1;
        int32_t st = 0;
        if (i < n_in) { e = list ? list[i] : (int32_t)i; st = stamp[e]; }
        const bool live = e >= 0 && marker_alive(st);
        const uint64_t m = __ballot(live);
        uint32_t base = 0;
        if (lane_id() == 0 && m) base = atomicAdd(&T.cnt[3], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, 0);
        if (!live) continue;
        const uint32_t u = (uint32_t)T.vmap[osrc[e]] - 1u, v = (uint32_t)T.vmap[ocol[e]] - 1u;
        atomicOr(&T.rows[(size_t)u * kTailRowWords + (v >> 6)], 1ull << (v & 63));
        atomicOr(&T.rows[(size_t)v * kTailRowWords + (u >> 6)], 1ull << (u & 63));
        const uint32_t j = base + (uint32_t)__popcll(m & lanemask_lt());
        T.pair[j] = (min(u, v) << 16) | max(u, v);
        T.sup[j] = sup[e];
        T.gid[j] = e;
    }
}

// ---- the tail peel
// LDS layout.  A: the LIVE adjacency matrix (n rows of W words; a peeled edge's bits are cleared, so
// row(u) & row(v) is exactly the set of live triangles of {u,v}).  U: the ORIGINAL upper adjacency
// (bits x > u of row u only, stored triangularly: row u keeps words u/64 .. W-1), immutable: with
// `pre` (set bits of the row before each word) and `base` (edges of the rows before) it gives every
// edge {lo,hi} a dense rank -- the index of its 16-bit support counter in S.
struct TailLds {
    unsigned long long *A;         // [n * W]
    unsigned long long *U;         // [tri(n)]
    uint16_t *pre;                 // [tri(n)]
    int32_t *base;                 // [n]
    uint32_t *S;                   // [(E+1)/2] two 16-bit live supports per word; kTFrontier / kTDead when peeled
    uint32_t *park;                // [16 * kTailPark] per-wave buffers of parked triangles
    uint32_t *q[2];                // [cap]    frontier queues of pair codes
    uint32_t W, cap;
};

// first word of row u in the triangular arrays; tail_tri(n, W) = their size
__device__ __forceinline__ uint32_t tail_tri(uint32_t u, uint32_t W)
{
    const uint32_t b = u >> 6;
    return 64u * (b * W - b * (b - 1) / 2) + (u - 64u * b) * (W - b);
}
__device__ __forceinline__ unsigned long long tail_upmask(uint32_t u, uint32_t j)
{
    const uint32_t ju = u >> 6;
    if (j > ju) return ~0ull;
    if (j < ju) return 0ull;
    return (u & 63) == 63 ? 0ull : (~0ull << ((u & 63) + 1));
}
// rank of edge {lo,hi}, lo < hi, among all edges in (lo,hi)-lexicographic order
__device__ __forceinline__ uint32_t tail_rank(const TailLds &t, uint32_t lo, uint32_t hi)
{
    const uint32_t w = tail_tri(lo, t.W) + (hi >> 6) - (lo >> 6);
    const unsigned long long bits = t.U[w] & ((1ull << (hi & 63)) - 1ull);
    return (uint32_t)t.base[lo] + (uint32_t)t.pre[w] + (uint32_t)__popcll(bits);
}
__device__ __forceinline__ uint32_t tail_sup(const TailLds &t, uint32_t r) { return (t.S[r >> 1] >> (16 * (r & 1))) & 0xFFFFu; }
// live -> frontier (0xFFFF) -> dead (0xFFFE): atomics, because the other half of the word is another edge
__device__ __forceinline__ void tail_to_frontier(const TailLds &t, uint32_t r) { atomicOr(&t.S[r >> 1], 0xFFFFu << (16 * (r & 1))); }
__device__ __forceinline__ void tail_to_dead(const TailLds &t, uint32_t r) { atomicAnd(&t.S[r >> 1], ~(1u << (16 * (r & 1)))); }

__global__ __launch_bounds__(1024) void k_truss_tail(PeelCtrl *ctrl, TailBufs T, int32_t *__restrict__ truss)
{
    __shared__ __attribute__((aligned(16))) uint32_t pool[kTailPoolWords];
    __shared__ uint32_t sh_cnt[2];                 // entries of the two queues
    __shared__ int32_t sh_min;
    __shared__ int32_t sh_ok;
    const uint32_t tid = threadIdx.x;
    const int lane = lane_id();
#ifdef KOMB_TAIL_TIMERS
    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
    auto tick = [&](int k) { const unsigned long long now = wall_clock64(); tm[k] += now - tlast; tlast = now; };
#else
    auto tick = [](int) {};
#endif
    const uint32_t n = T.cnt[0], E = T.cnt[1];
    TailLds t;
    t.W = (n + 63) / 64;
    // ---- layout (32-bit words)
    const uint32_t tri = tail_tri(n, t.W);
    const uint32_t wA = n * t.W * 2, wU = tri * 2, wPre = (tri + 1) / 2, wBase = n, wS = (E + 1) / 2, wPark = 16 * kTailPark;
    const uint32_t used = wA + wU + wPre + wBase + wS + wPark;
    const bool ok = T.cnt[2] == 0 && T.cnt[3] == E && n <= kTailMaxV && used + 2 * kTailMinQueue <= kTailPoolWords;
    if (!ok) {
        // refused: the general engine goes on; try again when half as many edges are left
        if (tid == 0) { ctrl->tail_limit = ctrl->remaining / 2; ctrl->done = 0; }
        return;
    }
    t.A = reinterpret_cast<unsigned long long *>(pool);
    t.U = reinterpret_cast<unsigned long long *>(pool + wA);
    t.pre = reinterpret_cast<uint16_t *>(pool + wA + wU);
    t.base = reinterpret_cast<int32_t *>(pool + wA + wU + wPre);
    t.S = pool + wA + wU + wPre + wBase;
    t.park = t.S + wS;
    t.cap = (kTailPoolWords - used) / 2;
    t.q[0] = pool + used;
    t.q[1] = pool + used + t.cap;

    // ---- matrices, rank tables, supports
    for (uint32_t w = tid; w < n * t.W; w += 1024) {
        const uint32_t u = w / t.W, j = w % t.W;
        const unsigned long long bits = T.rows[(size_t)u * kTailRowWords + j];
        t.A[w] = bits;
        if (j >= (u >> 6)) t.U[tail_tri(u, t.W) + j - (u >> 6)] = bits & tail_upmask(u, j);
    }
    for (uint32_t w = tid; w < wS; w += 1024) t.S[w] = 0u;
    if (tid == 0) { sh_cnt[0] = sh_cnt[1] = 0u; sh_ok = 1; }
    __syncthreads();
    tick(6);
    if (tid < n) {
        uint32_t run = 0;
        const uint32_t w0 = tail_tri(tid, t.W), nw = t.W - (tid >> 6);
        for (uint32_t j = 0; j < nw; ++j) {
            t.pre[w0 + j] = (uint16_t)run;
            run += (uint32_t)__popcll(t.U[w0 + j]);
        }
        t.base[tid] = (int32_t)run;                // upper degree, scanned below
    }
    __syncthreads();
    if (tid < kWave) {                             // exclusive scan of <= 1024 upper degrees by one wavefront
        uint32_t carry = 0;
        for (uint32_t c0 = 0; c0 < n; c0 += kWave) {
            const uint32_t i = c0 + (uint32_t)lane;
            const uint32_t v = i < n ? (uint32_t)t.base[i] : 0u;
            const uint32_t incl = wave_incl_scan(v);
            if (i < n) t.base[i] = (int32_t)(carry + incl - v);
            carry += (uint32_t)__shfl((int)incl, kWave - 1);
        }
        if (lane == 0 && carry != E) sh_ok = 0;    // bits and edge list disagree: refuse
    }
    __syncthreads();
    if (!sh_ok) {
        if (tid == 0) { ctrl->tail_limit = 0; ctrl->done = 0; }
        return;
    }
    tick(7);
    for (uint32_t j = tid; j < E; j += 1024) {
        const uint32_t code = T.pair[j], lo = code >> 16, hi = code & 0xFFFFu;
        const uint32_t r = tail_rank(t, lo, hi);
        atomicOr(&t.S[r >> 1], ((uint32_t)T.sup[j] & 0xFFFFu) << (16 * (r & 1)));
        T.gid_by_rank[r] = T.gid[j];
        T.pair_by_rank[r] = code;
    }
    __syncthreads();

    auto q_get = [&](int sel, uint32_t i) -> uint32_t {
        const uint32_t *q = sel ? t.q[1] : t.q[0];
        uint32_t *sp = sel ? T.spill[1] : T.spill[0];
        return i < t.cap ? q[i] : __hip_atomic_load(&sp[i - t.cap], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // append `code` of the lanes with `pred`; reached by all 64 lanes (one LDS atomic per wavefront:
    // same-address LDS atomics serialise, a 2000-edge frontier pushed lane by lane cost ~15 us)
    auto q_push = [&](int sel, uint32_t code) {
        uint32_t *q = sel ? t.q[1] : t.q[0];
        uint32_t *sp = sel ? T.spill[1] : T.spill[0];
        const uint32_t slot = atomicAdd(&sh_cnt[sel], 1u);
        if (slot < t.cap) q[slot] = code;
        else __hip_atomic_store(&sp[slot - t.cap], code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    tick(0);
    int32_t L = ctrl->level;
    uint32_t alive = E, rounds = 0, levels = 0;
    int32_t max_level = ctrl->max_level;
    int sel = 0;
    int32_t state = 1;                             // 1 done, 2 inconsistent
    const uint32_t gl = (uint32_t)lane & 15u;
    uint32_t *park = t.park + (tid >> 6) * kTailPark;
    while (alive > 0) {
        // ---- SCAN, linear in the rank-ordered supports.  Pass 1: the smallest live support = the next
        // populated level (everything <= the level just finished is gone).  Pass 2: the edges at that
        // level become the frontier; their vertex pairs come from the rank-ordered copy of the edge list.
        if (tid == 0) { sh_cnt[sel] = 0u; sh_cnt[sel ^ 1] = 0u; sh_min = 0x7FFFFFFF; }
        __syncthreads();
        int32_t lmin = 0x7FFFFFFF;
        for (uint32_t w = tid; w < wS; w += 1024) {
            const uint32_t two = t.S[w];
            const uint32_t s0 = two & 0xFFFFu, s1 = two >> 16;
            if (s0 < kTDead) lmin = min(lmin, (int32_t)s0);
            if (s1 < kTDead && 2 * w + 1 < E) lmin = min(lmin, (int32_t)s1);
        }
        lmin = wave_min(lmin);
        if (lane == 0 && lmin != 0x7FFFFFFF) atomicMin(&sh_min, lmin);
        __syncthreads();
        if (sh_min == 0x7FFFFFFF) { state = 2; break; }
        L = max(L, sh_min);
        for (uint32_t w = tid; w < wS; w += 1024) {
            const uint32_t two = t.S[w];
#pragma unroll
            for (uint32_t h = 0; h < 2; ++h) {
                const uint32_t s = (two >> (16 * h)) & 0xFFFFu, r = 2 * w + h;
                if (s < kTDead && r < E && (int32_t)s <= L) {
                    q_push(sel, __hip_atomic_load(&T.pair_by_rank[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    tail_to_frontier(t, r);
                    T.truss_by_rank[r] = L + 2;
                }
            }
        }
        __syncthreads();
        uint32_t ncur = sh_cnt[sel];
        tick(1);
        ++levels; max_level = L;
        while (ncur > 0) {
            // ---- destroy the triangles of the frontier edges.  Extraction: a 16-lane group per edge, a matrix
            // word per lane, one common live neighbour x per lane per trip -- cheap, but the lanes' bit counts
            // are uneven.  So the triangles {u,v,x} are parked in the wave's LDS buffer and the expensive part
            // (two rank lookups, two support reads, the decrements) runs on 64 parked triangles at a time.
            uint32_t npark = 0;                                     // wave-uniform
            auto handle = [&](bool valid, uint32_t item) {           // reached by all 64 lanes
                const uint32_t u = item & 1023u, v = (item >> 10) & 1023u, x = item >> 20;
                const uint32_t code = (u << 16) | v;
                const uint32_t k1 = (min(u, x) << 16) | max(u, x), k2 = (min(v, x) << 16) | max(v, x);
                bool t1 = false, t2 = false;
                if (valid) {
                    const uint32_t r1 = tail_rank(t, k1 >> 16, k1 & 0xFFFFu), r2 = tail_rank(t, k2 >> 16, k2 & 0xFFFFu);
                    const uint32_t s1 = tail_sup(t, r1), s2 = tail_sup(t, r2);
                    const bool f1 = s1 == kTFrontier, f2 = s2 == kTFrontier;      // (a dead edge has no bit in A)
                    if (!f1 && (!f2 || code < k2)) {
                        const uint32_t sh = 16 * (r1 & 1);
                        t1 = (int32_t)((atomicSub(&t.S[r1 >> 1], 1u << sh) >> sh) & 0xFFFFu) == L + 1;
                    }
                    if (!f2 && (!f1 || code < k1)) {
                        const uint32_t sh = 16 * (r2 & 1);
                        t2 = (int32_t)((atomicSub(&t.S[r2 >> 1], 1u << sh) >> sh) & 0xFFFFu) == L + 1;
                    }
                }
                if (t1) q_push(sel ^ 1, k1);
                if (t2) q_push(sel ^ 1, k2);
            };
            for (uint32_t i0 = (tid >> 6) * 4; i0 < ncur; i0 += 64) {
                const uint32_t i = i0 + ((uint32_t)lane >> 4);
                uint32_t u = 0, v = 0;
                unsigned long long bits = 0ull;
                if (i < ncur) {
                    const uint32_t code = q_get(sel, i);
                    u = code >> 16; v = code & 0xFFFFu;
                    if (gl < t.W) bits = t.A[u * t.W + gl] & t.A[v * t.W + gl];
                }
                const uint32_t uv = u | (v << 10);
                while (true) {
                    const bool has = bits != 0ull;
                    const uint64_t m = __ballot(has);
                    if (m == 0) break;
                    if (has) {
                        const uint32_t x = gl * 64 + (uint32_t)__ffsll((long long)bits) - 1u;
                        bits &= bits - 1ull;
                        park[npark + (uint32_t)__popcll(m & lanemask_lt())] = uv | (x << 20);
                    }
                    npark += (uint32_t)__popcll(m);
                    __builtin_amdgcn_wave_barrier();
                    if (npark >= (uint32_t)kWave) {
                        npark -= kWave;
                        const uint32_t item = park[npark + (uint32_t)lane];
                        __builtin_amdgcn_wave_barrier();
                        handle(true, item);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (npark) handle((uint32_t)lane < npark, park[(uint32_t)lane < npark ? lane : 0]);
            __syncthreads();
            tick(3);
            // ---- the frontier leaves the matrix; the triggered edges are the next frontier
            const uint32_t nnext = sh_cnt[sel ^ 1];
            if (tid == 0) sh_cnt[sel] = 0u;        // this queue is the next sub-round's output (its entries are still read below, by count)
            for (uint32_t i = tid; i < ncur; i += 1024) {
                const uint32_t code = q_get(sel, i);
                const uint32_t u = code >> 16, v = code & 0xFFFFu;
                tail_to_dead(t, tail_rank(t, u, v));
                atomicAnd(&t.A[u * t.W + (v >> 6)], ~(1ull << (v & 63)));
                atomicAnd(&t.A[v * t.W + (u >> 6)], ~(1ull << (u & 63)));
            }
            for (uint32_t i = tid; i < nnext; i += 1024) {
                const uint32_t code = q_get(sel ^ 1, i);
                const uint32_t r = tail_rank(t, code >> 16, code & 0xFFFFu);
                tail_to_frontier(t, r);
                T.truss_by_rank[r] = L + 2;
            }
            __syncthreads();
            tick(4);
            alive -= ncur;
            ++rounds;
            ncur = nnext;
            sel ^= 1;
        }
        L += 1;
    }
    __syncthreads();
    // ---- results into the general engine's arrays, statistics, done
    for (uint32_t r = tid; r < E; r += 1024)
        truss[__hip_atomic_load(&T.gid_by_rank[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] =
            __hip_atomic_load(&T.truss_by_rank[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tick(5);
    if (tid == 0) {
#ifdef KOMB_TAIL_TIMERS
        for (int k = 0; k < 8; ++k) T.cnt[8 + k] = (uint32_t)tm[k];
#endif
        ctrl->remaining = alive;
        ctrl->n_levels += (int32_t)levels;
        ctrl->n_rounds += (int32_t)rounds;
        ctrl->n_scans += (int32_t)levels;
        ctrl->max_level = max_level;
        ctrl->level = L;
        ctrl->done = state;
    }
}

} // namespace
} // namespace komb
