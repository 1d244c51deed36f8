// core_tail.h -- the end of the k-core peel in ONE workgroup's LDS (the k-core sibling of truss_tail.h).
//
// The last levels of the k-core peel run on a few hundred hub vertices (|V| = 1M: 894 vertices left at level
// 40 of 46; |V| = 10M: 1018 at level 65 of 72), 10-20 sub-rounds per level, each a launch of the general
// engine with its ~10 dependent trips to memory.  When a level starts with at most kCoreTailV vertices left,
// the control block says "hand over" (done = 3): the live vertices are numbered, the adjacency among them
// becomes a bit matrix (their CSR rows are scanned once, against a bitmap of the live vertices), and one
// workgroup finishes the peel in LDS: live degrees as LDS counters, a frontier vertex's live neighbours are
// the set bits of row(v) & live mask, a decrement is an LDS atomic, a sub-round two __syncthreads().
// Coreness is unique, so any peel order inside a level gives the same values as the general engine.
// Nothing of the general engine's state is written before the tail has finished.
#pragma once

#include "peel_dev.h"

namespace komb {
namespace {

constexpr uint32_t kCoreTailV = 1024;             // vertices; bit rows of 16 words -> 128 KB of LDS
constexpr uint32_t kCoreTailWords = kCoreTailV / 64;

struct CoreTailBufs {
    unsigned long long *livebits;   // [ceil(nv/64)] bit per vertex: live at hand-over
    int32_t *vnum;                  // [nv] tail number + 1 of a live vertex (only read where livebits is set)
    uint32_t *cnt;                  // [4] 0: live vertices
    int32_t *vlist;                 // [kCoreTailV] original ids
    unsigned long long *rows;       // [kCoreTailV * kCoreTailWords]
};

// ---- setup 1: number the live vertices (any order: k-core needs no tie-break)
__global__ __launch_bounds__(kBlock) void k_ctail_mark(const int32_t *__restrict__ list, uint32_t n_in, const int32_t *__restrict__ core,
                                                       CoreTailBufs T)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_in; i += gridDim.x * kBlock) {
        const int32_t v = list ? list[i] : (int32_t)i;
        if (!marker_alive(core[v])) continue;
        const uint32_t id = atomicAdd(&T.cnt[0], 1u);
        if (id < kCoreTailV) {
            T.vlist[id] = v;
            T.vnum[v] = (int32_t)id + 1;
            atomicOr(&T.livebits[(uint32_t)v >> 6], 1ull << ((uint32_t)v & 63));
        }
    }
}

// ---- setup 2: the CSR rows of the live vertices against the live bitmap.  blockIdx.x = live vertex,
// blockIdx.y = which share of the row's 4096-slot pieces (a 134k-slot hub row is the critical path of the
// hand-over when one workgroup walks it alone); the pieces' bits are OR-ed into the zeroed global rows
constexpr uint32_t kCtailPiece = 4096;
__global__ __launch_bounds__(kBlock) void k_ctail_rows(const uint32_t *__restrict__ rowptr, const int32_t *__restrict__ col, CoreTailBufs T)
{
    __shared__ unsigned long long row[kCoreTailWords];
    const uint32_t n = T.cnt[0];
    if (n > kCoreTailV) return;
    for (uint32_t id = blockIdx.x; id < n; id += gridDim.x) {
        const int32_t v = T.vlist[id];
        const uint32_t b = rowptr[v], e = rowptr[v + 1];
        if (b + blockIdx.y * kCtailPiece >= e) continue;                 // uniform for the workgroup
        if (threadIdx.x < kCoreTailWords) row[threadIdx.x] = 0ull;
        __syncthreads();
        for (uint32_t p0 = b + blockIdx.y * kCtailPiece; p0 < e; p0 += gridDim.y * kCtailPiece) {
            const uint32_t p1 = min(e, p0 + kCtailPiece);
            for (uint32_t j = p0 + threadIdx.x; j < p1; j += kBlock) {
                const uint32_t w = (uint32_t)col[j];
                if ((T.livebits[w >> 6] >> (w & 63)) & 1ull) {
                    const uint32_t x = (uint32_t)T.vnum[w] - 1u;
                    atomicOr(&row[x >> 6], 1ull << (x & 63));
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < kCoreTailWords && row[threadIdx.x]) atomicOr(&T.rows[(size_t)id * kCoreTailWords + threadIdx.x], row[threadIdx.x]);
        __syncthreads();
    }
}

// ---- the tail peel
__global__ __launch_bounds__(1024) void k_core_tail(PeelCtrl *ctrl, CoreTailBufs T, const int32_t *__restrict__ degw, int32_t *__restrict__ core)
{
    __shared__ unsigned long long A[kCoreTailV * kCoreTailWords];     // 128 KB
    __shared__ int32_t deg[kCoreTailV];
    __shared__ uint16_t lvl[kCoreTailV];
    __shared__ uint16_t q[2][kCoreTailV];
    __shared__ unsigned long long M[kCoreTailWords];                    // live and not (yet) in a frontier
    __shared__ uint32_t sh_cnt[2];
    __shared__ int32_t sh_min;
    const uint32_t tid = threadIdx.x;
    const int lane = lane_id();
    const uint32_t n = T.cnt[0];
    if (n > kCoreTailV || n == 0) {
        // refused (cannot happen when the hand-over threshold is <= kCoreTailV): the general engine goes on
        if (tid == 0) { ctrl->tail_limit = 0; ctrl->done = 0; }
        return;
    }
    const uint32_t W = (n + 63) / 64;
    for (uint32_t w = tid; w < n * kCoreTailWords; w += 1024) A[w] = T.rows[w];
    if (tid < kCoreTailWords) M[tid] = 0ull;
    if (tid == 0) { sh_cnt[0] = sh_cnt[1] = 0u; }
    __syncthreads();
    if (tid < n) {
        deg[tid] = degw[T.vlist[tid]];
        atomicOr(&M[tid >> 6], 1ull << (tid & 63));
    }
    __syncthreads();

    int32_t L = ctrl->level;
    uint32_t alive = n, rounds = 0, levels = 0;
    int32_t max_level = ctrl->max_level;
    int sel = 0;
    int32_t state = 1;
    while (alive > 0) {
        // ---- SCAN: live vertices with degree <= L; the minimum live degree if there is none
        if (tid == 0) { sh_cnt[sel] = 0u; sh_cnt[sel ^ 1] = 0u; sh_min = 0x7FFFFFFF; }
        __syncthreads();
        int32_t lmin = 0x7FFFFFFF;
        if (tid < n && ((M[tid >> 6] >> (tid & 63)) & 1ull)) {
            const int32_t d = deg[tid];
            if (d <= L) q[sel][atomicAdd(&sh_cnt[sel], 1u)] = (uint16_t)tid;
            else lmin = d;
        }
        lmin = wave_min(lmin);
        if (lane == 0 && lmin != 0x7FFFFFFF) atomicMin(&sh_min, lmin);
        __syncthreads();
        uint32_t ncur = sh_cnt[sel];
        if (ncur == 0) {
            if (sh_min == 0x7FFFFFFF) { state = 2; break; }
            L = sh_min;
            __syncthreads();
            continue;
        }
        ++levels; max_level = L;
        while (ncur > 0) {
            // the frontier leaves the live mask first: its members are not decremented by each other
            for (uint32_t i = tid; i < ncur; i += 1024) {
                const uint32_t v = q[sel][i];
                lvl[v] = (uint16_t)L;
                atomicAnd(&M[v >> 6], ~(1ull << (v & 63)));
            }
            __syncthreads();
            // a 16-lane group per frontier vertex, a matrix word per lane
            const uint32_t grp = tid >> 4, gl = tid & 15;
            for (uint32_t i = grp; i < ncur; i += 64) {
                const uint32_t v = q[sel][i];
                unsigned long long bits = gl < W ? (A[v * kCoreTailWords + gl] & M[gl]) : 0ull;
                while (bits) {
                    const uint32_t x = gl * 64 + (uint32_t)__ffsll((long long)bits) - 1u;
                    bits &= bits - 1ull;
                    if (atomicSub(&deg[x], 1) == L + 1) q[sel ^ 1][atomicAdd(&sh_cnt[sel ^ 1], 1u)] = (uint16_t)x;
                }
            }
            __syncthreads();
            const uint32_t nnext = sh_cnt[sel ^ 1];
            if (tid == 0) sh_cnt[sel] = 0u;
            alive -= ncur;
            ++rounds;
            ncur = nnext;
            sel ^= 1;
            __syncthreads();
        }
        L += 1;
    }
    __syncthreads();
    if (state == 1 && tid < n) core[T.vlist[tid]] = (int32_t)lvl[tid];
    if (tid == 0) {
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
