// truss_gather.h -- step 4 of the k-truss path (ktruss.hip): results in canonical (min,max)-lexicographic edge order with
// ORIGINAL vertex ids (what invmap gives at reference src/graph.cpp:531-532); and the graph moments of komb_stats.  Included by ktruss.hip only.
#pragma once

#include "peel_dev.h"

namespace komb {

namespace {

// -------------------------------------------------------------- result gather
// The graph carries its canonical edge list (original ids, (min,max)-lexicographic: ctx->d_ceu / d_cev) and canon2e[k] = the
// internal edge id (oriented slot) of canonical edge k (graph_build.hip).  The values of edge k -- trussness, and the support
// the peel started from = the length of its incidence slice -- are read where the oriented slot put them: one gather per
// canonical edge, no search, no sort.  (Until round 4 the graph was processed in original ids and the two orders were tied
// together by rank arithmetic over the CSR plus a stable sort of the reversed oriented slots)
// Trussness after the peel: an edge the engine peeled carries the sub-round it went in (stamp), and rlevel[] says which level
// that sub-round worked at (PeelQueues::rlevel; stamp 0 = triangle-free = level 0); an edge a finish took over (local fixed
// point, LDS tail) still carries its alive marker and has its value in truss[] already.  One coalesced pass instead of a
// scattered 4-byte result store per peeled edge inside the peel (0.6 ms there at C3, 0.25 ms here).  The pass also packs
// the edge's initial support next to its trussness.
__global__ __launch_bounds__(kBlock) void k_truss_resolve(const int32_t *__restrict__ stamp, const int32_t *__restrict__ rlevel,
                                                         const int32_t *__restrict__ truss, const uint2 *__restrict__ off2,
                                                         int2 *__restrict__ res, int64_t m)
{
    // res[e] = (trussness, the support the peel started from): ONE 8-byte word per edge for the gather's one random read
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < m; e += (int64_t)gridDim.x * kBlock) {
        const int32_t s = stamp[e];
        res[e] = make_int2(marker_alive(s) ? truss[e] : rlevel[s] + 2, (int32_t)off2[e].y);
    }
}

// (k_lo, k_hi: the canonical edges this run materialises -- all of them, or the rank's slice of komb_truss_run_slice)
__global__ __launch_bounds__(kBlock) void k_gather_canonical(const uint32_t *__restrict__ canon2e, int64_t k_lo, int64_t k_hi, const int2 *__restrict__ res,
                                                             int32_t *__restrict__ tr_out, int32_t *__restrict__ sup_out)
{
    for (int64_t k = k_lo + (int64_t)blockIdx.x * kBlock + threadIdx.x; k < k_hi; k += (int64_t)gridDim.x * kBlock) {
        const int2 r = res[canon2e[k]];
        tr_out[k] = r.x;
        sup_out[k] = r.y;
    }
}

// induced subgraph: canonical edge k of the whole graph is kept when bit k of kbits is set; it is result number
// krank[k / 64] + (kept bits below k), and its values sit at the rank of its oriented slot among the kept oriented slots
__global__ __launch_bounds__(kBlock) void k_gather_sub(const uint32_t *__restrict__ canon2e, int64_t ne,
                                                       const unsigned long long *__restrict__ kbits, const uint32_t *__restrict__ krank,
                                                       const unsigned long long *__restrict__ obits, const uint32_t *__restrict__ wrank,
                                                       const int2 *__restrict__ res,
                                                       int32_t *__restrict__ tr_out, int32_t *__restrict__ sup_out)
{
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < ne; k += (int64_t)gridDim.x * kBlock) {
        const unsigned long long kw = kbits[k >> 6];
        if (!((kw >> (k & 63)) & 1ull)) continue;
        const uint32_t c = krank[k >> 6] + (uint32_t)__popcll(kw & ((1ull << (k & 63)) - 1ull));
        const uint32_t e = canon2e[k];
        const uint32_t o = wrank[e >> 6] + (uint32_t)__popcll(obits[e >> 6] & ((1ull << (e & 63u)) - 1ull));
        const int2 r = res[o];
        tr_out[c] = r.x;
        sup_out[c] = r.y;
    }
}

// vmask arrives by original vertex id; the kernels index by internal id
__global__ __launch_bounds__(kBlock) void k_mask_internal(const uint8_t *__restrict__ mask_o, const int32_t *__restrict__ i2o, int64_t nv,
                                                          uint8_t *__restrict__ mask_i)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kBlock) mask_i[i] = mask_o[i2o[i]];
}

// degrees inside an induced subgraph from its oriented slots (deg zeroed by the caller)
__global__ __launch_bounds__(kBlock) void k_sub_degree(const int32_t *__restrict__ osrc, const int32_t *__restrict__ ocol, int64_t m,
                                                       int32_t *__restrict__ deg)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < m; e += (int64_t)gridDim.x * kBlock) {
        atomicAdd(&deg[osrc[e]], 1);
        atomicAdd(&deg[ocol[e]], 1);
    }
}

// sum_v d(v)^2 and sum_e min(d(u),d(v)) for the roofline's algorithmic bytes
__global__ __launch_bounds__(kBlock) void k_graph_moments(const int32_t *__restrict__ deg, int64_t nv,
                                                          const int32_t *__restrict__ osrc, const int32_t *__restrict__ ocol,
                                                          int64_t m, const uint32_t *__restrict__ orow,
                                                          unsigned long long *out /*[5]: sum d^2, sum min, max d, (unused), sum d+ + d+*/)
{
    unsigned long long s2 = 0, smin = 0, mx = 0, so = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kBlock) {
        const unsigned long long d = (unsigned long long)deg[i];
        s2 += d * d;
        mx = d > mx ? d : mx;
    }
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < m; e += (int64_t)gridDim.x * kBlock)
    {
        const int32_t a = osrc[e], b = ocol[e];
        smin += (unsigned long long)min(deg[a], deg[b]);
        so += (unsigned long long)(orow[a + 1] - orow[a]) + (unsigned long long)(orow[b + 1] - orow[b]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s2 += __shfl_xor(s2, o); smin += __shfl_xor(smin, o); so += __shfl_xor(so, o);
        const unsigned long long t = __shfl_xor(mx, o); mx = t > mx ? t : mx;
    }
    if (lane_id() == 0) { atomicAdd(&out[0], s2); atomicAdd(&out[1], smin); atomicMax(&out[2], mx); atomicAdd(&out[4], so); }
}

} // namespace

} // namespace komb
