// truss_gather.h -- step 4 of the k-truss path (ktruss.hip): results in canonical (min,max)-lexicographic edge order with
// ORIGINAL vertex ids (what invmap gives at reference src/graph.cpp:531-532).  Included by ktruss.hip only.
#pragma once

#include "peel_dev.h"

namespace komb {

namespace {

// -------------------------------------------------------------- result gather
// The preparation carries the canonical edge list (original ids, (min,max)-lexicographic: TrussPrep::ceu / cev) and canon2e[k] =
// the internal edge id (oriented slot) of canonical edge k (truss_prep.hip).  The values of edge k -- trussness, and the support
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

} // namespace

} // namespace komb
