// truss_gather.h -- step 4 of the k-truss path (ktruss.hip): results in canonical (min,max)-lexicographic edge order with
// ORIGINAL vertex ids (what invmap gives at reference src/graph.cpp:531-532).  Included by ktruss.hip only.
#pragma once

#include "peel_dev.h"

namespace komb {

namespace {

// -------------------------------------------------------------- results in canonical order
// The preparation carries e2k[e] = the canonical id of internal edge e (oriented slot) -- written where the oriented rows are
// written, as a stream (truss_prep.hip); the canonical edge list itself (the endpoints) is made when a fetch asks for it.
// The values of an edge go where its canonical id says: ONE pass, one scattered 4-byte store per edge, no search, no sort.
// (Round 4 kept the inverse map, canon2e, and gathered: the map cost the preparation 1.4 ms of scattered stores, the gather
// 1.3 ms of random 8-byte reads after a 0.4 ms packing pass; until round 4 the graph was processed in original ids and the two
// orders were tied together by rank arithmetic over the CSR plus a stable sort of the reversed oriented slots.)
// Trussness after the peel: an edge the engine peeled carries the sub-round it went in (stamp), and rlevel[] says which level
// that sub-round worked at (PeelQueues::rlevel; stamp 0 = triangle-free = level 0); an edge a finish took over (local fixed
// point, LDS tail) still carries its alive marker and has its value in truss[] already.  No result store per peeled edge inside
// the peel.  Every edge's initial support is the length of its incidence slice: the slice table stays with the result, and the
// support vector in canonical order -- an extra of this library (igraph_trussness has no such output) -- is made from it when
// komb_truss_fetch_support asks (k_scatter_len).
// (k_lo, k_hi: the canonical edges this run materialises -- all of them, or the rank's slice of komb_truss_run_slice)
__global__ __launch_bounds__(kBlock) void k_truss_results(const int32_t *__restrict__ stamp, const int32_t *__restrict__ rlevel,
                                                         const int32_t *__restrict__ truss,
                                                         const uint32_t *__restrict__ e2k, int64_t m, uint32_t k_lo, uint32_t k_hi,
                                                         int32_t *__restrict__ tr_out)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < m; e += (int64_t)gridDim.x * kBlock) {
        const int32_t s = stamp[e];
        const int32_t t = marker_alive(s) ? truss[e] : rlevel[s] + 2;
        const uint32_t k = e2k[e];
        if (k - k_lo < k_hi - k_lo) tr_out[k] = t;
    }
}

// the initial supports in canonical order: the length of every edge's index slice, at its canonical id
__global__ __launch_bounds__(kBlock) void k_scatter_len(const uint2 *__restrict__ off2, const uint32_t *__restrict__ e2k, int64_t m,
                                                       uint32_t k_lo, uint32_t k_hi, int32_t *__restrict__ out)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < m; e += (int64_t)gridDim.x * kBlock) {
        const uint32_t k = e2k[e];
        if (k - k_lo < k_hi - k_lo) out[k] = (int32_t)off2[e].y;
    }
}

} // namespace

} // namespace komb
