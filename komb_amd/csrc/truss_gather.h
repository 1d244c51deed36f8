// truss_gather.h -- step 4 of the k-truss path (ktruss.hip): results in canonical (min,max)-lexicographic edge order with
// ORIGINAL vertex ids (what invmap gives at reference src/graph.cpp:531-532), without a search; and the graph moments of
// komb_stats.  Included by ktruss.hip only.
#pragma once

#include "peel_dev.h"

namespace komb {

namespace {

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
// Trussness after the peel: an edge the engine peeled carries the sub-round it went in (stamp), and rlevel[] says which level
// that sub-round worked at (PeelQueues::rlevel; stamp 0 = triangle-free = level 0); an edge a finish took over (local fixed
// point, LDS tail) still carries its alive marker and has its value in truss[] already.  One coalesced pass instead of a
// scattered 4-byte result store per peeled edge inside the peel (0.6 ms there at C3, 0.25 ms here).
__global__ __launch_bounds__(kBlock) void k_truss_resolve(const int32_t *__restrict__ stamp, const int32_t *__restrict__ rlevel,
                                                         int32_t *__restrict__ truss, int64_t m)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < m; e += (int64_t)gridDim.x * kBlock) {
        const int32_t s = stamp[e];
        if (!marker_alive(s)) truss[e] = rlevel[s] + 2;
    }
}

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

} // namespace komb
