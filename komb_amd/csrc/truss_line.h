// truss_line.h -- what the k-truss preparation (truss_prep.hip) and the triangle enumeration (truss_wedge.h, truss_tri.h)
// share: the 64-byte line that describes a vertex' oriented row, and the enumeration's task table.
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

struct __attribute__((packed, aligned(4))) Int4U { int32_t x, y, z, w; };      // 16 bytes at a 4-byte aligned address
struct __attribute__((packed, aligned(4))) UInt2U { uint32_t x, y; };

#ifndef KOMB_TRI_CAP
#define KOMB_TRI_CAP 256
#endif
constexpr int kTriCap = KOMB_TRI_CAP;           // oriented slots a wavefront of the enumeration stages in LDS
constexpr int kTriWaves = kBlock / kWave;

// ---- one 64-byte line per vertex: { start of its oriented row, length, 6 pivots, 4 x 64-bit blocked Bloom signature }
constexpr int kLineWords = 16;
constexpr int kSigBlocks = 4;
constexpr int kPivots = 6;                      // the row's elements at positions seg, 2 seg, ... 6 seg, seg = ceil(length / 7)

__device__ __forceinline__ void sig_slot(int32_t c, uint32_t &blk, unsigned long long &mask)
{
    const uint32_t h = (uint32_t)c * 0x9E3779B1u;
    blk = h >> 30;                                                   // 0 .. 3
    mask = (1ull << (h & 63u)) | (1ull << ((h >> 6) & 63u));
}

// position of c in the ascending row a line describes (0xFFFFFFFF: absent): the pivots name the segment; a segment of up to
// 8 elements is two 16-byte loads issued together; longer ones (rows beyond 56 slots) are halved in global memory first.
// `ocol` must be readable 8 elements past the row's end (the oriented CSR is padded).
__device__ __forceinline__ uint32_t line_find(const uint4 &l0, const uint4 &l1, int32_t c, const int32_t *__restrict__ ocol)
{
    const uint32_t start = l0.x, len = l0.y, seg = (len + (uint32_t)kPivots) / (uint32_t)(kPivots + 1);
    const int32_t p0 = (int32_t)l0.z, p1 = (int32_t)l0.w, p2 = (int32_t)l1.x, p3 = (int32_t)l1.y, p4 = (int32_t)l1.z, p5 = (int32_t)l1.w;
    uint32_t sidx;                                                   // the number of pivots <= c (unused pivots are INT32_MAX)
    if (p3 <= c) sidx = p5 <= c ? 6u : (p4 <= c ? 5u : 4u);
    else sidx = p1 <= c ? (p2 <= c ? 3u : 2u) : (p0 <= c ? 1u : 0u);
    const uint32_t lo = sidx * seg;
    uint32_t l = start + lo, n = lo < len ? min(seg, len - lo) : 0u;
    while (n > 8u) {
        const uint32_t half = n >> 1;
        if (ocol[l + half] <= c) { l += half; n -= half; } else n = half;
    }
    if (!n) return 0xFFFFFFFFu;
    const Int4U q0 = *reinterpret_cast<const Int4U *>(ocol + l);
    Int4U q1 = {0, 0, 0, 0};
    if (n > 4u) q1 = *reinterpret_cast<const Int4U *>(ocol + l + 4);
    const uint32_t pos = q0.x == c ? 0u : (n > 1u && q0.y == c) ? 1u : (n > 2u && q0.z == c) ? 2u : (n > 3u && q0.w == c) ? 3u :
                         (n > 4u && q1.x == c) ? 4u : (n > 5u && q1.y == c) ? 5u : (n > 6u && q1.z == c) ? 6u : (n > 7u && q1.w == c) ? 7u : 0xFFFFFFFFu;
    return pos == 0xFFFFFFFFu ? pos : l + pos;
}

// ---- the enumeration's tasks.  A task is a run of consecutive source vertices -- or, for a row too long to stage, one of
// several PARTS of that row's owned edges.  Vertex ids are (degree,id) ranks, so heavy rows are neighbours: a fixed number
// of vertices per task would hand one wavefront sixteen hub rows in a row (a K_250 inside a sparse graph: 12.7 ms of an
// otherwise 1 ms enumeration on 16 wavefronts).  Rule: a row of kHeavyRow slots or more is a task of its own; lighter
// vertices are grouped up to `group` (<= kWedgeV) of them, never across a multiple of `group`; a row beyond the LDS budget
// (unstaged) is cut into parts of about kPartPairs wedges, its 64-edge batches dealt round-robin to the parts.
// Descriptor: x = first vertex, y = vertices | part << 6 | parts << 19.
#ifndef KOMB_WEDGE_V
#define KOMB_WEDGE_V 32
#endif
constexpr int kWedgeV = KOMB_WEDGE_V;           // light vertices per task at most (<= 63: lane l holds orow[v0 + l]); same-box A/B at
                                                // |E| = 100M: 8: 6.77 ms, 16: 5.76-6.0, 32: 5.46-5.69, 48: 5.58, 63: 5.58
static_assert(kWedgeV >= 1 && kWedgeV <= 63, "a task's row pointers live in the lanes of one wavefront");
#ifndef KOMB_HEAVY_ROW
#define KOMB_HEAVY_ROW 32
#endif
constexpr uint32_t kHeavyRow = KOMB_HEAVY_ROW;
constexpr unsigned long long kPartPairs = 16384;
constexpr uint32_t kMaxParts = 8191;
#ifndef KOMB_WEDGE_EU
#define KOMB_WEDGE_EU 4
#endif

__device__ __forceinline__ bool task_starts(const uint32_t *__restrict__ orow, int64_t v, int group)
{
    if (v % group == 0) return true;
    if (orow[v + 1] - orow[v] >= kHeavyRow) return true;
    return orow[v] - orow[v - 1] >= kHeavyRow;                     // (v > 0 here)
}

__device__ __forceinline__ uint32_t task_parts(unsigned long long d)       // tasks a row of d slots that starts a task is cut into
{
    if (d <= (unsigned long long)kTriCap) return 1u;
    const unsigned long long parts = (d * (d - 1ull) / 2ull + kPartPairs - 1ull) / kPartPairs;
    const unsigned long long batches = (d + kWave - 1ull) / kWave;
    return (uint32_t)min(min(parts, batches), (unsigned long long)kMaxParts);
}

} // namespace

} // namespace komb
