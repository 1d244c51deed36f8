// truss_tri.h -- the probe enumeration of rounds 1-3 (igraph_trussness's igraph_list_triangles + per-edge support count,
// reference src/graph.cpp:508), kept for the exact count-scan-fill TWO-PASS build of the incidence index:
// the fallback of ktruss.hip when the record stream of the wedge enumeration (truss_wedge.h) does not fit in memory or
// runs out, and a second, independent enumeration the tests cross-check the first against.  Also the definitions the two
// enumerations share (record stream, dense own-role blocks).  Included by ktruss.hip only.
#pragma once

#include "truss_line.h"

namespace komb {

namespace {

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
//   TRI_SINGLE writes each edge's incidence pairs into its EXACT slice
//              [off[x], off[x+1]) (after TRI_COUNT + scan): own-role entries
//              from the front (LDS cursor), third-role entries from the back
//              (a global cursor counted down): the two ends meet precisely.
// Tasks whose rows exceed the LDS budget fall back to global binary search and
// global atomics for all three roles.
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
#ifndef KOMB_TRI_R
#define KOMB_TRI_R 4
#endif
constexpr int kTriR = KOMB_TRI_R;              // consecutive elements of one row N+(b) a lane probes per trip (16-byte loads; 4 or 8)
static_assert(kTriR == 4 || kTriR == 8, "chunks are loaded as 16-byte vectors");
constexpr int kTriBuf = 128;                   // parked triangles per wave on the unstaged path (handled once >= 64 are waiting)
#ifndef KOMB_TRI_REC
#define KOMB_TRI_REC 384
#endif
constexpr int kTriRec = KOMB_TRI_REC;           // triangle records a staged task keeps until it is done (own-role entries, 8 bytes each)
constexpr int kTriCand = KOMB_TRI_CAND;        // parked lookup candidates per wave (searched once >= 64 are waiting)
#ifndef KOMB_TRI_SIGW
#define KOMB_TRI_SIGW 8
#endif
constexpr int kTriSigW = KOMB_TRI_SIGW;          // 32-bit words of a source row's Bloom signature (a power of two)
constexpr int kTriSigShift = 32 - 5 - (kTriSigW == 2 ? 1 : kTriSigW == 4 ? 2 : kTriSigW == 8 ? 3 : 4);

enum : int { TRI_COUNT = 0, TRI_SINGLE = 2 };

// ---- shared with the wedge enumeration (truss_wedge.h)
// DENSE own-role blocks: a staged task keeps one 8-byte record per triangle in LDS and, when it is done, writes the own-role
// entries of all its edges as ONE compact block of `dense` (claimed from a cursor in chunks), edge after edge: ownoff[e] =
// where edge e's entries start.  176 M scattered 8-byte stores (one HBM line each) become a coalesced stream.  A task
// without a block (a row too long to stage, a region that has run out) says ownoff[e] = kOwnSpill: its entries are records.
constexpr unsigned long long kOwnSpill = ~0ull;
constexpr uint32_t kOwnChunk = 4096;                // entries a wavefront claims from dense_cursor at a time (one atomic per ~15 tasks)

// STREAM: every incidence entry that does not go into a dense own-role block -- the third-role entry of every triangle, and
// all three entries of a triangle whose task has no block -- is appended as a record (key = the edge the entry belongs to,
// value = the other two edges) to ONE stream, 64 records per store instruction, no atomic and no scattered store per
// triangle.  A wavefront claims kRecChunk positions of the stream at a time; what it leaves unused gets a sentinel key
// (larger than every edge id).  The host then sorts the records by bin (truss_index.h) and merges them with the dense
// blocks.  A claim beyond `cap` writes nothing: the host sees the cursor pass the capacity and falls back to the exact
// two-pass build.
constexpr uint32_t kRecChunk = 1024;
struct TriStream {
    uint32_t *key;                       // [cap]
    int2 *val;                           // [cap]
    unsigned long long *cursor;          // positions claimed so far
    unsigned long long cap;
    uint32_t sentinel;                   // key of an unused position: above every edge id ...
    uint32_t spread_mask;                // ... with the position's low bits in the key's bin field, so that they spread over the bins (truss_index.h)
    int spread_shift;
};

template <int MODE>
__global__ __launch_bounds__(kBlock, KOMB_TRI_EU) void k_triangles(const uint32_t *__restrict__ orow, const int32_t *__restrict__ ocol,
                                                      int64_t nv, int64_t task_lo, int64_t task_hi,
                                                      uint32_t *own, uint32_t *other_or_cursor,
                                                      const uint32_t *__restrict__ off, int2 *__restrict__ inc, int ablate, int tv)
{
    // other_or_cursor: TRI_COUNT: the third-role counters; TRI_SINGLE: every edge's back cursor, which starts at off[x+1]-1,
    // the last position of x's slice, and is counted DOWN (k_back_cursors): the returning atomic is the third-role write
    // position itself
    // tv: consecutive source vertices per task (<= kTriV).  Fewer than kTriV when the graph has few vertices for its work
    // (a 20 000-vertex graph with 4 M edges is 1 250 tasks of 16 vertices: not even one per SIMD)
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
                    } else {
                        const uint32_t pe = off[e] + atomicAdd(&own[e], 1u);
                        const uint32_t pi = off[i] + atomicAdd(&own[i], 1u);
                        const uint32_t pj = atomicSub(&other_or_cursor[jj], 1u);
                        inc[pe] = make_int2((int)i, (int)jj);
                        inc[pi] = make_int2((int)e, (int)jj);
                        inc[pj] = make_int2((int)e, (int)i);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            n_tri = 0;
        };

        // ---- staged sub-range.  A triangle's own-role cursors are LDS atomics taken when it is found; the rest waits in
        // the record buffer: record = (e_rel | i_rel << 8 | cursor_e << 16 | cursor_i << 24, j).  The records are worked off
        // densely, 64 per pass (`drain`): the third-role entry (one returning global atomic + one store) and the two
        // own-role stores.
        uint32_t n_rec = 0;                                     // wave-uniform: records waiting
        auto drain = [&]() {
            __builtin_amdgcn_wave_barrier();
            if (MODE == TRI_SINGLE) for (uint32_t b0 = 0; b0 < n_rec; b0 += kWave) {
                const uint32_t x = b0 + (uint32_t)lane;
                if (x < n_rec) {
                    const uint2 rc = s_rec[x];
                    const uint32_t e = S0 + (rc.x & 0xFFu), i = S0 + ((rc.x >> 8) & 0xFFu), jj = rc.y;
                    if (!(ablate & 32)) inc[atomicSub(&other_or_cursor[jj], 1u)] = make_int2((int)e, (int)i);
                    if (!(ablate & 16)) {
                        inc[off[e] + ((rc.x >> 16) & 0xFFu)] = make_int2((int)i, (int)jj);
                        inc[off[i] + (rc.x >> 24)] = make_int2((int)e, (int)jj);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            n_rec = 0;
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
                    if (n_rec >= (uint32_t)kWave) drain();                    // 64 or more are waiting: one dense pass over all of them
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
#pragma unroll
                        for (int k4 = 0; k4 < kTriR; k4 += 4) {
                            if (k4 == 0 || (uint32_t)k4 < nin) {
                                const Int4U q = *reinterpret_cast<const Int4U *>(ocol + j0 + k4);
                                wv[k4] = q.x; wv[k4 + 1] = q.y; wv[k4 + 2] = q.z; wv[k4 + 3] = q.w;
                            } else { wv[k4] = 0; wv[k4 + 1] = 0; wv[k4 + 2] = 0; wv[k4 + 3] = 0; }
                        }
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
        if (!staged) { flush_tris(); continue; }
        search_cands();
        if (MODE == TRI_SINGLE) drain();
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = (uint32_t)lane; k < E; k += kWave) own[S0 + k] = s_cnt[k];
      }   // sub-ranges
    }
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
} // namespace

} // namespace komb
