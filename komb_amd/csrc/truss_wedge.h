// truss_wedge.h -- step 2a of the k-truss path (ktruss.hip): the triangle enumeration by WEDGES (igraph_trussness's
// igraph_list_triangles + per-edge support count, reference src/graph.cpp:508).
//
// k_triangles (truss_tri.h) finds the triangles {a,b,c}, a -> b -> c in (degree,id) order, from the oriented edge e = (a->b)
// by streaming ALL of N+(b) past the LDS-staged row of a: sum over the edges of d+(b) probes -- 2.08 G at |E| = 100 M, each a
// hash, an LDS signature read and a ballot -- plus the row fetch itself for every edge.  The work that can produce a
// triangle is much smaller: a triangle of edge (a->b) needs a c in N+(a) ABOVE b (rows are ascending, and ids are
// (degree,id) ranks), i.e. one of the slots of row a behind the edge's own.  Those (edge, later slot) pairs are the WEDGES
// a -> {b, c}: sum over the vertices of C(d+(a), 2) = 0.59 G at |E| = 100 M, 3.5 x fewer than the probes, and the last edge
// of every row (10 % of the edges) has none.
//
// This kernel turns the test around: the candidates c come out of the staged row of a, and what is fetched per edge is
// ONE 64-byte line of b -- start and length of N+(b), six PIVOTS (the elements that cut the row into seven equal segments)
// and a 256-bit blocked Bloom signature of its elements (the preparation writes the lines with the oriented rows,
// truss_prep.hip; truss_line.h).  A candidate that passes the signature (the true hits, 16 % at |E| = 100 M, plus a few % false positives) is parked
// and looked up in N+(b) itself, 64 candidates at a time: the pivots (in LDS) name its segment, and a segment of up to 8
// elements is two 16-byte loads issued together -- ONE trip to memory per survivor, where a binary search over the row
// made five or six dependent ones (measured: 4.6 of the kernel's 8.0 ms); only then is a row of b touched at all.  What happens
// to a found triangle -- LDS cursors of the two owned edges, the record for the third edge, the dense own-role block of
// the task -- is the scheme of truss_tri.h, unchanged.
//
// Included by ktruss.hip only, after truss_tri.h.
#pragma once

#include "truss_tri.h"

namespace komb {

namespace {

constexpr int kLdsLine = 10;                    // words of a line kept in LDS: 4 signature blocks, start, length
constexpr int kWCand = 128;                     // parked candidates per wavefront
constexpr uint32_t kScratchRec = (uint32_t)kTriCap * (kTriCap - 1) / 2 + 128;     // records of a staged sub-range at most (+ one LDS buffer's slack)

// MODE: TRI_COUNT (supports only: own[] by plain stores, other[] by atomics) or TRI_SINGLE (the record-stream build:
// own-role entries as dense per-task blocks, everything else as records; truss_tri.h)
template <int MODE>
__global__ __launch_bounds__(kBlock, KOMB_WEDGE_EU) void k_wedges(const uint32_t *__restrict__ orow, const int32_t *__restrict__ ocol,
                                                                   const uint4 *__restrict__ line, const uint2 *__restrict__ tasks,
                                                                   int64_t task_lo, int64_t task_hi,
                                                                   uint32_t *own, uint32_t *other,
                                                                   int2 *__restrict__ dense, unsigned long long *dense_cursor, unsigned long long dense_cap,
                                                                   unsigned long long *__restrict__ ownoff, TriStream ts, uint2 *__restrict__ scratch, int ablate)
{
    // scratch (TRI_SINGLE): kScratchRec record slots per wavefront of the grid.  A staged sub-range keeps one 8-byte record per
    // triangle until it is done (its dense own-role block needs every cursor first); what the LDS buffer cannot hold moves
    // here -- wave-private, written and read back as a stream -- instead of giving the block up.  A staged sub-range has at most
    // C(kTriCap, 2) triangles, which fits: only rows too long to stage send their own-role entries through the record stream.
    // ablate (debug builds, KOMB_TRI_ABLATE; breaks results on purpose): 1 = survivors are dropped (no look-up), 2 = no candidate tests, 4 = no line loads, 8 = look-ups find nothing
    constexpr bool STREAM = MODE == TRI_SINGLE;
    __shared__ int32_t sh_col[kTriWaves][kTriCap];
    __shared__ uint32_t sh_cnt[kTriWaves][kTriCap];
    __shared__ uint32_t sh_orow[kTriWaves][kWedgeV + 1];
    __shared__ uint32_t sh_pref[kTriWaves][kWave];
    __shared__ uint32_t sh_rend[kTriWaves][kWave];
    __shared__ uint2 sh_rec[kTriWaves][kTriRec];
    __shared__ uint2 sh_cand[kTriWaves][kWCand];
    // of every edge's line LDS keeps what the candidate tests read -- the signature -- and start / length; the six pivots stay in
    // the registers of the edge's lane and reach a survivor's lane by shuffle (8 KB of LDS less per workgroup: 4 instead of 3
    // workgroups per CU)
    __shared__ __attribute__((aligned(16))) uint32_t sh_line[kTriWaves][kWave][kLdsLine];
    __shared__ uint8_t sh_rid[kTriWaves][kTriCap];
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    int32_t *s_col = sh_col[w];
    uint32_t *s_cnt = sh_cnt[w], *s_orow = sh_orow[w], *s_pref = sh_pref[w], *s_rend = sh_rend[w];
    uint2 *s_rec = sh_rec[w];
    uint3 *s_tri = reinterpret_cast<uint3 *>(sh_rec[w]);
    uint2 *s_cand = sh_cand[w];
    uint32_t (*s_line)[kLdsLine] = sh_line[w];
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

    // the heaviest tasks (the last ones: hub rows) go first
    for (int64_t task = task_hi - 1 - gw; task >= task_lo; task -= nw) {
      const uint2 td = tasks[task];
      const int64_t v0t = (int64_t)td.x;
      const int nvt_all = (int)(td.y & 63u);
      const uint32_t part = (td.y >> 6) & kMaxParts, nparts = td.y >> 19;       // (nparts > 1: one unstaged row, this task owns its batches part, part + nparts, ...)
      const uint32_t myrow = (lane <= nvt_all) ? orow[v0t + lane] : 0u;       // lane l holds orow[v0t + l]
      // A task whose rows exceed the LDS budget is cut into sub-ranges of consecutive vertices that fit; a single row
      // longer than the budget runs unstaged (its candidates are read from global memory, all three roles are records).
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
        if (staged) {
            for (uint32_t k = (uint32_t)lane; k < E; k += kWave) {
                s_col[k] = ocol[S0 + k]; s_cnt[k] = 0u;
                int lo = 0, hi = nvt - 1;                     // source row of the slot: last idx with s_orow[idx] <= S0 + k
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_orow[mid] <= S0 + k) lo = mid; else hi = mid - 1; }
                s_rid[k] = (uint8_t)lo;
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- unstaged sub-range: triangles are parked and handled 64 at a time, all three roles through records / global counters
        uint32_t n_tri = 0;                                     // parked triangles (wave-uniform)
        auto flush_tris = [&]() {
            __builtin_amdgcn_wave_barrier();
            for (uint32_t b0 = 0; b0 < n_tri; b0 += kWave) {
                const uint32_t x = b0 + (uint32_t)lane;
                const bool has = x < n_tri;
                const uint3 tr = has ? s_tri[x] : make_uint3(0u, 0u, 0u);
                const uint32_t e = S0 + tr.x, i = S0 + tr.y, jj = tr.z;
                if (MODE == TRI_COUNT) {
                    if (has) { atomicAdd(&other[e], 1u); atomicAdd(&other[i], 1u); atomicAdd(&other[jj], 1u); }
                } else {
                    rec_append(has, e, make_int2((int)i, (int)jj));
                    rec_append(has, i, make_int2((int)e, (int)jj));
                    rec_append(has, jj, make_int2((int)e, (int)i));
                }
            }
            __builtin_amdgcn_wave_barrier();
            n_tri = 0;
        };

        // ---- staged sub-range: record = (e_rel | i_rel << 8 | cursor_e << 16 | cursor_i << 24, j); see truss_tri.h
        uint32_t n_rec = 0, n_done = 0;                         // wave-uniform: records, records whose third role is written
        uint32_t n_scr = 0;                                     // wave-uniform: records moved to the wavefront's scratch
        uint2 *scr = scratch ? scratch + (size_t)gw * kScratchRec : nullptr;
        bool spilled = false;                                   // wave-uniform: the own-role entries are records too
        auto drain = [&](uint32_t lo, uint32_t hi, bool third, bool own_role) {
            __builtin_amdgcn_wave_barrier();
            if (STREAM) for (uint32_t b0 = lo; b0 < hi; b0 += kWave) {
                const uint32_t x = b0 + (uint32_t)lane;
                const bool has = x < hi;
                const uint2 rc = has ? s_rec[x] : make_uint2(0u, 0u);
                const uint32_t e = S0 + (rc.x & 0xFFu), i = S0 + ((rc.x >> 8) & 0xFFu), jj = rc.y;
                if (third) rec_append(has, jj, make_int2((int)e, (int)i));
                if (own_role) {
                    rec_append(has, e, make_int2((int)i, (int)jj));
                    rec_append(has, i, make_int2((int)e, (int)jj));
                }
            }
            __builtin_amdgcn_wave_barrier();
        };

        // candidates that pass the signature are parked as (lane of their edge in the batch | slot of c in row a << 6, c)
        // and looked up in N+(b) densely, 64 at a time; p0 = the batch's first slot
        uint32_t n_cand = 0;                                    // wave-uniform
        int32_t pv0 = 0, pv1 = 0, pv2 = 0, pv3 = 0, pv4 = 0, pv5 = 0;      // the pivots of this lane's edge of the current batch
        auto search_cands = [&](uint32_t p0) {
            __builtin_amdgcn_wave_barrier();
            if (ablate & 1) n_cand = 0;
            for (uint32_t b0 = 0; b0 < n_cand; b0 += kWave) {
                const uint32_t x = b0 + (uint32_t)lane;
                uint32_t t = 0, i_rel = 0, l = 0, n = 0;
                int32_t c = 0;
                const bool have = x < n_cand;
                if (have) {
                    const uint2 cd = s_cand[x];
                    t = cd.x & 63u; i_rel = cd.x >> 6; c = (int32_t)cd.y;
                }
                // the pivots of the candidate's edge, from the registers of that edge's lane (all lanes take part in the shuffles)
                const int32_t p0v = __shfl(pv0, (int)t), p1v = __shfl(pv1, (int)t), p2v = __shfl(pv2, (int)t);
                const int32_t p3v = __shfl(pv3, (int)t), p4v = __shfl(pv4, (int)t), p5v = __shfl(pv5, (int)t);
                if (have) {
                    const uint32_t *ln = s_line[t];
                    l = ln[8];
                    const uint32_t len = ln[9], seg = (len + (uint32_t)kPivots) / (uint32_t)(kPivots + 1);
                    // segment of c: the number of pivots <= c (the pivots are ascending; unused ones are INT32_MAX)
                    uint32_t sidx;
                    if (p3v <= c) sidx = p5v <= c ? 6u : (p4v <= c ? 5u : 4u);
                    else sidx = p1v <= c ? (p2v <= c ? 3u : 2u) : (p0v <= c ? 1u : 0u);
                    const uint32_t lo = sidx * seg;
                    l += lo;
                    n = lo < len ? min(seg, len - lo) : 0u;
                }
                // segments longer than 8 elements (rows longer than 56) are halved in global memory first
                while (__ballot(n > 8u)) {
                    const bool on = n > 8u;
                    const uint32_t half = n >> 1;
                    const int32_t pvv = on ? ocol[l + half] : 0;
                    const bool go = on && pvv <= c;                           // c, if present, is at or behind l + half
                    l = go ? l + half : l;
                    n = on ? (go ? n - half : half) : n;
                }
                // the segment: up to 8 consecutive elements, two 16-byte loads issued together
                uint32_t pos = 0xFFFFFFFFu;
                if (n) {
                    const Int4U q0 = *reinterpret_cast<const Int4U *>(ocol + l);
                    Int4U q1 = {0, 0, 0, 0};
                    if (n > 4u) q1 = *reinterpret_cast<const Int4U *>(ocol + l + 4);
                    pos = q0.x == c ? 0u : (n > 1u && q0.y == c) ? 1u : (n > 2u && q0.z == c) ? 2u : (n > 3u && q0.w == c) ? 3u :
                          (n > 4u && q1.x == c) ? 4u : (n > 5u && q1.y == c) ? 5u : (n > 6u && q1.z == c) ? 6u : (n > 7u && q1.w == c) ? 7u : 0xFFFFFFFFu;
                }
                const bool hit = pos != 0xFFFFFFFFu && !((ablate & 8) && c != -1);          // (ablate 8: the look-up's loads happen, no triangle is reported)
                l += hit ? pos : 0u;
                const uint64_t hm = __ballot(hit);
                if (!hm) continue;
                const uint32_t e_rel = p0 + t;
                if (!staged) {
                    if (hit) s_tri[n_tri + (uint32_t)__popcll(hm & lanemask_lt())] = make_uint3(e_rel, i_rel, l);
                    n_tri += (uint32_t)__popcll(hm);
                    if (n_tri >= (uint32_t)kTriBuf - kWave) flush_tris();
                    continue;
                }
                if (hit) {
                    const uint32_t ce = atomicAdd(&s_cnt[e_rel], 1u), ci = atomicAdd(&s_cnt[i_rel], 1u);
                    if (MODE == TRI_COUNT) atomicAdd(&other[l], 1u);
                    else s_rec[n_rec + (uint32_t)__popcll(hm & lanemask_lt())] = make_uint2(e_rel | (i_rel << 8) | (ce << 16) | (ci << 24), l);
                }
                if (MODE == TRI_SINGLE) {
                    n_rec += (uint32_t)__popcll(hm);
                    if (n_rec - n_done >= (uint32_t)kWave) {                 // 64 or more are waiting: one dense pass over all of them
                        drain(n_done, n_rec, true, spilled);
                        n_done = n_rec;
                    }
                    if (n_rec > (uint32_t)kTriRec - kWave) {
                        // the buffer is full: the sub-range gives up its dense block -- everything kept so far, and what follows, become records
                        if (!spilled && scr && n_scr + n_rec <= kScratchRec) {
                            drain(n_done, n_rec, true, false);                    // their third-role records go out now
                            for (uint32_t x = (uint32_t)lane; x < n_rec; x += kWave) scr[n_scr + x] = s_rec[x];
                            n_scr += n_rec;
                            if (lane == 0) atomicAdd(dense_cursor + 1, 1ull);     // statistics
                        } else if (!spilled) {
                            drain(n_done, n_rec, true, false);
                            drain(0, n_rec, false, true);
                            spilled = true;
                        } else drain(n_done, n_rec, true, true);
                        n_rec = 0; n_done = 0;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            n_cand = 0;
        };

        // The line of the NEXT batch's edge is requested before this batch's candidates are tested: its trip to memory runs
        // beside the tests (LDS work) instead of before them.  16 registers per lane.
        uint4 nq0 = make_uint4(0u, 0u, 0u, 0u), nq1 = nq0, nq2 = nq0, nq3 = nq0;
        uint32_t n_ncand = 0, n_rend = 0;
        auto request_lines = [&](uint32_t p0) {
            const uint32_t rel = p0 + (uint32_t)lane;
            n_ncand = 0; n_rend = 0;
            if (rel < E) {
                n_rend = staged ? s_orow[(uint32_t)s_rid[rel] + 1u] - S0 : E;
                n_ncand = n_rend - rel - 1u;
            }
            if (ablate & 4) n_ncand = 0;
            if (n_ncand) {
                const int32_t b = staged ? s_col[rel] : ocol[S0 + rel];
                const uint4 *L = line + 4 * (int64_t)b;
                nq0 = L[0]; nq1 = L[1]; nq2 = L[2]; nq3 = L[3];
            }
        };
        if (part * (uint32_t)kWave < E) request_lines(part * (uint32_t)kWave);
        for (uint32_t p0 = part * (uint32_t)kWave; p0 < E; p0 += nparts * (uint32_t)kWave) {
            // lane <-> owned edge e = S0 + p0 + lane = (a -> b); its candidates are the slots of row a behind it
            uint32_t ncand = n_ncand;
            const uint32_t rend = n_rend;
            const uint4 q0 = nq0, q1 = nq1, q2 = nq2, q3 = nq3;
            if (p0 + nparts * (uint32_t)kWave < E) request_lines(p0 + nparts * (uint32_t)kWave);
            if (ncand) {
                uint32_t *d = s_line[lane];
                d[0] = q2.x; d[1] = q2.y; d[2] = q2.z; d[3] = q2.w; d[4] = q3.x; d[5] = q3.y; d[6] = q3.z; d[7] = q3.w;
                d[8] = q0.x; d[9] = q0.y;
                pv0 = (int32_t)q0.z; pv1 = (int32_t)q0.w; pv2 = (int32_t)q1.x; pv3 = (int32_t)q1.y; pv4 = (int32_t)q1.z; pv5 = (int32_t)q1.w;
                if (q0.y == 0u) ncand = 0;                      // b has no out-neighbours
            }
            // the candidates of the 64 edges are cut into chunks of 4 consecutive slots of ONE row and the chunks are flattened
            // over the lanes (one owner search per 4 candidates)
            const uint32_t incl = wave_incl_scan((ncand + 3u) >> 2);
            const uint32_t total = (uint32_t)__shfl((int)incl, kWave - 1);
            __builtin_amdgcn_wave_barrier();
            s_pref[lane] = incl; s_rend[lane] = rend;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t it0 = 0; it0 < ((ablate & 2) ? 0u : total); it0 += kWave) {
                const uint32_t it = it0 + (uint32_t)lane;
                const bool live = it < total;
                int t = 0;                                    // owner: smallest t with s_pref[t] > it (branchless, 6 fixed steps)
#pragma unroll
                for (int st = kWave / 2; st > 0; st >>= 1) t += (s_pref[t + st - 1] <= it) ? st : 0;
                t = live ? t : 0;
                const uint32_t first = t ? s_pref[t - 1] : 0u;
                const uint32_t i0 = p0 + (uint32_t)t + 1u + ((it - first) << 2);           // first candidate slot of the chunk
                const uint32_t nin = live ? min(4u, s_rend[t] - i0) : 0u;
                const uint32_t *ln = s_line[t];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    bool cand = false;
                    int32_t c = 0;
                    if ((uint32_t)k < nin) {
                        c = staged ? s_col[i0 + (uint32_t)k] : ocol[S0 + i0 + (uint32_t)k];
                        uint32_t blk; unsigned long long mask;
                        sig_slot(c, blk, mask);
                        const unsigned long long word = *reinterpret_cast<const unsigned long long *>(ln + 2 * blk);
                        cand = (word & mask) == mask;
                    }
                    const uint64_t cm = __ballot(cand);
                    if (cm) {
                        if (cand) s_cand[n_cand + (uint32_t)__popcll(cm & lanemask_lt())] = make_uint2((uint32_t)t | ((i0 + (uint32_t)k) << 6), (uint32_t)c);
                        n_cand += (uint32_t)__popcll(cm);
                        if (n_cand >= (uint32_t)kWave) search_cands(p0);       // at most 63 are waiting when the next 64 arrive
                    }
                }
            }
            search_cands(p0);                                  // (the parked candidates name their edge by its lane in THIS batch)
        }
        if (!staged) {
            flush_tris();
            if (MODE == TRI_SINGLE && part == 0u) for (uint32_t k = (uint32_t)lane; k < E; k += kWave) { own[S0 + k] = 0u; ownoff[S0 + k] = kOwnSpill; }
            continue;
        }
        if (MODE == TRI_SINGLE) { drain(n_done, n_rec, true, spilled); n_done = n_rec; }
        bool to_dense = MODE == TRI_SINGLE && !spilled;              // wave-uniform
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
                    if (chunk_end > dense_cap) {                // the region has run out: this sub-range's entries become records
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
            for (uint32_t b0 = 0; b0 < n_rec + n_scr; b0 += kWave) {
                const uint32_t x = b0 + (uint32_t)lane;
                if (x < n_rec + n_scr) {
                    const uint2 rc = x < n_rec ? s_rec[x] : scr[x - n_rec];
                    const uint32_t er = rc.x & 0xFFu, ir = (rc.x >> 8) & 0xFFu;
                    dense[base + (uint32_t)s_col[er] + ((rc.x >> 16) & 0xFFu)] = make_int2((int)(S0 + ir), (int)rc.y);
                    dense[base + (uint32_t)s_col[ir] + (rc.x >> 24)] = make_int2((int)(S0 + er), (int)rc.y);
                }
            }
        } else {
            if (MODE == TRI_SINGLE && !spilled) {                // no room in the region: own-role entries of every record become records
                drain(0, n_rec, false, true);
                for (uint32_t b0 = 0; b0 < n_scr; b0 += kWave) {
                    const uint32_t x = b0 + (uint32_t)lane;
                    const bool has = x < n_scr;
                    const uint2 rc = has ? scr[x] : make_uint2(0u, 0u);
                    const uint32_t e = S0 + (rc.x & 0xFFu), i = S0 + ((rc.x >> 8) & 0xFFu), jj = rc.y;
                    rec_append(has, e, make_int2((int)i, (int)jj));
                    rec_append(has, i, make_int2((int)e, (int)jj));
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t k = (uint32_t)lane; k < E; k += kWave) {
                own[S0 + k] = MODE == TRI_SINGLE ? 0u : s_cnt[k];  // (stream: those entries are records, counted with the sorted stream)
                if (MODE == TRI_SINGLE) ownoff[S0 + k] = kOwnSpill;
            }
        }
      }   // sub-ranges
    }
    if (STREAM) for (unsigned long long q = rec_pos + (unsigned long long)lane; q < rec_end && q < ts.cap; q += kWave)
        ts.key[q] = ts.sentinel | (((uint32_t)q & ts.spread_mask) << ts.spread_shift);
}

} // namespace

} // namespace komb
