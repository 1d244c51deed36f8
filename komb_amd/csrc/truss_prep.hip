// truss_prep.hip -- the k-truss side of a graph: what igraph_trussness does to its argument before it lists a triangle
// (reference src/graph.cpp:508; igraph orders the vertices by degree and orients every edge from its lower to its higher
// endpoint), made on the device from the resident symmetric CSR when a k-truss call first needs it --
// inside the call, inside its time (stats.ms_prepare) -- and kept until the graph goes.  k-core, CoreA and the default
// komb2 run never pay for it.
//
// Product (TrussPrep, common.h): INTERNAL vertex ids = ranks in (degree, original id) order, so that the orientation is
// an id compare; the oriented CSR in those ids with ascending rows (internal edge id = oriented slot); the canonical edge
// list (original ids) and the internal id of every canonical edge; one 64-byte line per vertex (start, length, pivots,
// Bloom signature of its oriented row) and the task table of the enumeration (truss_line.h).
//
// How (round 5; rounds 1-4 sorted |E| 12-byte records by a 48-bit key here: 6 radix passes, 4.8 of 14 ms at |E| = 100M).
// No record is sorted.  A row of the oriented CSR is a SUBSET of a row of the symmetric one, so:
//   1. vertices are radix-sorted by degree (stable: ties in id order)                      -> o2i, i2o;
//   2. one pass over the UPPER half of the symmetric rows (every canonical edge (v,w), v < w, once: one gather of o2i[w])
//      writes the canonical edge list and hands the edge to the row of its lower-RANK endpoint as an (internal id of the
//      other endpoint, canonical id) pair in a scratch array shaped like the symmetric CSR: v's own edges from the FRONT of
//      v's region (a running count, no atomic), the edges v gives to w from the BACK of w's region (one atomic on w's back
//      cursor).  d+(v) <= d(v), so the two ends never meet;
//   3. d+ = front count + back count; its scan (in internal order) gives the oriented row pointers;
//   4. one pass per oriented row sorts its <= ~10^2 entries by rank counting in LDS and writes targets, sources, the
//      row's line, and -- every entry carries its canonical id -- the internal id of every canonical edge;
//   5. the task table.
// (Round 5's first version walked BOTH halves of every row -- no atomics, twice the gathers -- and then searched the
// oriented rows for the half of the canonical edges whose lower endpoint has the higher rank: 9.6 ms at |E| = 100M, the
// gathers of 40 MB tables running at ~55 G/s; this one: see DESIGN.md section 4.1.)
// Rows too long for a wavefront (symmetric rows beyond kPrepHeavy slots, oriented rows beyond kRowCap) are done by a
// workgroup each.
#include "truss_line.h"

#include <algorithm>

namespace komb {

namespace {

constexpr uint32_t kPrepHeavy = 2048;           // symmetric rows longer than this are walked in chunks by the whole grid (k_prep_kept_heavy)
constexpr uint32_t kRowCap = 1024;              // oriented entries a wavefront sorts in LDS at a time; longer rows: k_prep_rows_heavy
constexpr uint32_t kRowStage = 8192;            // ... which stages up to this many in LDS
constexpr int kPW = kBlock / kWave;
constexpr int kPrepU = 4;                       // trips of a wavefront whose loads are in flight together (k_prep_kept)

// ---- 1. degrees as sort keys; first upper slot (column above row) of every row and how many there are
__global__ __launch_bounds__(kBlock) void k_prep_vertex(const uint32_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t nv,
                                                        uint32_t *__restrict__ dkey, uint32_t *__restrict__ dval,
                                                        uint32_t *__restrict__ first_upper, uint32_t *__restrict__ upper_cnt)
{
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v <= nv; v += (int64_t)gridDim.x * kBlock) {
        if (v == nv) { upper_cnt[v] = 0u; continue; }
        const uint32_t b = rowptr[v], e = rowptr[v + 1];
        dkey[v] = e - b;
        dval[v] = (uint32_t)v;
        uint32_t lo = b, hi = e;                                  // first slot with col > v (rows hold no loops)
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (col[mid] < (int32_t)v) lo = mid + 1; else hi = mid;
        }
        first_upper[v] = lo;
        upper_cnt[v] = e - lo;
    }
}

// ... and every vertex' REGION of the scratch array step 2 fills: d(v) entries at regbase[rank of v] (the exclusive scan of the
// sorted degrees), i.e. the regions follow each other in INTERNAL order -- step 4 walks internal vertices and reads them as a
// stream (with the regions where the symmetric CSR has the rows, step 4 spent 1.3 of its 2.6 ms on two random places per row).
// Regions and back cursors are addressed by internal id: whoever hands an edge to a vertex has gathered that id anyway, and
// this kernel scatters one word per vertex (o2i), not three
__global__ __launch_bounds__(kBlock) void k_prep_invert(const uint32_t *__restrict__ sorted_ids, const uint32_t *__restrict__ sorted_deg,
                                                        const uint32_t *__restrict__ regbase, int64_t nv, int32_t *__restrict__ i2o, int32_t *__restrict__ o2i,
                                                        uint32_t *__restrict__ backcur, uint32_t *__restrict__ nlocal)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kBlock) {
        const uint32_t v = sorted_ids[i];
        i2o[i] = (int32_t)v;
        o2i[v] = (int32_t)i;
        const uint32_t r0 = regbase[i], d = sorted_deg[i];
        backcur[i] = r0 + d - 1u;                                 // (by INTERNAL id) the last position of the region: counted DOWN by the edges the vertex is given
        if (d > kPrepHeavy) nlocal[v] = 0u;                       // (k_prep_kept_heavy counts the kept edges of a long row UP from here)
    }
}

// owner of flattened position `it`: smallest t with s_end[t] > it (s_end ascending, 64 entries; branchless, 6 fixed steps)
__device__ __forceinline__ int owner_of(const uint32_t *s_end, uint32_t it)
{
    int lo = 0;
#pragma unroll
    for (int st = kWave / 2; st > 0; st >>= 1) lo += (s_end[lo + st - 1] <= it) ? st : 0;
    return lo;
}

// ---- 2. every canonical edge to the row of its lower-rank endpoint.  A wavefront takes 64 consecutive ORIGINAL vertices; the
// UPPER parts of their rows (columns above the row: the canonical edges, ids ebase[v] + offset) are flattened over the lanes.
// Edge (v, w): a = o2i[v], b = o2i[w] (the one gather).  b > a: the oriented edge a -> b is v's: tmp[regbase[a] + r] = (b, k),
// r = v's running count.  b < a: it is w's: tmp[backcur[b]--] = (a, k).
__global__ __launch_bounds__(kBlock) void k_prep_kept(const uint32_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t nv,
                                                      const int32_t *__restrict__ o2i, const uint32_t *__restrict__ fu, const uint32_t *__restrict__ ebase,
                                                      const uint32_t *__restrict__ regbase,
                                                      uint2 *__restrict__ tmp, uint32_t *__restrict__ nlocal, uint32_t *__restrict__ backcur)
{
    __shared__ uint32_t sh_end[kPW][kWave], sh_beg[kPW][kWave], sh_a[kPW][kWave], sh_rp[kPW][kWave], sh_eb[kPW][kWave], sh_cnt[kPW][kWave];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    uint32_t *s_end = sh_end[w], *s_beg = sh_beg[w], *s_a = sh_a[w], *s_rp = sh_rp[w], *s_eb = sh_eb[w], *s_cnt = sh_cnt[w];
    const int64_t gw = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * kBlock) >> 6;
    const int64_t ntasks = (nv + kWave - 1) / kWave;
    for (int64_t task = gw; task < ntasks; task += nw) {
        const int64_t v0 = task * kWave, v = v0 + lane;
        const bool has = v < nv;
        const uint32_t beg = has ? rowptr[v] : 0u, end = has ? rowptr[v + 1] : 0u, up = has ? fu[v] : 0u;
        const bool heavy = end - beg > kPrepHeavy;                 // (left to k_prep_kept_heavy: no slot of it is walked here)
        const uint32_t incl = wave_incl_scan(heavy ? 0u : end - up);
        const uint32_t total = (uint32_t)__shfl((int)incl, kWave - 1);
        __builtin_amdgcn_wave_barrier();
        s_end[lane] = incl;
        s_beg[lane] = up;
        const uint32_t av = has ? (uint32_t)o2i[v] : 0u;
        s_a[lane] = av;
        s_rp[lane] = has ? regbase[av] : 0u;                        // v's region of tmp
        s_eb[lane] = has ? ebase[v] : 0u;
        s_cnt[lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        // kPrepU trips at a time: their owner searches (LDS) first, then all their column loads, then all their rank gathers
        for (uint32_t it0 = 0; it0 < total; it0 += kWave * kPrepU) {
            int t[kPrepU];
            uint32_t rowstart[kPrepU], j[kPrepU], b[kPrepU];
            int32_t wv[kPrepU];
            bool on[kPrepU];
#pragma unroll
            for (int u = 0; u < kPrepU; ++u) {
                const uint32_t it = it0 + (uint32_t)(u * kWave + lane);
                on[u] = it < total;
                t[u] = on[u] ? owner_of(s_end, it) : 0;
                rowstart[u] = t[u] ? s_end[t[u] - 1] : 0u;         // flattened position of the row's first upper slot
                j[u] = s_beg[t[u]] + (it - rowstart[u]);
            }
#pragma unroll
            for (int u = 0; u < kPrepU; ++u) wv[u] = on[u] ? col[j[u]] : 0;
#pragma unroll
            for (int u = 0; u < kPrepU; ++u) b[u] = on[u] ? (uint32_t)o2i[wv[u]] : 0u;
#pragma unroll
            for (int u = 0; u < kPrepU; ++u) {
                const uint32_t base = it0 + (uint32_t)(u * kWave), it = base + (uint32_t)lane;
                if (base >= total) break;                           // (wave-uniform)
                const int tt = t[u];
                const uint32_t a = s_a[tt];
                const uint32_t k = s_eb[tt] + (it - rowstart[u]);   // the canonical id of (v, w)
                const bool mine = on[u] && b[u] > a, theirs = on[u] && b[u] < a;
                if (theirs) tmp[atomicSub(&backcur[b[u]], 1u)] = make_uint2(a, k);
                const uint64_t K = __ballot(mine);
                const uint32_t f = rowstart[u] > base ? rowstart[u] - base : 0u;      // the lane where row tt starts in this trip (<= lane)
                const uint32_t r = s_cnt[tt] + (uint32_t)__popcll(K & lanemask_lt() & ~((1ull << f) - 1ull));
                if (mine) tmp[(size_t)s_rp[tt] + r] = make_uint2(b[u], k);
                __builtin_amdgcn_wave_barrier();
                // the last lane of every row's segment of this trip carries the row's count forward
                if (on[u] && (it + 1u == s_end[tt] || lane == kWave - 1)) s_cnt[tt] = r + (mine ? 1u : 0u);
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (has && !heavy) nlocal[v] = s_cnt[lane];
        __builtin_amdgcn_wave_barrier();
    }
}

// the same for the rows beyond kPrepHeavy slots (the tail of the degree order).  Their upper parts are cut into chunks of
// kHeavyChunk slots, numbered through all heavy rows; wavefront i of the grid takes chunks i, i + W, ...: a hub of 10^5 slots is
// walked by hundreds of wavefronts (a workgroup per row took 0.64 ms at C3 for its longest row: ~100 trips of gather -> atomic ->
// store).  An edge the row keeps takes its place in the front part from the row's own counter (nlocal, zeroed by k_prep_invert;
// one atomic per wavefront and 64 slots) -- step 4 sorts the row anyway.
constexpr uint32_t kHeavyChunk = 256;
__global__ __launch_bounds__(kBlock) void k_prep_kept_heavy(const uint32_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t nv,
                                                            const uint32_t *__restrict__ sorted_deg, const int32_t *__restrict__ i2o,
                                                            const int32_t *__restrict__ o2i, const uint32_t *__restrict__ fu, const uint32_t *__restrict__ ebase,
                                                            const uint32_t *__restrict__ regbase,
                                                            uint2 *__restrict__ tmp, uint32_t *__restrict__ nlocal, uint32_t *__restrict__ backcur)
{
    const int lane = lane_id();
    const uint32_t wi = (uint32_t)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6), W = (uint32_t)(((int64_t)gridDim.x * kBlock) >> 6);
    uint32_t cbase = 0;                                            // chunks of the heavy rows before this block of 64 rows (< ns / kHeavyChunk + nv)
    for (int64_t h0 = 0; h0 < nv; h0 += kWave) {
        const int64_t a = nv - 1 - h0 - lane;                      // internal ids from the top: degrees descending
        const uint32_t deg = a >= 0 ? sorted_deg[a] : 0u;
        const bool hv = deg > kPrepHeavy;
        const uint64_t HM = __ballot(hv);                          // (a prefix of the lanes)
        if (HM == 0) break;
        int32_t v = 0;
        uint32_t fuv = 0, eb = 0, reg = 0, end = 0;
        if (hv) { v = i2o[a]; fuv = fu[v]; eb = ebase[v]; reg = regbase[a]; end = rowptr[v] + deg; }
        const uint32_t nch = hv ? (end - fuv + kHeavyChunk - 1) / kHeavyChunk : 0u;
        uint32_t inc = nch;                                        // inclusive scan of the chunk counts over the lanes
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, d); if (lane >= d) inc += t; }
        const int nh = __popcll(HM);
        const uint32_t tot = (uint32_t)__shfl((int)inc, kWave - 1);
        // this wavefront's chunks among the tot chunks of these rows: global numbers cbase + rel with (cbase + rel) % W == wi
        for (uint32_t rel = (wi + W - (cbase % W)) % W; rel < tot; rel += W) {
            const int t = __popcll(__ballot(inc <= rel));          // the row the chunk is in (inc is non-decreasing over the lanes)
            const uint32_t c = rel - ((uint32_t)__shfl((int)inc, t) - (uint32_t)__shfl((int)nch, t));
            const int32_t rv = __shfl(v, t);
            const uint32_t ra = (uint32_t)(nv - 1 - h0 - t);
            const uint32_t r_fu = (uint32_t)__shfl((int)fuv, t), r_eb = (uint32_t)__shfl((int)eb, t), r_reg = (uint32_t)__shfl((int)reg, t), r_end = (uint32_t)__shfl((int)end, t);
            {
                const uint32_t j0 = r_fu + c * kHeavyChunk;
                constexpr int kU = (int)(kHeavyChunk / kWave);
                bool on[kU];
                int32_t wv[kU];
                uint32_t b[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) { const uint32_t j = j0 + (uint32_t)(u * kWave + lane); on[u] = j < r_end; wv[u] = on[u] ? col[j] : 0; }
#pragma unroll
                for (int u = 0; u < kU; ++u) b[u] = on[u] ? (uint32_t)o2i[wv[u]] : 0u;
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const uint32_t k = r_eb + (j0 + (uint32_t)(u * kWave + lane) - r_fu);
                    const bool mine = on[u] && b[u] > ra, theirs = on[u] && b[u] < ra;
                    if (theirs) tmp[atomicSub(&backcur[b[u]], 1u)] = make_uint2(ra, k);
                    const uint64_t K = __ballot(mine);
                    if (K) {
                        const int lead = __ffsll((long long)K) - 1;
                        uint32_t base = 0;
                        if (lane == lead) base = atomicAdd(&nlocal[rv], (uint32_t)__popcll(K));
                        base = (uint32_t)__shfl((int)base, lead);
                        if (mine) tmp[(size_t)r_reg + base + (uint32_t)__popcll(K & lanemask_lt())] = make_uint2(b[u], k);
                    }
                }
            }
        }
        cbase += tot;
        if (nh < kWave) break;
    }
}

// ---- 3. d+(v) = the edges v kept itself + the edges it was given; by internal id, for the scan -- and, also by internal id, where
// step 4 finds the two parts of the vertex' list: { start of the front part, its length, start of the back part } (one
// coalesced 16-byte read per row there instead of four gathers through i2o)
__global__ __launch_bounds__(kBlock) void k_prep_dplus(int64_t nv, const int32_t *__restrict__ i2o, const uint32_t *__restrict__ sorted_deg,
                                                       const uint32_t *__restrict__ regbase, const uint32_t *__restrict__ backcur,
                                                       const uint32_t *__restrict__ nlocal,
                                                       uint32_t *__restrict__ dplus_i, uint4 *__restrict__ where_i, unsigned long long *__restrict__ own_bound)
{
    unsigned long long ob = 0;
    for (int64_t a = (int64_t)blockIdx.x * kBlock + threadIdx.x; a < nv; a += (int64_t)gridDim.x * kBlock) {
        const uint32_t nl = nlocal[i2o[a]], bk = backcur[a] + 1u, reg = regbase[a];       // (the one gather: the kept count is by original id)
        const unsigned long long d = nl + (reg + sorted_deg[a] - bk);
        dplus_i[a] = (uint32_t)d;
        where_i[a] = make_uint4(reg, nl, bk, 0u);
        ob += d ? d * (d - 1ull) : 0ull;
    }
    block_add_u64(ob, own_bound);
}

// ---- 4. the oriented rows.  A wavefront takes 64 consecutive INTERNAL vertices; their lists (step 2: nlocal entries at the
// front of the vertex' region, the rest behind its back cursor) are read into LDS in batches of <= kRowCap entries (whole
// rows), every entry is ranked among its row's by counting, and leaves as the row's rank-th slot.
__global__ __launch_bounds__(kBlock) void k_prep_rows(const uint32_t *__restrict__ orow, int64_t nv, const uint4 *__restrict__ where_i,
                                                      const uint2 *__restrict__ tmp, int32_t *__restrict__ ocol,
                                                      uint32_t *__restrict__ e2k, uint4 *__restrict__ line)
{
    __shared__ int32_t sh_b[kPW][kRowCap];
    __shared__ uint32_t sh_kk[kPW][kRowCap];
    __shared__ unsigned long long sh_sig[kPW][kWave][kSigBlocks];
    __shared__ int32_t sh_piv[kPW][kWave][kPivots];
    __shared__ uint32_t sh_end[kPW][kWave], sh_rp[kPW][kWave], sh_ob[kPW][kWave], sh_nl[kPW][kWave], sh_bk[kPW][kWave];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    int32_t *s_b = sh_b[w];
    uint32_t *s_kk = sh_kk[w], *s_end = sh_end[w], *s_rp = sh_rp[w], *s_ob = sh_ob[w], *s_nl = sh_nl[w], *s_bk = sh_bk[w];
    unsigned long long (*s_sig)[kSigBlocks] = sh_sig[w];
    int32_t (*s_piv)[kPivots] = sh_piv[w];
    const int64_t gw = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * kBlock) >> 6;
    const int64_t ntasks = (nv + kWave - 1) / kWave;
    for (int64_t task = gw; task < ntasks; task += nw) {
        const int64_t a0 = task * kWave, a = a0 + lane;
        const bool has = a < nv;
        const uint32_t ob = has ? orow[a] : 0u, d = has ? orow[a + 1] - ob : 0u;
        const bool big = d > kRowCap;
        const uint4 wh = has ? where_i[a] : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t len = big ? 0u : d;
        __builtin_amdgcn_wave_barrier();
        s_rp[lane] = wh.x;                                         // the vertex' own edges: tmp[wh.x .. + wh.y)
        s_nl[lane] = wh.y;
        s_bk[lane] = wh.z;                                         // the edges it was given: tmp[wh.z ..)
        s_ob[lane] = ob;
#pragma unroll
        for (int x = 0; x < kSigBlocks; ++x) s_sig[lane][x] = 0ull;
#pragma unroll
        for (int x = 0; x < kPivots; ++x) s_piv[lane][x] = 0x7FFFFFFF;
        for (int done = 0; done < kWave;) {                       // (wave-uniform) rows [done, r1) form the next batch
            const uint32_t incl = wave_incl_scan(lane >= done ? len : 0u);
            const int nfit = __popcll(__ballot(lane >= done && incl <= kRowCap));       // a prefix of the lanes from `done` (>= 1: len <= kRowCap)
            const int r1 = done + nfit;
            const uint32_t E = (uint32_t)__shfl((int)incl, r1 - 1);
            __builtin_amdgcn_wave_barrier();
            s_end[lane] = lane < done ? 0u : (lane < r1 ? incl : 0xFFFFFFFFu);
            __builtin_amdgcn_wave_barrier();
            for (uint32_t k = (uint32_t)lane; k < E; k += kWave) {
                const int t = owner_of(s_end, k);
                const uint32_t first = t ? s_end[t - 1] : 0u;
                const uint32_t idx = k - first, nl = s_nl[t];
                const uint2 ent = tmp[idx < nl ? (size_t)s_rp[t] + idx : (size_t)s_bk[t] + (idx - nl)];
                s_b[k] = (int32_t)ent.x; s_kk[k] = ent.y;
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t k = (uint32_t)lane; k < E; k += kWave) {
                const int t = owner_of(s_end, k);
                const uint32_t first = t ? s_end[t - 1] : 0u, last = s_end[t];
                const int32_t mine = s_b[k];
                uint32_t rank = 0;
                for (uint32_t x = first; x < last; ++x) rank += s_b[x] < mine ? 1u : 0u;
                const uint32_t e = s_ob[t] + rank;
                ocol[e] = mine;
                e2k[e] = s_kk[k];
                uint32_t blk; unsigned long long mask;
                sig_slot(mine, blk, mask);
                atomicOr(&s_sig[t][blk], mask);
                const uint32_t seg = (last - first + (uint32_t)kPivots) / (uint32_t)(kPivots + 1);
                if (rank && rank % seg == 0u && rank / seg <= (uint32_t)kPivots) s_piv[t][rank / seg - 1u] = mine;
            }
            __builtin_amdgcn_wave_barrier();
            done = r1;
        }
        if (has && !big) {
            uint4 *L = line + 4 * a;
            const int32_t *pv = s_piv[lane];
            const unsigned long long *sg = s_sig[lane];
            L[0] = make_uint4(ob, d, (uint32_t)pv[0], (uint32_t)pv[1]);
            L[1] = make_uint4((uint32_t)pv[2], (uint32_t)pv[3], (uint32_t)pv[4], (uint32_t)pv[5]);
            L[2] = make_uint4((uint32_t)sg[0], (uint32_t)(sg[0] >> 32), (uint32_t)sg[1], (uint32_t)(sg[1] >> 32));
            L[3] = make_uint4((uint32_t)sg[2], (uint32_t)(sg[2] >> 32), (uint32_t)sg[3], (uint32_t)(sg[3] >> 32));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// the oriented rows beyond kRowCap entries, a workgroup each: the same rank count, over LDS up to kRowStage entries, over
// global memory beyond (quadratic in a row that long: a graph with such rows is far beyond the index's triangle limit)
__global__ __launch_bounds__(kBlock) void k_prep_rows_heavy(const uint32_t *__restrict__ hlist, const uint32_t *__restrict__ hcount,
                                                            const uint32_t *__restrict__ orow, const uint4 *__restrict__ where_i,
                                                            const uint2 *__restrict__ tmp, int32_t *__restrict__ ocol,
                                                            uint32_t *__restrict__ e2k, uint4 *__restrict__ line, uint32_t stage_cap)
{
    // stage_cap: rows up to this many entries are ranked out of LDS (kRowStage; less only in tests: option PREP_ROW_STAGE)
    __shared__ int32_t sh_b[kRowStage];
    __shared__ unsigned long long sh_sig[kSigBlocks];
    __shared__ int32_t sh_piv[kPivots];
    const uint32_t n = *hcount;
    for (uint32_t h = blockIdx.x; h < n; h += gridDim.x) {
        const uint32_t a = hlist[h], ob = orow[a], d = orow[a + 1] - ob;
        const uint4 wh = where_i[a];
        const uint32_t nl = wh.y;
        const uint2 *front = tmp + wh.x, *back = tmp + wh.z;
        auto entry = [&](uint32_t i) -> uint2 { return i < nl ? front[i] : back[i - nl]; };
        const bool staged = d <= stage_cap;
        __syncthreads();
        if (threadIdx.x < (uint32_t)kSigBlocks) sh_sig[threadIdx.x] = 0ull;
        if (threadIdx.x < (uint32_t)kPivots) sh_piv[threadIdx.x] = 0x7FFFFFFF;
        if (staged) for (uint32_t i = threadIdx.x; i < d; i += kBlock) sh_b[i] = (int32_t)entry(i).x;
        __syncthreads();
        const uint32_t seg = (d + (uint32_t)kPivots) / (uint32_t)(kPivots + 1);
        for (uint32_t i = threadIdx.x; i < d; i += kBlock) {
            const uint2 ent = entry(i);
            const int32_t mine = (int32_t)ent.x;
            uint32_t rank = 0;
            if (staged) for (uint32_t x = 0; x < d; ++x) rank += sh_b[x] < mine ? 1u : 0u;
            else for (uint32_t x = 0; x < d; ++x) rank += (int32_t)entry(x).x < mine ? 1u : 0u;
            const uint32_t e = ob + rank;
            ocol[e] = mine;
            e2k[e] = ent.y;
            uint32_t blk; unsigned long long mask;
            sig_slot(mine, blk, mask);
            atomicOr(&sh_sig[blk], mask);
            if (rank && rank % seg == 0u && rank / seg <= (uint32_t)kPivots) sh_piv[rank / seg - 1u] = mine;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint4 *L = line + 4 * (int64_t)a;
            L[0] = make_uint4(ob, d, (uint32_t)sh_piv[0], (uint32_t)sh_piv[1]);
            L[1] = make_uint4((uint32_t)sh_piv[2], (uint32_t)sh_piv[3], (uint32_t)sh_piv[4], (uint32_t)sh_piv[5]);
            L[2] = make_uint4((uint32_t)sh_sig[0], (uint32_t)(sh_sig[0] >> 32), (uint32_t)sh_sig[1], (uint32_t)(sh_sig[1] >> 32));
            L[3] = make_uint4((uint32_t)sh_sig[2], (uint32_t)(sh_sig[2] >> 32), (uint32_t)sh_sig[3], (uint32_t)(sh_sig[3] >> 32));
        }
    }
}

// ---- 5. the enumeration's tasks (truss_line.h) + the list of the oriented rows beyond kRowCap (made with the counts: one
// pass over the row pointers)
__global__ __launch_bounds__(kBlock) void k_task_count(const uint32_t *__restrict__ orow, int64_t nv, int group, uint32_t *__restrict__ cnt,
                                                       uint32_t *__restrict__ hlist, uint32_t *__restrict__ hcount, uint32_t hcap)
{
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v <= nv; v += (int64_t)gridDim.x * kBlock) {
        uint32_t c = 0;
        if (v < nv) {
            const uint32_t d = orow[v + 1] - orow[v];
            if (task_starts(orow, v, group)) c = task_parts(d);
            if (hlist && d > kRowCap) { const uint32_t p = atomicAdd(hcount, 1u); if (p < hcap) hlist[p] = (uint32_t)v; }
        }
        cnt[v] = c;
    }
}

// toff = exclusive scan of the counts; a vertex that starts a task runs to the next start (at most `group` <= 63 vertices on)
__global__ __launch_bounds__(kBlock) void k_task_fill(const uint32_t *__restrict__ toff, int64_t nv, int group, uint2 *__restrict__ tasks)
{
    const int lane = lane_id();
    for (int64_t v0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x - lane); v0 < nv; v0 += (int64_t)gridDim.x * kBlock) {
        const int64_t v = v0 + lane;
        const uint32_t t0 = v <= nv ? toff[v] : 0u, t1 = v < nv ? toff[v + 1] : t0;
        const uint32_t u0 = v + kWave <= nv ? toff[v + kWave] : 0u, u1 = v + kWave < nv ? toff[v + kWave + 1] : u0;
        const uint64_t here = __ballot(t1 != t0), next = __ballot(u1 != u0);       // which vertices of this window / the next one start a task
        if (t1 == t0) continue;
        // distance to the next start: above this lane in `here`, else in `next`
        const uint64_t up = lane == kWave - 1 ? 0ull : here >> (lane + 1);
        uint32_t dist = up ? (uint32_t)__ffsll((long long)up) : (next ? (uint32_t)(kWave - 1 - lane) + (uint32_t)__ffsll((long long)next) : 0xFFFFFFFFu);
        dist = min(dist, (uint32_t)group);
        if ((int64_t)dist > nv - v) dist = (uint32_t)(nv - v);
        const uint32_t n = t1 - t0;
        for (uint32_t p = 0; p < n; ++p) tasks[t0 + p] = make_uint2((uint32_t)v, dist | (p << 6) | (n << 19));
    }
}

__global__ void k_prep_flags(const uint32_t *__restrict__ toff, int64_t nv, const uint32_t *__restrict__ hcount, const unsigned long long *__restrict__ own_bound,
                             const uint32_t *__restrict__ orow, unsigned long long *__restrict__ out)
{
    if (threadIdx.x == 0) { out[0] = toff[nv]; out[1] = *own_bound; out[2] = orow[nv]; out[3] = *hcount; }
}

// sum_v d(v)^2, sum_e min(d(u),d(v)), max d, sum over the oriented edges of d+(a) + d+(b): the roofline model's inputs
__global__ __launch_bounds__(kBlock) void k_graph_moments(const uint32_t *__restrict__ rowptr, int64_t nv, const int32_t *__restrict__ i2o,
                                                          const int32_t *__restrict__ osrc, const int32_t *__restrict__ ocol,
                                                          int64_t m, const uint32_t *__restrict__ orow, unsigned long long *out)
{
    unsigned long long s2 = 0, smin = 0, mx = 0, so = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kBlock) {
        const unsigned long long d = rowptr[i + 1] - rowptr[i];
        s2 += d * d;
        mx = d > mx ? d : mx;
    }
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < m; e += (int64_t)gridDim.x * kBlock) {
        const int32_t a = osrc[e], b = ocol[e];
        const int32_t va = i2o[a], vb = i2o[b];
        smin += (unsigned long long)min(rowptr[va + 1] - rowptr[va], rowptr[vb + 1] - rowptr[vb]);
        so += (unsigned long long)(orow[a + 1] - orow[a]) + (unsigned long long)(orow[b + 1] - orow[b]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s2 += __shfl_xor(s2, o); smin += __shfl_xor(smin, o); so += __shfl_xor(so, o);
        const unsigned long long t = __shfl_xor(mx, o); mx = t > mx ? t : mx;
    }
    if (lane_id() == 0) { atomicAdd(&out[0], s2); atomicAdd(&out[1], smin); atomicMax(&out[2], mx); atomicAdd(&out[4], so); }
}

// ---- induced subgraph (a5: igraph_induced_subgraph_map, reference src/graph.cpp:502) as a symmetric CSR of its own.
// New ids are ranks among the kept vertices (monotone: rows stay ascending, the canonical edge order is the original one's).
__global__ __launch_bounds__(kBlock) void k_mask_flags(const uint8_t *__restrict__ mask, int64_t nv, uint32_t *__restrict__ flag)
{
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v <= nv; v += (int64_t)gridDim.x * kBlock) flag[v] = v < nv && mask[v] ? 1u : 0u;
}
__global__ __launch_bounds__(kBlock) void k_mask_list(const uint8_t *__restrict__ mask, const uint32_t *__restrict__ vnew, int64_t nv, int32_t *__restrict__ vold)
{
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += (int64_t)gridDim.x * kBlock) if (mask[v]) vold[vnew[v]] = (int32_t)v;
}
// the kept neighbours of kept vertex x, in order: counted (FILL = false) or written.  One wavefront per row up to kPrepHeavy
// slots; longer rows (the hubs a max-core mask selects) by a workgroup each (WG = true), from the list the counting
// wavefronts leave (hlist / hcount).
template <bool FILL, bool WG>
__global__ __launch_bounds__(kBlock) void k_induce(const uint32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const uint8_t *__restrict__ mask,
                                                   const uint32_t *__restrict__ vnew, const int32_t *__restrict__ vold, int64_t nk,
                                                   uint32_t *__restrict__ subdeg, const uint32_t *__restrict__ sub_rowptr, int32_t *__restrict__ sub_col,
                                                   uint32_t *__restrict__ hlist, uint32_t *__restrict__ hcount)
{
    __shared__ uint32_t sh_wc[kBlock / kWave];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    const int64_t first = WG ? (int64_t)blockIdx.x : (((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
    const int64_t step = WG ? (int64_t)gridDim.x : (((int64_t)gridDim.x * kBlock) >> 6);
    const uint32_t width = WG ? (uint32_t)kBlock : (uint32_t)kWave, me = WG ? threadIdx.x : (uint32_t)lane;
    const int64_t n = WG ? (int64_t)*hcount : nk;
    for (int64_t i = first; i < n; i += step) {
        const int64_t x = WG ? (int64_t)hlist[i] : i;
        const int32_t v = vold[x];
        const uint32_t beg = rowptr[v], end = rowptr[v + 1];
        if (!WG && end - beg > kPrepHeavy) {                       // (uniform per wavefront)
            if (!FILL && lane == 0) hlist[atomicAdd(hcount, 1u)] = (uint32_t)x;
            continue;
        }
        uint32_t run = 0;
        const uint32_t out0 = FILL ? sub_rowptr[x] : 0u;
        for (uint32_t j0 = beg; j0 < end; j0 += width) {
            const uint32_t j = j0 + me;
            const int32_t wv = j < end ? col[j] : 0;
            const bool keep = j < end && mask[wv];
            const uint64_t K = __ballot(keep);
            uint32_t before = 0, all = (uint32_t)__popcll(K);
            if (WG) {
                __syncthreads();
                if (lane == 0) sh_wc[w] = all;
                __syncthreads();
                all = 0;
#pragma unroll
                for (int q = 0; q < kBlock / kWave; ++q) { const uint32_t c = sh_wc[q]; if (q < w) before += c; all += c; }
            }
            if (FILL && keep) sub_col[out0 + run + before + (uint32_t)__popcll(K & lanemask_lt())] = (int32_t)vnew[wv];
            run += all;
        }
        if (!FILL && me == 0) subdeg[x] = run;
    }
}

// the canonical edge list of a symmetric CSR: edge k = the k-th upper slot (column above row) in row order, endpoints through
// `vold` when the CSR is an induced subgraph's (new -> original ids).  What igraph_edge answers after igraph_trussness
// (reference src/graph.cpp:529-532): made when the endpoints are asked for, not inside the timed call.  A wavefront per row.
__global__ __launch_bounds__(kBlock) void k_edge_list(const uint32_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t nv,
                                                      const uint32_t *__restrict__ fu, const uint32_t *__restrict__ ebase, const int32_t *__restrict__ vold,
                                                      int32_t *__restrict__ eu, int32_t *__restrict__ ev)
{
    const int lane = lane_id();
    const int64_t gw = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * kBlock) >> 6;
    for (int64_t v = gw; v < nv; v += nw) {
        const uint32_t j0 = fu[v], j1 = rowptr[v + 1], k0 = ebase[v];
        const int32_t uo = vold ? vold[v] : (int32_t)v;
        for (uint32_t j = j0 + (uint32_t)lane; j < j1; j += kWave) {
            const int32_t w = col[j];
            eu[k0 + (j - j0)] = uo;
            ev[k0 + (j - j0)] = vold ? vold[w] : w;
        }
    }
}

inline int id_bits(int64_t nv)
{
    int vb = 1;
    while (vb < 31 && (1ll << vb) < nv) ++vb;
    return vb;
}

template <class T> hipError_t pool_get(komb_ctx *ctx, T **out, size_t count)
{
    return ctx->pool.get((void **)out, (count ? count : 1) * sizeof(T));
}

} // namespace

void prep_free(komb_ctx *ctx, TrussPrep *p)
{
    void *all[] = {p->o2i, p->i2o, p->orow, p->ocol, p->osrc, p->e2k, p->vline, p->wtasks};   // (osrc: only when prep_sources made it)
    for (void *q : all) ctx->pool.put(q);
    *p = TrussPrep{};
}

int prep_build(komb_ctx *ctx, const uint32_t *rowptr, const int32_t *col, int64_t nv, int64_t ns, TrussPrep *out)
{
    hipStream_t s = ctx->stream;
    const int64_t ne = ns / 2;
    Range r_all("truss: prepare");
    prep_free(ctx, out);
    TrussPrep P;
    P.nv = nv; P.ne = ne;
    struct Fail { komb_ctx *c; TrussPrep *p; bool armed = true; ~Fail() { if (armed) prep_free(c, p); } } fail{ctx, &P};
    EventSet evs;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // start | vertices ordered | edges handed out | rows done | end
    for (auto &e : ev) KOMB_HIP(ctx, evs.make(&e));
    KOMB_HIP(ctx, hipEventRecord(ev[0], s));
    KOMB_HIP(ctx, pool_get(ctx, &P.o2i, (size_t)nv));
    KOMB_HIP(ctx, pool_get(ctx, &P.i2o, (size_t)nv));
    KOMB_HIP(ctx, pool_get(ctx, &P.orow, (size_t)nv + 1));
    KOMB_HIP(ctx, pool_get(ctx, &P.ocol, (size_t)ne + 8));           // + 8: the enumeration and line_find read 16 bytes at a time, past the end of the last row
    KOMB_HIP(ctx, pool_get(ctx, &P.e2k, (size_t)ne));
    KOMB_HIP(ctx, pool_get(ctx, &P.vline, 4 * (size_t)nv));
    // the task table's size is known on the device only: room for the most it can be (a start per vertex, a row of d slots in
    // at most d / 64 + 1 parts); what is not used is never touched
    const size_t task_cap = 2 * (size_t)nv + (size_t)ne / kWave + 64;
    KOMB_HIP(ctx, pool_get(ctx, (uint2 **)&P.wtasks, task_cap));
    KOMB_HIP(ctx, hipMemsetAsync(P.ocol + ne, 0, 8 * sizeof(int32_t), s));

    DevBufs bufs(ctx);
    uint32_t *d_dk[2] = {nullptr, nullptr}, *d_dv[2] = {nullptr, nullptr}, *d_fu = nullptr, *d_uc = nullptr, *d_ebase = nullptr, *d_dplus = nullptr;
    uint32_t *d_nlocal = nullptr, *d_backcur = nullptr, *d_regbase = nullptr;
    uint4 *d_where = nullptr;
    uint32_t *d_tcnt = nullptr, *d_hlist = nullptr, *d_words = nullptr;
    unsigned long long *d_acc = nullptr;                             // [0] own bound, [1..4] what the host reads back
    uint2 *d_tmp = nullptr;
    for (int i = 0; i < 2; ++i) { KOMB_HIP(ctx, bufs.alloc(&d_dk[i], (size_t)nv)); KOMB_HIP(ctx, bufs.alloc(&d_dv[i], (size_t)nv)); }
    KOMB_HIP(ctx, bufs.alloc(&d_fu, (size_t)nv + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_uc, (size_t)nv + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_ebase, (size_t)nv + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_dplus, (size_t)nv + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_nlocal, (size_t)nv));
    KOMB_HIP(ctx, bufs.alloc(&d_where, (size_t)nv));
    KOMB_HIP(ctx, bufs.alloc(&d_regbase, (size_t)nv + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_backcur, (size_t)nv));
    KOMB_HIP(ctx, bufs.alloc(&d_tcnt, (size_t)nv + 1));
    const uint32_t hcap = (uint32_t)(ne / kRowCap + 64);
    KOMB_HIP(ctx, bufs.alloc(&d_hlist, (size_t)hcap));
    KOMB_HIP(ctx, bufs.alloc(&d_words, 4));                          // [0] rows beyond kRowCap
    KOMB_HIP(ctx, bufs.alloc(&d_acc, 8));
    KOMB_HIP(ctx, bufs.alloc(&d_tmp, (size_t)ns));
    KOMB_HIP(ctx, hipMemsetAsync(d_words, 0, 4 * sizeof(uint32_t), s));
    KOMB_HIP(ctx, hipMemsetAsync(d_acc, 0, 8 * sizeof(unsigned long long), s));
    KOMB_HIP(ctx, hipMemsetAsync(d_dplus + nv, 0, sizeof(uint32_t), s));

    const int gv = grid_for(nv + 1);
    const int gwave = grid_for((nv + kWave - 1) / kWave, kPW, 256 * 8);
    // 1. vertices by (degree, original id): a degree is below nv, so the sort looks at that many bits only
    k_prep_vertex<<<gv, kBlock, 0, s>>>(rowptr, col, nv, d_dk[0], d_dv[0], d_fu, d_uc);
    uint32_t *sk = d_dk[0], *sv = d_dv[0];
    if (nv > 0) {
        KOMB_TRY(prim_sort_pairs_u32_u32(ctx, d_dk[0], d_dk[1], d_dv[0], d_dv[1], nv, id_bits(nv), &sk, &sv));
        KOMB_TRY(prim_exclusive_sum_u32(ctx, sk, d_regbase, nv));                   // the scratch regions, in internal order (the last one ends at ns)
        k_prep_invert<<<grid_for(nv), kBlock, 0, s>>>(sv, sk, d_regbase, nv, P.i2o, P.o2i, d_backcur, d_nlocal);
    }
    KOMB_TRY(prim_exclusive_sum_u32(ctx, d_uc, d_ebase, nv + 1));
    KOMB_HIP(ctx, hipEventRecord(ev[1], s));
    // 2. every canonical edge to the row of its lower-rank endpoint; the canonical edge list; 3. d+ and the oriented row pointers
    if (nv > 0) {
        k_prep_kept<<<gwave, kBlock, 0, s>>>(rowptr, col, nv, P.o2i, d_fu, d_ebase, d_regbase, d_tmp, d_nlocal, d_backcur);
        k_prep_kept_heavy<<<256 * 4, kBlock, 0, s>>>(rowptr, col, nv, sk, P.i2o, P.o2i, d_fu, d_ebase, d_regbase, d_tmp, d_nlocal, d_backcur);
        k_prep_dplus<<<grid_for(nv), kBlock, 0, s>>>(nv, P.i2o, sk, d_regbase, d_backcur, d_nlocal, d_dplus, d_where, d_acc);
    }
    KOMB_TRY(prim_exclusive_sum_u32(ctx, d_dplus, P.orow, nv + 1));
    KOMB_HIP(ctx, hipEventRecord(ev[2], s));
    // 5a. task counts + the list of the long oriented rows (one pass over the row pointers)
    const int group = (int)std::max<int64_t>(1, std::min<int64_t>(kWedgeV, nv / (256 * KOMB_WEDGE_EU * kTriWaves * 4)));
    k_task_count<<<gv, kBlock, 0, s>>>(P.orow, nv, group, d_tcnt, d_hlist, d_words, hcap);
    // 4. oriented rows, lines, the canonical map
    if (nv > 0) {
        k_prep_rows<<<gwave, kBlock, 0, s>>>(P.orow, nv, d_where, d_tmp, P.ocol, P.e2k, P.vline);
        uint32_t stage_cap = kRowStage;
        if (const char *e = ctx_opt(ctx, "PREP_ROW_STAGE")) stage_cap = std::min<uint32_t>(kRowStage, (uint32_t)strtoul(e, nullptr, 10));   // (tests: the unstaged path)
        k_prep_rows_heavy<<<1024, kBlock, 0, s>>>(d_hlist, d_words, P.orow, d_where, d_tmp, P.ocol, P.e2k, P.vline, stage_cap);
    }
    KOMB_HIP(ctx, hipEventRecord(ev[3], s));
    // 5b. tasks
    KOMB_TRY(prim_exclusive_sum_u32(ctx, d_tcnt, d_tcnt, nv + 1));
    if (nv > 0) k_task_fill<<<grid_for(nv), kBlock, 0, s>>>(d_tcnt, nv, group, (uint2 *)P.wtasks);
    k_prep_flags<<<1, 64, 0, s>>>(d_tcnt, nv, d_words, d_acc, P.orow, d_acc + 1);
    KOMB_HIP(ctx, hipEventRecord(ev[4], s));
    unsigned long long h[4] = {0, 0, 0, 0};
    KOMB_HIP(ctx, d2h(ctx, h, d_acc + 1, sizeof(h)));
    float ms = 0.f, part[4] = {0.f, 0.f, 0.f, 0.f};
    (void)hipEventElapsedTime(&ms, ev[0], ev[4]);
    for (int i = 0; i < 4; ++i) (void)hipEventElapsedTime(&part[i], ev[i], ev[i + 1]);
    for (int i = 0; i < 4; ++i) P.ms_part[i] = (double)part[i];
    if (h[2] != (unsigned long long)ne) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "k-truss preparation: %llu oriented slots for %lld edges", h[2], (long long)ne);
    if (h[3] > hcap) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "k-truss preparation: %llu oriented rows beyond %u slots, room for %u", h[3], kRowCap, hcap);
    if (h[0] > task_cap) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "k-truss preparation: %llu tasks, room for %zu", h[0], task_cap);
    P.n_wtasks = (int64_t)h[0];
    P.own_bound = (int64_t)h[1];
    P.ms = (double)ms;
    P.valid = true;
    fail.armed = false;
    *out = P;
    return KOMB_OK;
}

// osrc[e] = the source of oriented slot e: read by the LDS tail (option FINISH=lds) and by the moments kernel only, so it is
// made when one of them asks (it used to be one more 4-byte stream out of k_prep_rows in every preparation)
namespace {
__global__ __launch_bounds__(kBlock) void k_prep_sources(const uint32_t *__restrict__ orow, int64_t nv, int32_t *__restrict__ osrc)
{
    const int lane = lane_id();
    const int64_t gw = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * kBlock) >> 6;
    for (int64_t a = gw; a < nv; a += nw)
        for (uint32_t e = orow[a] + (uint32_t)lane, e1 = orow[a + 1]; e < e1; e += kWave) osrc[e] = (int32_t)a;
}
} // namespace

int prep_sources(komb_ctx *ctx, TrussPrep *p)
{
    if (p->osrc || !p->valid) return KOMB_OK;
    KOMB_HIP(ctx, ctx->pool.get((void **)&p->osrc, (size_t)(p->ne ? p->ne : 1) * sizeof(int32_t)));
    if (p->nv > 0) k_prep_sources<<<grid_for(p->nv, kPW, 256 * 8), kBlock, 0, ctx->stream>>>(p->orow, p->nv, p->osrc);
    KOMB_HIP(ctx, hipGetLastError());
    return KOMB_OK;
}

int prep_ensure(komb_ctx *ctx)
{
    if (ctx->prep.valid) return KOMB_OK;
    return prep_build(ctx, ctx->d_o_rowptr, ctx->d_o_col, ctx->nv, 2 * ctx->ne, &ctx->prep);
}

int graph_moments(komb_ctx *ctx, int64_t out[5])
{
    KOMB_TRY(prep_ensure(ctx));
    KOMB_TRY(prep_sources(ctx, &ctx->prep));
    const TrussPrep &P = ctx->prep;
    unsigned long long *d_mom = nullptr, h[5] = {0, 0, 0, 0, 0};
    DevBufs bufs(ctx);
    KOMB_HIP(ctx, bufs.alloc(&d_mom, 5));
    KOMB_HIP(ctx, hipMemsetAsync(d_mom, 0, 5 * sizeof(unsigned long long), ctx->stream));
    k_graph_moments<<<1024, kBlock, 0, ctx->stream>>>(ctx->d_o_rowptr, P.nv, P.i2o, P.osrc, P.ocol, P.ne, P.orow, d_mom);
    KOMB_HIP(ctx, d2h(ctx, h, d_mom, sizeof(h)));
    for (int i = 0; i < 5; ++i) out[i] = (int64_t)h[i];
    return KOMB_OK;
}

void induced_free(komb_ctx *ctx, InducedCsr *g)
{
    ctx->pool.put(g->rowptr); ctx->pool.put(g->col); ctx->pool.put(g->vold);
    *g = InducedCsr{};
}

int induce_csr(komb_ctx *ctx, const uint8_t *vmask_host, InducedCsr *out)
{
    hipStream_t s = ctx->stream;
    const int64_t nv = ctx->nv;
    InducedCsr G;
    struct Fail { komb_ctx *c; InducedCsr *g; bool armed = true; ~Fail() { if (armed) induced_free(c, g); } } fail{ctx, &G};
    DevBufs bufs(ctx);
    uint8_t *d_mask = nullptr;
    uint32_t *d_flag = nullptr, *d_vnew = nullptr, *d_subdeg = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_mask, (size_t)nv));
    KOMB_HIP(ctx, bufs.alloc(&d_flag, (size_t)nv + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_vnew, (size_t)nv + 1));
    KOMB_HIP(ctx, hipMemcpyAsync(d_mask, vmask_host, (size_t)nv, hipMemcpyHostToDevice, s));
    k_mask_flags<<<grid_for(nv + 1), kBlock, 0, s>>>(d_mask, nv, d_flag);
    KOMB_TRY(prim_exclusive_sum_u32(ctx, d_flag, d_vnew, nv + 1));
    uint32_t nk = 0;
    KOMB_HIP(ctx, d2h(ctx, &nk, d_vnew + nv, sizeof(uint32_t)));
    G.nv = (int64_t)nk;
    KOMB_HIP(ctx, pool_get(ctx, &G.vold, (size_t)nk));
    KOMB_HIP(ctx, pool_get(ctx, &G.rowptr, (size_t)nk + 1));
    KOMB_HIP(ctx, bufs.alloc(&d_subdeg, (size_t)nk + 1));
    uint32_t n_slots = 0;
    if (nk > 0) {
        uint32_t *d_hlist = nullptr, *d_hcount = nullptr;
        KOMB_HIP(ctx, bufs.alloc(&d_hlist, (size_t)(2 * ctx->ne) / kPrepHeavy + 64));      // rows beyond kPrepHeavy slots: at most that many
        KOMB_HIP(ctx, bufs.alloc(&d_hcount, 1));
        KOMB_HIP(ctx, hipMemsetAsync(d_hcount, 0, sizeof(uint32_t), s));
        k_mask_list<<<grid_for(nv), kBlock, 0, s>>>(d_mask, d_vnew, nv, G.vold);
        KOMB_HIP(ctx, hipMemsetAsync(d_subdeg + nk, 0, sizeof(uint32_t), s));
        const int gw = grid_for(nk, kPW, 256 * 8);
        k_induce<false, false><<<gw, kBlock, 0, s>>>(ctx->d_o_rowptr, ctx->d_o_col, d_mask, d_vnew, G.vold, nk, d_subdeg, nullptr, nullptr, d_hlist, d_hcount);
        k_induce<false, true><<<1024, kBlock, 0, s>>>(ctx->d_o_rowptr, ctx->d_o_col, d_mask, d_vnew, G.vold, nk, d_subdeg, nullptr, nullptr, d_hlist, d_hcount);
        KOMB_TRY(prim_exclusive_sum_u32(ctx, d_subdeg, G.rowptr, (int64_t)nk + 1));
        KOMB_HIP(ctx, d2h(ctx, &n_slots, G.rowptr + nk, sizeof(uint32_t)));
        KOMB_HIP(ctx, pool_get(ctx, &G.col, (size_t)n_slots));
        if (n_slots > 0) {
            k_induce<true, false><<<gw, kBlock, 0, s>>>(ctx->d_o_rowptr, ctx->d_o_col, d_mask, d_vnew, G.vold, nk, nullptr, G.rowptr, G.col, d_hlist, d_hcount);
            k_induce<true, true><<<1024, kBlock, 0, s>>>(ctx->d_o_rowptr, ctx->d_o_col, d_mask, d_vnew, G.vold, nk, nullptr, G.rowptr, G.col, d_hlist, d_hcount);
        }
    } else {
        KOMB_HIP(ctx, hipMemsetAsync(G.rowptr, 0, sizeof(uint32_t), s));
        KOMB_HIP(ctx, pool_get(ctx, &G.col, 1));
    }
    G.ns = (int64_t)n_slots;
    KOMB_HIP(ctx, hipStreamSynchronize(s));                          // (the scratch goes back to the pool when this returns)
    fail.armed = false;
    *out = G;
    return KOMB_OK;
}

int edge_list(komb_ctx *ctx, const uint32_t *rowptr, const int32_t *col, int64_t nv, const int32_t *vold, int32_t *eu, int32_t *ev)
{
    hipStream_t s = ctx->stream;
    DevBufs bufs(ctx);
    uint32_t *d_dk = nullptr, *d_dv = nullptr, *d_fu = nullptr, *d_uc = nullptr, *d_ebase = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_dk, (size_t)nv)); KOMB_HIP(ctx, bufs.alloc(&d_dv, (size_t)nv));
    KOMB_HIP(ctx, bufs.alloc(&d_fu, (size_t)nv + 1)); KOMB_HIP(ctx, bufs.alloc(&d_uc, (size_t)nv + 1)); KOMB_HIP(ctx, bufs.alloc(&d_ebase, (size_t)nv + 1));
    k_prep_vertex<<<grid_for(nv + 1), kBlock, 0, s>>>(rowptr, col, nv, d_dk, d_dv, d_fu, d_uc);
    KOMB_TRY(prim_exclusive_sum_u32(ctx, d_uc, d_ebase, nv + 1));
    if (nv > 0) k_edge_list<<<grid_for(nv, kPW, 256 * 8), kBlock, 0, s>>>(rowptr, col, nv, d_fu, d_ebase, vold, eu, ev);
    KOMB_HIP(ctx, hipStreamSynchronize(s));                          // (the scratch goes back to the pool when this returns)
    return KOMB_OK;
}

} // namespace komb
