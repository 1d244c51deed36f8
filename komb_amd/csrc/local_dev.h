// local_dev.h -- the "local" finish of a peel: an h-index fixed point on the compacted remainder.
//
// A level-synchronous peel spends most of its steps where almost nothing is left: at |E| = 100M the
// truss peel removes 99 % of the edges in four levels, then needs ~450 more dependent sub-rounds for
// the last 1 %; the k-core peel is the same story with hub vertices.  Each of those steps is a chain of
// dependent trips to memory however few units it peels (peel_dev.h).  Coreness and trussness are also the
// unique greatest fixed point of the h-index operator (Lu et al. 2016 for k-core; Sariyuce, Seshadhri,
// Pinar 2017 "Local algorithms for hierarchical dense subgraph discovery" for k-truss / nuclei):
//       tau(u) <- H{ value(item) : item in slice(u) }   (capped at tau(u)),
//       value = tau(neighbour) (k-core) / min(tau(x), tau(y)) over the triangle's other two edges (k-truss),
//       H(S)  = the largest h with at least h members of S that are >= h,
// started from the live keys (live degree / live support, upper bounds).  Every sweep is one plain
// data-parallel launch -- no frontier queues, no tie-breaks, no atomics except one counter -- and the
// number of sweeps is 15-45 where the peel needs hundreds of sub-rounds (scripts/sim/*.c measure both on
// the benchmark graphs).  So the peel engine keeps the bulk levels, where its steps are large and each item
// is visited once, and hands the remainder over (PeelCtrl::tail_limit):
//   1. the live units are renumbered 0..n-1, heavy ones (long live slices) first; their live keys become
//      tau and the slice lengths of the compact index;
//   2. the live items are collected into the compact index by the peel engine itself, run once over all
//      live units with a "collect" problem (its light batches / heavy chunks walk the original slices);
//   3. k_local_step sweeps until a sweep changes nothing; values only ever decrease, updates are in place;
//   4. the results are scattered back (coreness = tau, trussness = tau + 2).
// Why the result is exact whatever the interleaving: (a) tau stays >= the true level, because H is monotone
// and the true levels are a fixed point; (b) a unit is re-evaluated in the next sweep whenever one of its
// items' values may have dropped below its own value (the notification rule below is conservative under stale
// reads: a stale value is only ever too high); (c) every evaluation reads each item exactly once, so its count
// is a consistent sample; when a sweep changes nothing, every unit's last evaluation saw final values and found
// count(items >= tau) >= tau, which makes {tau >= k} a k-core / (k+2)-truss for every k, i.e. tau <= the true
// level.  Bit-exact parity with the peel is tested like every other path (tests/test_gpu_parity.py).
#pragma once

#include "peel_dev.h"

namespace komb {

constexpr int kLocBlock = 512;                 // 8 wave64 per workgroup, two workgroups per CU
constexpr int kLocWaves = kLocBlock / kWave;
constexpr int kLocHB = 4096;                   // histogram bins of the heavy path
constexpr int kLocBins = kLocHB / kLocBlock;   // bins per thread in the suffix search
constexpr int kCntWays = 8;                    // per-sweep counters are spread over this many 128-byte lines
constexpr int kCntWords = (5 * kCntWays + 5) * 32;   // 4 ring slots of change counters + 1 of evaluation counters, 3 queue heads, 2 list lengths
constexpr int kCntTimerWords = 6 * 12 * 2;     // -DKOMB_LOCAL_TIMERS: 6 sweeps x 12 64-bit stopwatch sums behind the counters
constexpr int kHvU = 8;                        // items per thread per trip on the workgroup path (independent load chains)
constexpr uint32_t kKeyBins = 4096;            // histogram of the live keys at hand-over (bound on the largest level)
constexpr uint32_t kMedMax = 2048;             // heavy units up to this many items are evaluated by one wavefront (values staged in LDS)
constexpr int kGiantWin = 8;                   // drops of at most this much are read off counters kept during the count: no second pass
constexpr uint32_t kGiantChunk = 8192;              // the longest units are counted in chunks of this many items, a workgroup each
constexpr int kLocBatch = 6;                   // launches queued between two looks at the control block


struct LocalGraph {                            // the compacted remainder
    uint32_t n, nh;          // units; ids [0, nh) are heavy (more than the problem's kItems items)
    uint32_t *off;           // [n+1] compact slice offsets
    int32_t *val;            // [n] tau
    int32_t *mark[2];        // [n] each: mark[k & 1][u] == k <=> u is evaluated in sweep k (written during sweep k-1 only)
    int32_t *gid;            // [n] unit id of the general engine
    uint32_t *len;           // [n+1] live key at hand-over = compact slice length (scan input)
    uint32_t *cur;           // [n] fill cursors of the collect pass
    uint4 *giant;            // [ng] the heavy units with more than kMedMax items: {id, first chunk, chunks, 0}
    uint32_t ng;
    uint4 *gchunk;           // [nchunk] {id, index in giant[], chunk of the unit, chunks of the unit}: the queue all workgroups share
    uint32_t nchunk;
    unsigned long long *gacc;    // [ng] (chunks arrived << 32) | items >= value so far, this sweep
    uint32_t *gwin;          // [ng * kGiantWin] items with value == the unit's value - 1 - b, b = 0 .. kGiantWin-1, so far, this sweep
    int4 *gnote;             // [ng] {new value, old value, sweep was full, sweep}: the notification a changed unit owes (k_local_giant_notify)
    uint32_t *khist;         // [kKeyBins] how many units have live key k (the last bin: k >= kKeyBins - 1)
    uint32_t *list;          // [n - nh] the light units marked for the sweep at hand, compacted by k_local_list just before it
    uint32_t flags;          // kLocUseList: the sweeps read that list (else they walk the ids and test the marks);
                             // kLocDeferNotify: a hub's notifications are written by k_local_giant_notify (else by the sweep itself)
};
constexpr uint32_t kLocUseList = 1u, kLocDeferNotify = 2u;
constexpr uint32_t kLocDeferChunks = 256;        // hub chunks from which the notification kernel is used

// ---- 1. numbering.  `list` (or all `units` when null) holds the candidates; live = alive marker in `marker`;
// the live key is key[u].  Heavy units get ids from 0 up, light ones from n_live-1 down (the host knows n_live
// = PeelCtrl::remaining).  Any order inside a class is fine: the fixed point is unique.
constexpr int kNumBlock = 1024;
static __global__ __launch_bounds__(kNumBlock) void k_local_number(const int32_t *__restrict__ list, uint32_t n_in, uint32_t n_live,
                                                         const int32_t *__restrict__ marker, const int32_t *__restrict__ key,
                                                         uint32_t light_max, int32_t *__restrict__ num, LocalGraph g, LocalCtrl *ctrl)
{
    // one reservation per workgroup and trip (two atomics for 1024 candidates): every live unit taking its id with an
    // atomic of its own queues 10^5..10^6 of them on two words
    __shared__ uint32_t sh_cnt[kNumBlock / kWave][2];
    __shared__ uint32_t sh_base[2];
    __shared__ uint32_t sh_kh[kKeyBins];
    __shared__ unsigned long long sh_ks;
    unsigned long long ks = 0ull;
    if (threadIdx.x == 0) sh_ks = 0ull;
    for (uint32_t i = threadIdx.x; i < kKeyBins; i += kNumBlock) sh_kh[i] = 0u;
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    for (uint32_t i0 = blockIdx.x * kNumBlock; i0 < n_in; i0 += gridDim.x * kNumBlock) {
        const uint32_t i = i0 + threadIdx.x;
        int32_t u = -1, k = 0;
        if (i < n_in) {
            u = list ? list[i] : (int32_t)i;
            if (!marker_alive(marker[u])) u = -1; else k = key[u];
        }
        const bool heavy = u >= 0 && (uint32_t)k > light_max, light = u >= 0 && !heavy;
        const uint64_t mh = __ballot(heavy), ml = __ballot(light);
        __syncthreads();
        if (lane == 0) { sh_cnt[w][0] = (uint32_t)__popcll(mh); sh_cnt[w][1] = (uint32_t)__popcll(ml); }
        __syncthreads();
        uint32_t bh = 0, bl = 0, th = 0, tl = 0;
#pragma unroll
        for (int x = 0; x < kNumBlock / kWave; ++x) {
            const uint32_t a = sh_cnt[x][0], b = sh_cnt[x][1];
            if (x < w) { bh += a; bl += b; }
            th += a; tl += b;
        }
        if (threadIdx.x == 0) {
            sh_base[0] = th ? atomicAdd(&ctrl->n_heavy, th) : 0u;
            sh_base[1] = tl ? atomicAdd(&ctrl->n_light, tl) : 0u;
        }
        __syncthreads();
        if (u < 0) continue;
        ks += (unsigned long long)(uint32_t)k;
        const uint32_t id = heavy ? sh_base[0] + bh + (uint32_t)__popcll(mh & lanemask_lt())
                                  : n_live - 1u - (sh_base[1] + bl + (uint32_t)__popcll(ml & lanemask_lt()));
        if (id >= n_live) { atomicAdd(&ctrl->bad, 1u); continue; }      // more live units than the control block said
        num[u] = (int32_t)id;
        g.gid[id] = u; g.len[id] = (uint32_t)k; g.val[id] = k; g.mark[1][id] = 1; g.mark[0][id] = 0; g.cur[id] = 0u;
        atomicAdd(&sh_kh[min((uint32_t)k, kKeyBins - 1u)], 1u);
        if ((uint32_t)k > kMedMax) {                                                     // a few hundred at most
            const uint32_t nch = ((uint32_t)k + kGiantChunk - 1u) / kGiantChunk;
            g.giant[atomicAdd(&ctrl->n_giant, 1u)] = make_uint4(id, atomicAdd(&ctrl->n_chunk, nch), nch, 0u);
        }
    }
    for (int o = 32; o > 0; o >>= 1) ks += __shfl_xor(ks, o);
    if (lane == 0 && ks) atomicAdd(&sh_ks, ks);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kKeyBins; i += kNumBlock) if (sh_kh[i]) atomicAdd(&g.khist[i], sh_kh[i]);
    if (threadIdx.x == 0 && sh_ks) atomicAdd(&ctrl->key_sum, sh_ks);
}

// the chunk queue of the longest units, one workgroup per unit
static __global__ __launch_bounds__(kBlock) void k_local_chunks(LocalGraph g)
{
    for (uint32_t gi = blockIdx.x; gi < g.ng; gi += gridDim.x) {
        const uint4 u = g.giant[gi];
        for (uint32_t j = threadIdx.x; j < u.z; j += kBlock) g.gchunk[u.y + j] = make_uint4(u.x, gi, j, u.z);
        if (threadIdx.x == 0) { g.gacc[gi] = 0ull; g.gnote[gi] = make_int4(0, 0, 0, 0); }
        if (threadIdx.x < (uint32_t)kGiantWin) g.gwin[gi * kGiantWin + threadIdx.x] = 0u;
    }
}

// Every level of the remainder is <= K, the h-index of its live keys (a level-t core / truss holds more than t units of key
// >= t): the values start at min(key, K), so a hub's first evaluations do not have to refine a range of 10^5.
static __global__ __launch_bounds__(kBlock) void k_local_clamp(LocalGraph g, int32_t K)
{
    for (uint32_t id = blockIdx.x * kBlock + threadIdx.x; id < g.n; id += gridDim.x * kBlock)
        if (g.val[id] > K) g.val[id] = K;
}

// Collect pass: position of this lane's entry in the compact slice of unit `id`.  Called by the lanes that have a live
// item (divergent); lanes of one unit share ONE cursor atomic (every item of a hub taking its slot with an atomic of its
// own queues them all on one word).
__device__ __forceinline__ uint32_t local_slot(uint32_t id, uint32_t *cur)
{
    const int lane = lane_id();
    uint64_t todo = __ballot(true);
    uint32_t pos = 0;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lid = (uint32_t)__shfl((int)id, leader);
        const uint64_t grp = __ballot(id == lid) & todo;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&cur[lid], (uint32_t)__popcll(grp));
        base = (uint32_t)__shfl((int)base, leader);
        if ((grp >> lane) & 1ull) pos = base + (uint32_t)__popcll(grp & lanemask_lt());
        todo &= ~grp;
    }
    return pos;
}

// after the collect pass: every compact slice must be exactly full
static __global__ __launch_bounds__(kBlock) void k_local_check(LocalGraph g, LocalCtrl *ctrl)
{
    uint32_t bad = 0;
    for (uint32_t id = blockIdx.x * kBlock + threadIdx.x; id < g.n; id += gridDim.x * kBlock)
        bad += (g.cur[id] != g.off[id + 1] - g.off[id]) ? 1u : 0u;
    bad = wave_sum(bad);
    if (lane_id() == 0 && bad) atomicAdd(&ctrl->bad, bad);
    if (blockIdx.x == 0 && threadIdx.x == 0 && (ctrl->n_heavy != g.nh || ctrl->n_heavy + ctrl->n_light != g.n)) atomicAdd(&ctrl->bad, 1u);
}

// ---- the light units marked for sweep k, as a list (launched before every sweep).  A sweep that looks at the marks group by
// group costs a batch's chain of dependent loads for every group with ONE marked unit in it -- a sweep with 10% of the
// units marked took as long as a full one; from the list the cost follows the number of marked units and the wavefronts
// share them evenly.  One reservation per workgroup; cnt[(5 * kCntWays + 3 + (k & 1)) * 32] = the list's length (zeroed
// by the sweep before).
constexpr int kListBlock = 1024;
static __global__ __launch_bounds__(kListBlock) void k_local_list(const LocalCtrl *ctrl, uint32_t *cnt, LocalGraph g, int32_t k)
{
    if (ctrl->done) return;                              // (launches queued behind the fixed point)
    __shared__ uint32_t sh_n[kListBlock / kWave];
    __shared__ uint32_t sh_base;
    const int32_t *mark_cur = g.mark[k & 1];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    uint32_t *tail = cnt + (5 * kCntWays + 3 + (k & 1)) * 32;
    const uint32_t nlight = g.n - g.nh;
    for (uint32_t i0 = blockIdx.x * kListBlock; i0 < nlight; i0 += gridDim.x * kListBlock) {
        const uint32_t i = i0 + threadIdx.x;
        const bool m = i < nlight && mark_cur[g.nh + i] == k;
        const uint64_t bm = __ballot(m);
        __syncthreads();
        if (lane == 0) sh_n[w] = (uint32_t)__popcll(bm);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int x = 0; x < kListBlock / kWave; ++x) { const uint32_t c = sh_n[x]; if (x < w) before += c; total += c; }
        if (threadIdx.x == 0) sh_base = total ? atomicAdd(tail, total) : 0u;
        __syncthreads();
        if (m) g.list[sh_base + before + (uint32_t)__popcll(bm & lanemask_lt())] = g.nh + i;
    }
}

// ---- 3. one sweep.  Problem concept (all __device__):
//   static constexpr int kU;                     values per lane of a light batch (light unit: <= 64 * kU items)
//   static constexpr int kGroups;                light groups per wavefront aimed at (see the light path)
//   static constexpr int kN;                     units an item refers to (k-core: the neighbour; k-truss: the other two edges)
//   void ids(pos, uint32_t (&id)[kN]) const;     the item's unit ids
// The value of an item is the smallest current value of its units.  When a unit drops from a to h in sweep k it marks
// for sweep k+1 every unit w of its items whose count this can lower, i.e. h < val[w] <= a -- or just h < val[w] when w
// is itself evaluated in this sweep (mark_cur[w] == k): only then can the value we see for it be stale (too high).
template <class P>
__device__ __forceinline__ int32_t local_value(const P &p, uint32_t pos, const int32_t *val)
{
    uint32_t id[P::kN];
    p.ids(pos, id);
    int32_t v = val[id[0]];
#pragma unroll
    for (int i = 1; i < P::kN; ++i) v = min(v, val[id[i]]);
    return v;
}
// the marks one item owes, loads first (so that several items' loads are in flight together), stores after
template <class P>
struct LocalNotify {
    uint32_t id[P::kN];
    bool hit[P::kN];
    // all: every unit is evaluated in this sweep (a full sweep: the marks say nothing)
    __device__ __forceinline__ void load(const P &p, uint32_t pos, const int32_t *val, const int32_t *mark_cur, int32_t h, int32_t a, int32_t k, bool active, bool all)
    {
#pragma unroll
        for (int i = 0; i < P::kN; ++i) hit[i] = false;
        if (!active) return;
        p.ids(pos, id);
        int32_t v[P::kN];
#pragma unroll
        for (int i = 0; i < P::kN; ++i) v[i] = val[id[i]];
#pragma unroll
        for (int i = 0; i < P::kN; ++i) hit[i] = v[i] > h && (all || v[i] <= a || mark_cur[id[i]] == k);
    }
    // kN == 1 and the item's value v0 as the evaluation read it: no second gather of the value, and the mark is looked at
    // only where v0 > a.  v0 may be stale (too high) -- then its unit is being evaluated in this sweep, which the mark says
    __device__ __forceinline__ void load_known(const P &p, uint32_t pos, int32_t v0, const int32_t *mark_cur, int32_t h, int32_t a, int32_t k, bool active, bool all)
    {
        static_assert(P::kN == 1, "one unit per item");
        hit[0] = false;
        if (!active || v0 <= h) return;
        p.ids(pos, id);
        hit[0] = all || v0 <= a || mark_cur[id[0]] == k;
    }
    __device__ __forceinline__ void store(int32_t *mark_next, int32_t k) const
    {
#pragma unroll
        for (int i = 0; i < P::kN; ++i) if (hit[i]) mark_next[id[i]] = k + 1;
    }
};

template <class P>
__global__ __launch_bounds__(kLocBlock) void k_local_step(LocalCtrl *ctrl, uint32_t *cnt, LocalGraph g, P p, int32_t k, uint32_t full_thr)
{
    // cnt: per-sweep counters spread over kCntWays words on separate 128-byte lines (a single word would queue one
    // atomic per workgroup, ~88 per microsecond): cnt[(slot * kCntWays + way) * 32 + 0] = units changed in the sweep
    // using ring slot `slot` = sweep % 4 (slot 4 = evaluations, statistics).
    // full_thr: while a sweep changes at least this many units, the next one does not notify (notifying ~ every unit costs
    // more than evaluating every unit once more) and the one after it is a FULL sweep: every unit is evaluated, marks are
    // not read.  Sweep k knows changed(k-1) and changed(k-2) at entry: it skips its notifications iff changed(k-1) >=
    // full_thr and is full iff sweep k-1 skipped them, i.e. changed(k-2) >= full_thr (changed(0) := n).
    constexpr int kU = P::kU;
    constexpr uint32_t kItems = (uint32_t)kWave * kU;
    __shared__ uint32_t sh_hist[kLocHB];
    __shared__ uint32_t sh_part[kLocWaves];
    __shared__ unsigned long long sh_best;
    __shared__ uint32_t sh_m[kLocBlock];
    __shared__ uint32_t sh_mn;
    __shared__ uint16_t sh_med[kLocWaves][kMedMax];
    __shared__ int32_t sh_i[2];
    __shared__ uint32_t sh_pick[4];
    __shared__ uint32_t sh_end[kLocWaves][kWave];
    __shared__ uint32_t sh_a[kLocWaves][4][kWave];
    const uint32_t tid = threadIdx.x;
    const int lane = lane_id(), w = (int)(tid >> 6);

    if (tid < (uint32_t)kCntWays) sh_part[tid] = (k > 1) ? cnt[(((k - 1) & 3) * kCntWays + tid) * 32] : 0u;
    else if (tid < 2u * kCntWays) sh_m[tid] = (k > 2) ? cnt[(((k - 2) & 3) * kCntWays + (tid - kCntWays)) * 32] : 0u;
    if (tid == 0) sh_i[0] = ctrl->done;
    __syncthreads();
    if (tid == 0) {
        uint32_t c1 = 0, c2 = 0;
        for (int i = 0; i < kCntWays; ++i) { c1 += sh_part[i]; c2 += sh_m[kCntWays + i]; }
        if (k == 1) c1 = g.n;
        if (k <= 2) c2 = g.n;
        sh_i[1] = c1 ? 1 : 0;
        sh_pick[0] = (c1 >= full_thr ? 1u : 0u) | ((k == 1 || c2 >= full_thr) ? 2u : 0u);
    }
    __syncthreads();
    if (sh_i[0]) return;
    if (sh_i[1] == 0) {                                  // the previous sweep changed nothing: fixed point
        if (blockIdx.x == 0 && tid == 0) { ctrl->done = 1; ctrl->iters = k - 1; }
        return;
    }
    const bool skip_notify = sh_pick[0] & 1u, full = sh_pick[0] & 2u;
    if (blockIdx.x == 0 && tid == 0) ctrl->spare[0] = (uint32_t)k;     // progress word: the sweep that is running (the host's guard reads it)
    __syncthreads();                                     // sh_pick and sh_m are reused below
    if (blockIdx.x == 0 && tid < (uint32_t)kCntWays) cnt[(((k + 1) & 3) * kCntWays + tid) * 32] = 0u;   // nobody reads or adds to that slot in this launch
    if (blockIdx.x == 0 && tid == 0) cnt[(5 * kCntWays + ((k + 1) % 3)) * 32] = 0u;                    // next sweep's queue of the longest units
    if (blockIdx.x == 0 && tid == 1) cnt[(5 * kCntWays + 3 + ((k + 1) & 1)) * 32] = 0u;                // ... and the length of its list of marked light units
    uint32_t n_changed = 0, n_evals = 0;                 // meaningful in thread 0 (heavy) / lane 0 of each wave (light)
#ifdef KOMB_LOCAL_TIMERS
    // per-wave stopwatch (100 MHz): [0] heavy-mark scan, [1] medium units, [2] workgroup units, [3] group setup (marks,
    // offsets), [4] item loads + first count, [5] search, [6] notification, [7] batches, [8] groups with a marked unit;
    // summed over the waves behind the counters, the slowest wave's total in [9], the number of waves in [10]
    unsigned long long tmv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
    const unsigned long long tstart = tlast;
    auto tick = [&](int i) { const unsigned long long now = wall_clock64(); tmv[i] += now - tlast; tlast = now; };
#define KOMB_LT(i) tick(i)
#else
#define KOMB_LT(i) do { } while (0)
#endif

    // ---- heavy units of medium length (<= kMedMax items): workgroup b owns the heavy ids b, b + G, b + 2G, ...; its threads
    // look at their marks side by side and queue the marked medium ones in LDS; one wavefront evaluates each, the item
    // values staged once in the wavefront's LDS buffer (16 bits each: values <= kMedMax).  No workgroup barrier follows, so
    // a wavefront without medium units goes straight on to the light ones.  (The longest units come last, below.)
    const int32_t *mark_cur = g.mark[k & 1];
    int32_t *mark_next = g.mark[(k + 1) & 1];
    for (uint32_t base = blockIdx.x; base < g.nh; base += gridDim.x * kLocBlock) {
        __syncthreads();
        if (tid == 0) sh_mn = 0u;
        __syncthreads();
        {
            const uint64_t hu64 = (uint64_t)base + (uint64_t)tid * gridDim.x;
            if (hu64 < g.nh && (full || mark_cur[hu64] == k) && g.off[hu64 + 1] - g.off[hu64] <= kMedMax) sh_m[atomicAdd(&sh_mn, 1u)] = (uint32_t)hu64;
        }
        __syncthreads();
        KOMB_LT(0);
        // medium units: wavefront w takes queue entries w, w + 8, ...
        const uint32_t nm = sh_mn;
        uint16_t *vb = sh_med[w];
#ifdef KOMB_LOCAL_EXP
        if (KOMB_LOCAL_EXP & 2) continue;
#endif
        for (uint32_t mi = (uint32_t)w; mi < nm; mi += kLocWaves) {
            const uint32_t hu = sh_m[mi];
            const int32_t cap = g.val[hu];                       // written by this wavefront only
            if (lane == 0) ++n_evals;
            if (cap <= 0) continue;
            const uint32_t beg = g.off[hu], len = g.off[hu + 1] - beg;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j0 = (uint32_t)lane; j0 < len; j0 += kWave * 8) {
                int32_t r[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) { const uint32_t j = j0 + (uint32_t)x * kWave; r[x] = j < len ? local_value(p, beg + j, g.val) : 0; }
#pragma unroll
                for (int x = 0; x < 8; ++x) { const uint32_t j = j0 + (uint32_t)x * kWave; if (j < len) vb[j] = (uint16_t)min(max(r[x], 0), 65535); }
            }
            __builtin_amdgcn_wave_barrier();
            auto count_ge = [&](uint32_t thr) -> uint32_t {
                uint32_t c = 0;
                for (uint32_t j = (uint32_t)lane; j < len; j += kWave) c += (uint32_t)vb[j] >= thr ? 1u : 0u;
                return wave_sum(c);
            };
            const uint32_t c0 = count_ge((uint32_t)cap);
            if (c0 >= (uint32_t)cap) continue;                   // unchanged
            uint32_t lo = c0, hi = (uint32_t)cap - 1u;           // H >= c0: the c0 items >= cap are >= c0 too
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1u) >> 1;
                if (count_ge(mid) >= mid) lo = mid; else hi = mid - 1u;
            }
            const int32_t H = (int32_t)lo;
            if (lane == 0) { g.val[hu] = H; ++n_changed; }
            if (!skip_notify) for (uint32_t j0 = (uint32_t)lane; j0 < len; j0 += kWave * 8) {
                LocalNotify<P> nt[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const uint32_t j = j0 + (uint32_t)x * kWave;
                    if constexpr (P::kN == 1) nt[x].load_known(p, beg + j, j < len ? (int32_t)vb[j] : 0, mark_cur, H, cap, k, j < len, full);
                    else nt[x].load(p, beg + j, g.val, mark_cur, H, cap, k, j < len, full);
                }
#pragma unroll
                for (int x = 0; x < 8; ++x) nt[x].store(mark_next, k);
            }
        }
        KOMB_LT(1);
    }

    // ---- light units: a wavefront takes a group of consecutive ids (64, fewer when there are fewer groups than
    // wavefronts), packs the marked ones, and evaluates them in batches of <= 64 * kU items flattened over the lanes.
    // Only the item VALUES stay in registers (and each item's owner, a byte): counts are ballots masked by each unit's
    // lane range.  The items of the few units that change are loaded again for the notification.
    uint32_t *s_end = sh_end[w];
    uint32_t (*s_a)[kWave] = sh_a[w];
    // (a full sweep walks all light ids, any other the list k_local_list has just made of the marked ones)
    const bool walk = full || !(g.flags & kLocUseList);
    const uint32_t nlight = walk ? g.n - g.nh : cnt[(5 * kCntWays + 3 + (k & 1)) * 32];
    const uint32_t nw = gridDim.x * kLocWaves;
    uint32_t gsz = kWave;
    // at least P::kGroups groups per wavefront while groups stay >= 8 units (measured at C3: k-truss units are short, its
    // sweep time fell 5.4 -> 3.2 ms going from 1 to 8; k-core's rose 3%)
    while (gsz > 1 && (nlight + gsz - 1) / gsz < (gsz > 8 ? (uint32_t)P::kGroups : 1u) * nw) gsz >>= 1;      // a few groups per wavefront: the slowest wavefront sets the sweep's time
    const uint32_t ngrp = (nlight + gsz - 1) / gsz;
    for (uint32_t grp = blockIdx.x * kLocWaves + (uint32_t)w; grp < ngrp; grp += nw) {
        const uint32_t li = grp * gsz + (uint32_t)lane;
        const uint32_t u = (uint32_t)lane < gsz && li < nlight ? (walk ? g.nh + li : g.list[li]) : 0u;
        const bool act = (uint32_t)lane < gsz && li < nlight && (full || !walk || mark_cur[u] == k);
        const uint64_t am = __ballot(act);
        if (!am) continue;
        const uint32_t na = (uint32_t)__popcll(am);
        __builtin_amdgcn_wave_barrier();
        if (act) {
            const uint32_t q = (uint32_t)__popcll(am & lanemask_lt());
            const uint32_t b0 = g.off[u];
            s_a[0][q] = u; s_a[1][q] = b0; s_a[2][q] = g.off[u + 1] - b0; s_a[3][q] = (uint32_t)g.val[u];
        }
        __builtin_amdgcn_wave_barrier();
        KOMB_LT(3);
#ifdef KOMB_LOCAL_TIMERS
        tmv[8] += 1;
#endif
        for (uint32_t start = 0; start < na;) {
            const bool has = start + (uint32_t)lane < na;
            uint32_t mu = 0, mlen = 0;
            int32_t mcap = 0;
            if (has) { mu = s_a[0][start + lane]; mlen = s_a[2][start + lane]; mcap = (int32_t)s_a[3][start + lane]; }
            const uint32_t incl = wave_incl_scan(mlen);
            uint32_t nb = (uint32_t)__popcll(__ballot(has && incl <= kItems));   // a prefix of the lanes
            if (nb == 0) {                               // a light unit longer than a batch: cannot happen (numbering rule)
                if (lane == 0) atomicAdd(&ctrl->bad, 1u);
                start += 1;
                continue;
            }
            const bool own = (uint32_t)lane < nb;
            const uint32_t total = (uint32_t)__shfl((int)incl, (int)nb - 1);
            const int32_t first = (int32_t)(incl - mlen), last = (int32_t)incl;  // my items: flattened positions [first, last)
            __builtin_amdgcn_wave_barrier();
            s_end[lane] = own ? incl : 0xFFFFFFFFu;
            __builtin_amdgcn_wave_barrier();
            int32_t r[kU];
            uint32_t tp[(kU + 3) / 4];                   // owner of item x: byte x of tp
#pragma unroll
            for (int x = 0; x < (kU + 3) / 4; ++x) tp[x] = 0u;
            auto owner_of = [&](int x) -> int { return (int)((tp[x >> 2] >> (8 * (x & 3))) & 0xFFu); };
            auto item_pos = [&](int x, int o) -> uint32_t {
                const uint32_t idx = (uint32_t)(x * kWave + lane);
                return s_a[1][start + (uint32_t)o] + (idx - (o ? s_end[o - 1] : 0u));
            };
#pragma unroll
            for (int x = 0; x < kU; ++x) {
                const uint32_t idx = (uint32_t)(x * kWave + lane);
                int o = 0;                               // owner: smallest t with s_end[t] > idx
#pragma unroll
                for (int st = kWave / 2; st > 0; st >>= 1) o += (s_end[o + st - 1] <= idx) ? st : 0;
                r[x] = -1;                               // an empty slot never counts
                if (idx < total) {
                    tp[x >> 2] |= (uint32_t)o << (8 * (x & 3));
                    r[x] = local_value(p, item_pos(x, o), g.val);
                }
            }
            // per-owner count of its items with value >= thr (thr differs per owner; thr >= 0)
            auto count_ge = [&](int32_t thr) -> uint32_t {
                uint32_t c = 0;
#pragma unroll
                for (int x = 0; x < kU; ++x) {
                    const int32_t th = __shfl(thr, owner_of(x));
                    const uint64_t B = __ballot(r[x] >= th);
                    int a = first - x * kWave, b = last - x * kWave;
                    a = a < 0 ? 0 : a; b = b > kWave ? kWave : b;
                    if (b > a) {
                        const uint64_t mb = (b == kWave) ? ~0ull : ((1ull << b) - 1ull);
                        c += (uint32_t)__popcll(B & mb & ~((1ull << a) - 1ull));
                    }
                }
                return c;
            };
            const uint32_t c0 = count_ge(mcap > 0 ? mcap : 0);
            KOMB_LT(4);
            const bool fail = own && mcap > 0 && c0 < (uint32_t)mcap;
            int32_t lo = fail ? (int32_t)c0 : mcap, hi = fail ? mcap - 1 : mcap;   // H >= c0: the c0 items >= cap are >= c0 too
            while (__ballot(lo < hi)) {
                const int32_t mid = (int32_t)(((int64_t)lo + hi + 1) >> 1);
                const uint32_t c = count_ge(mid > 0 ? mid : 0);
                if (lo < hi) { if (c >= (uint32_t)mid) lo = mid; else hi = mid - 1; }
            }
            const uint64_t fm = __ballot(fail);
            KOMB_LT(5);
            if (fm && fail) g.val[mu] = lo;
            if (fm && !skip_notify) {
                // the items of the units that dropped are loaded again (their ids were not kept: registers), all loads
                // first, then the marks
                const int32_t thr_n = fail ? lo : 0x7FFFFFFF;
                LocalNotify<P> nt[kU];
#pragma unroll
                for (int x = 0; x < kU; ++x) {
                    const int o = owner_of(x);
                    const int32_t th = __shfl(thr_n, o), old = __shfl(mcap, o);
                    if constexpr (P::kN == 1) nt[x].load_known(p, item_pos(x, o), r[x], mark_cur, th, old, k, r[x] >= 0 && th != 0x7FFFFFFF, full);
                    else nt[x].load(p, item_pos(x, o), g.val, mark_cur, th, old, k, r[x] >= 0 && th != 0x7FFFFFFF, full);
                }
#pragma unroll
                for (int x = 0; x < kU; ++x) nt[x].store(mark_next, k);
            }
            n_changed += (uint32_t)__popcll(fm);
            n_evals += nb;
            start += nb;
            __builtin_amdgcn_wave_barrier();
#ifdef KOMB_LOCAL_TIMERS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tick(6); tmv[7] += 1;
#endif
        }
    }
    // ---- the longest units (> kMedMax items), in chunks of kGiantChunk items taken from a queue all workgroups share, so
    // whoever is done with its light and medium units takes the next one and a hub of 10^5 items is counted by a dozen
    // workgroups side by side.  A plain count of the items >= cap first: every chunk adds its count and an arrival to the
    // unit's accumulator with ONE 64-bit atomic, and the workgroup that arrives last sees the total.  Most re-evaluations
    // end there (the unit still has cap items >= cap); otherwise that workgroup alone refines an LDS histogram of the item
    // values until the new value is exact, and notifies.
    if (g.nchunk) {
        uint32_t *gq = cnt + (5 * kCntWays + (k % 3)) * 32;
        for (;;) {
            __syncthreads();
            if (tid == 0) {
                const uint32_t i = atomicAdd(gq, 1u);
                int32_t pick = -1;                                   // queue exhausted
                if (i < g.nchunk) {
                    const uint4 c = g.gchunk[i];
                    pick = (full || mark_cur[c.x] == k) ? (int32_t)c.x : -2;
                    sh_pick[0] = c.y; sh_pick[1] = c.z; sh_pick[2] = c.w;
                }
                sh_i[0] = pick;
            }
            __syncthreads();
            const int32_t pick = sh_i[0];
            if (pick == -1) break;
            if (pick == -2) continue;
            const uint32_t hu = (uint32_t)pick;
            const uint32_t gi = sh_pick[0], cj = sh_pick[1], nch = sh_pick[2];
            const int32_t cap = g.val[hu];                       // written by the unit's last arrival only, after every chunk has read it
            if (cap <= 0) continue;
            const uint32_t beg = g.off[hu], len = g.off[hu + 1] - beg;
            int32_t lo = 0, hi = cap - 1, H = cap;
            {
                uint32_t ge = 0, wn[kGiantWin];
#pragma unroll
                for (int b = 0; b < kGiantWin; ++b) wn[b] = 0u;
                const uint32_t c_end = min(len, (cj + 1u) * kGiantChunk);
                for (uint32_t j0 = cj * kGiantChunk + tid; j0 < c_end; j0 += kHvU * kLocBlock) {      // kHvU independent load chains in flight
                    int32_t r[kHvU];
#pragma unroll
                    for (int x = 0; x < kHvU; ++x) { const uint32_t j = j0 + (uint32_t)x * kLocBlock; r[x] = j < c_end ? local_value(p, beg + j, g.val) : -1; }
#pragma unroll
                    for (int x = 0; x < kHvU; ++x) {
                        ge += r[x] >= cap ? 1u : 0u;
                        const uint32_t below = (uint32_t)(cap - 1 - r[x]);          // 0 .. kGiantWin-1: just under the unit's value
#pragma unroll
                        for (int b = 0; b < kGiantWin; ++b) wn[b] += below == (uint32_t)b ? 1u : 0u;
                    }
                }
                ge = wave_sum(ge);
#pragma unroll
                for (int b = 0; b < kGiantWin; ++b) wn[b] = wave_sum(wn[b]);
                __syncthreads();
                if (lane == 0) {
                    sh_part[w] = ge;
#pragma unroll
                    for (int b = 0; b < kGiantWin; ++b) sh_hist[w * kGiantWin + b] = wn[b];
                }
                __syncthreads();
                if (w == 0) {
                    // the window counters first, then -- once they are performed -- count and arrival in one 64-bit atomic
                    if (lane < kGiantWin) {
                        uint32_t t = 0;
                        for (int i = 0; i < kLocWaves; ++i) t += sh_hist[i * kGiantWin + lane];
                        if (t) atomicAdd(&g.gwin[gi * kGiantWin + (uint32_t)lane], t);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) {
                        uint32_t mine = 0;
                        for (int i = 0; i < kLocWaves; ++i) mine += sh_part[i];
                        const unsigned long long old = atomicAdd(&g.gacc[gi], (1ull << 32) | (unsigned long long)mine);
                        const bool last = (uint32_t)(old >> 32) == nch - 1u;
                        if (last) g.gacc[gi] = 0ull;             // every chunk of this sweep has arrived; the next use is the next launch
                        sh_pick[3] = last ? (uint32_t)old + mine : 0xFFFFFFFFu;
                    }
                }
                __syncthreads();
                const uint32_t c0 = sh_pick[3];
                if (c0 == 0xFFFFFFFFu) continue;                 // another workgroup finishes this unit
                if (tid == 0) ++n_evals;
                // every chunk has arrived: the window counters are complete; take them (and leave zeros for the next sweep)
                if (tid < (uint32_t)kGiantWin) sh_hist[tid] = atomicExch(&g.gwin[gi * kGiantWin + tid], 0u);
                __syncthreads();
#ifdef KOMB_LOCAL_EXP
                if (KOMB_LOCAL_EXP & 1) continue;
#endif
                if (c0 >= (uint32_t)cap) continue;               // still has cap items >= cap: unchanged
                lo = (int32_t)c0;                                // the c0 items >= cap are >= c0 as well
                // a small drop is read off the window: count(items >= cap-1-b) = c0 + win[0..b]
                {
                    uint32_t run = c0;
                    int32_t found = -1;
                    for (int b = 0; b < kGiantWin && cap - 1 - b >= 0; ++b) {
                        run += sh_hist[b];
                        if (run >= (uint32_t)(cap - 1 - b)) { found = cap - 1 - b; break; }
                    }
                    __syncthreads();                             // (sh_hist is reused by the refinement)
                    if (found >= 0) H = found;
                    else hi = cap - 1 - kGiantWin;               // below the window
                }
            }
            if (H == cap && hi < 0) H = 0;
            while (H == cap) {
                // histogram of the values in [lo, hi], `above` = values > hi, all from ONE pass over the items
                const uint32_t width = (uint32_t)(hi - lo) + 1u;
                int sh = 0;
                while (((width - 1u) >> sh) >= (uint32_t)kLocHB) ++sh;
                const uint32_t nb = ((width - 1u) >> sh) + 1u;
                __syncthreads();
                for (uint32_t i = tid; i < nb; i += kLocBlock) sh_hist[i] = 0u;
                if (tid == 0) sh_best = 0ull;
                __syncthreads();
                uint32_t ab = 0;
                for (uint32_t j0 = tid; j0 < len; j0 += kHvU * kLocBlock) {
                    int32_t r[kHvU];
#pragma unroll
                    for (int x = 0; x < kHvU; ++x) { const uint32_t j = j0 + (uint32_t)x * kLocBlock; r[x] = j < len ? local_value(p, beg + j, g.val) : -1; }
#pragma unroll
                    for (int x = 0; x < kHvU; ++x) {
                        if (r[x] > hi) ++ab;
                        else if (r[x] >= lo) atomicAdd(&sh_hist[(uint32_t)(r[x] - lo) >> sh], 1u);
                    }
                }
                ab = wave_sum(ab);
                if (lane == 0) sh_part[w] = ab;
                __syncthreads();
                uint32_t above = 0;
#pragma unroll
                for (int i = 0; i < kLocWaves; ++i) above += sh_part[i];
                // largest bin b with count(values >= lo + (b << sh)) >= lo + (b << sh); thread t owns kLocBins consecutive bins
                uint32_t h4[kLocBins], mine = 0;
#pragma unroll
                for (int i = 0; i < kLocBins; ++i) { const uint32_t bi = kLocBins * tid + (uint32_t)i; h4[i] = bi < nb ? sh_hist[bi] : 0u; mine += h4[i]; }
                uint32_t suf = mine;                                         // suffix sum over the lanes above, inclusive
                for (int o = 1; o < kWave; o <<= 1) { const uint32_t t = (uint32_t)__shfl_down((int)suf, o); if (lane + o < kWave) suf += t; }
                __syncthreads();                                             // sh_part is reused
                if (lane == 0) sh_part[w] = suf;                             // wave total
                __syncthreads();
                uint32_t run = above + suf - mine;                           // values in the bins above this thread's
                for (int i = w + 1; i < kLocWaves; ++i) run += sh_part[i];
                unsigned long long best = 0ull;
#pragma unroll
                for (int i = kLocBins - 1; i >= 0; --i) {
                    const uint32_t b = kLocBins * tid + (uint32_t)i;
                    run += h4[i];                                            // count(values >= start of bin b)
                    if (b < nb && best == 0ull && (unsigned long long)run >= (unsigned long long)lo + ((unsigned long long)b << sh))
                        best = (unsigned long long)b + 1ull;
                }
                if (best) atomicMax(&sh_best, best);
                __syncthreads();
                const unsigned long long bb = sh_best;
                if (bb == 0ull) {                                            // values dropped under the range while we looked
                    if (lo == 0) { H = 0; break; }
                    hi = lo - 1; lo = 0;
                    continue;
                }
                const uint32_t b = (uint32_t)bb - 1u;
                const int32_t nlo = lo + (int32_t)(b << sh);
                const int32_t nhi = min(hi, nlo + (int32_t)((1u << sh) - 1u));
                lo = nlo; hi = nhi;
                if (sh == 0) { H = lo; break; }
            }
            if (H < cap) {
                if (tid == 0) {
                    g.val[hu] = H; ++n_changed;
                    // the marks this drop owes: written by k_local_giant_notify, chunk by chunk, right after this launch ...
                    if (!skip_notify && (g.flags & kLocDeferNotify)) g.gnote[gi] = make_int4(H, cap, full ? 1 : 0, k);
                }
                // ... or, in a small remainder (where a third launch per sweep costs more than it saves), here
                if (!skip_notify && !(g.flags & kLocDeferNotify)) for (uint32_t j0 = tid; j0 < len; j0 += kHvU * kLocBlock) {
                    LocalNotify<P> nt[kHvU];
#pragma unroll
                    for (int x = 0; x < kHvU; ++x) { const uint32_t j = j0 + (uint32_t)x * kLocBlock; nt[x].load(p, beg + j, g.val, mark_cur, H, cap, k, j < len, full); }
#pragma unroll
                    for (int x = 0; x < kHvU; ++x) nt[x].store(mark_next, k);
                }
            }
        }
    }
    KOMB_LT(2);
#ifdef KOMB_LOCAL_TIMERS
    if (lane == 0 && (k == 1 || k == 4 || k == 8 || k == 12 || k == 16 || k == 20)) {
        const int slot = k == 1 ? 0 : k / 4;
        unsigned long long *tm = reinterpret_cast<unsigned long long *>(cnt + kCntWords) + slot * 12;
        for (int i = 0; i < 9; ++i) atomicAdd(&tm[i], tmv[i]);
        atomicMax(&tm[9], wall_clock64() - tstart);
        atomicAdd(&tm[10], 1ull);
    }
#endif
    // one add per workgroup, spread over kCntWays words
    __syncthreads();
    if (lane == 0) { sh_end[w][0] = n_changed; sh_end[w][1] = n_evals; }
    __syncthreads();
    if (tid == 0) {
        uint32_t c = 0, e = 0;
        for (int i = 0; i < kLocWaves; ++i) { c += sh_end[i][0]; e += sh_end[i][1]; }
        uint32_t *mine = cnt + ((k & 3) * kCntWays + (blockIdx.x % kCntWays)) * 32;
        if (c) atomicAdd(mine, c);
        if (e) atomicAdd(cnt + (4 * kCntWays + (blockIdx.x % kCntWays)) * 32, e);
    }
}

// ---- the notifications of the longest units that dropped in sweep k (launched after it when there are such units): a
// workgroup per chunk, so a hub of 10^5 items marks its neighbours in microseconds instead of one workgroup's 50.
template <class P>
__global__ __launch_bounds__(kLocBlock) void k_local_giant_notify(const LocalCtrl *ctrl, LocalGraph g, P p, int32_t k)
{
    if (ctrl->done) return;
    const int32_t *mark_cur = g.mark[k & 1];
    int32_t *mark_next = g.mark[(k + 1) & 1];
    for (uint32_t c = blockIdx.x; c < g.nchunk; c += gridDim.x) {
        const uint4 ch = g.gchunk[c];                            // {id, index in giant[], chunk, chunks}
        const int4 nt4 = g.gnote[ch.y];
        if (nt4.w != k) continue;                                // (uniform: every thread read the same words)
        const uint32_t beg = g.off[ch.x], len = g.off[ch.x + 1] - beg;
        const uint32_t c_end = min(len, (ch.z + 1u) * kGiantChunk);
        for (uint32_t j0 = ch.z * kGiantChunk + threadIdx.x; j0 < c_end; j0 += kHvU * kLocBlock) {
            LocalNotify<P> nt[kHvU];
#pragma unroll
            for (int x = 0; x < kHvU; ++x) { const uint32_t j = j0 + (uint32_t)x * kLocBlock; nt[x].load(p, beg + j, g.val, mark_cur, nt4.x, nt4.y, k, j < c_end, nt4.z != 0); }
#pragma unroll
            for (int x = 0; x < kHvU; ++x) nt[x].store(mark_next, k);
        }
    }
}

// ---- 4. results back to the general engine's arrays (+ statistics: largest value, which values occur)
constexpr uint32_t kFinWords = 2048;               // values below 65536 go through a bitmap in LDS
static __global__ __launch_bounds__(kBlock) void k_local_finish(LocalGraph g, int32_t add, int32_t *__restrict__ out,
                                                         uint32_t *present, LocalCtrl *ctrl)
{
    // which values occur: 10^5..10^6 units share a few dozen values, so the bits are collected per workgroup in LDS
    // and every workgroup ORs its non-zero words into the global bitmap once
    __shared__ uint32_t sh_bits[kFinWords];
    for (uint32_t i = threadIdx.x; i < kFinWords; i += kBlock) sh_bits[i] = 0u;
    __syncthreads();
    int32_t mx = 0;
    for (uint32_t id = blockIdx.x * kBlock + threadIdx.x; id < g.n; id += gridDim.x * kBlock) {
        const int32_t v = g.val[id];
        out[g.gid[id]] = v + add;
        mx = max(mx, v);
        const uint32_t wd = (uint32_t)v >> 5, bit = 1u << ((uint32_t)v & 31u);
        if (wd < kFinWords) { if (!(sh_bits[wd] & bit)) atomicOr(&sh_bits[wd], bit); }
        else atomicOr(&present[wd], bit);
    }
    __shared__ int32_t sh_mx;
    if (threadIdx.x == 0) sh_mx = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kFinWords; i += kBlock) if (sh_bits[i]) atomicOr(&present[i], sh_bits[i]);
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    if (lane_id() == 0 && mx) atomicMax(&sh_mx, mx);
    __syncthreads();
    if (threadIdx.x == 0 && sh_mx) atomicMax(&ctrl->max_val, sh_mx);           // one per workgroup
}
static __global__ __launch_bounds__(kBlock) void k_local_levels(const uint32_t *__restrict__ present, uint32_t words, const uint32_t *__restrict__ cnt, LocalCtrl *ctrl)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { uint32_t e = 0; for (int i = 0; i < kCntWays; ++i) e += cnt[(4 * kCntWays + i) * 32]; ctrl->evals = e; }
    uint32_t c = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < words; i += gridDim.x * kBlock) c += (uint32_t)__popc(present[i]);
    c = wave_sum(c);
    if (lane_id() == 0 && c) atomicAdd(&ctrl->levels, c);
}

// Sweep until the control block reports the fixed point.  Like drive_peel: the host keeps a batch of launches
// queued and looks at a copy of the control block one batch behind.
template <class P>
int local_fixpoint(komb_ctx *ctx, LocalCtrl *d_ctrl, uint32_t *d_cnt, const LocalGraph &g, const P &p, uint64_t total_items, int *launches_out)
{
    hipStream_t s = ctx->stream;
    const int grid = 512;                                 // two 512-thread workgroups per CU (<= 96 VGPRs)
    const uint32_t nlight_all = g.n - g.nh;
    const int list_grid = (int)std::min<uint32_t>(256u, (nlight_all + kListBlock - 1) / kListBlock ? (nlight_all + kListBlock - 1) / kListBlock : 1u);
    EventSet evs;
    hipEvent_t ev[2] = {nullptr, nullptr};
    KOMB_HIP(ctx, evs.make(&ev[0], hipEventDisableTiming));
    KOMB_HIP(ctx, evs.make(&ev[1], hipEventDisableTiming));
    LocalCtrl *h = ctx->h_local;
    // a sweep that changes anything lowers the sum of the values (<= total_items at the start) by at least 1, so this
    // many launches cannot be reached; the loop ends on `done`
    const uint64_t max_launches = total_items + (uint64_t)g.n + 64u;
    int launches = 0, slot = 0, status = KOMB_OK;
    bool have_prev = false, finished = false, stuck = false;
    int32_t k = 0, seen_iters = -1;
    // option LOCAL_DEBUG=2: an event after every launch, the per-sweep times on stderr
    // notifications are skipped (and the next sweep is full) while a sweep changes at least an eighth of the units (measured at C3: n/2 .. n/16 within 4%, never = +20%)
    uint32_t full_thr = g.n / 8u + 1u;
#ifdef KOMB_DEBUG_SWITCHES
    if (const char *e = getenv("KOMB_LOCAL_FULL")) { const long d = atol(e); full_thr = d > 0 ? g.n / (uint32_t)d + 1u : 0xFFFFFFFFu; }   // n / d; 0 = never
#endif
    const char *dbg_env = ctx_opt(ctx, "LOCAL_DEBUG");
    const bool per_sweep = dbg_env && atoi(dbg_env) >= 2;
    std::vector<hipEvent_t> sw;
    if (per_sweep) { sw.resize(1); if (evs.make(&sw[0]) == hipSuccess) (void)hipEventRecord(sw[0], s); }
    while (!finished && (uint64_t)launches < max_launches && k < 0x3FFFFFF0) {
        for (int i = 0; i < kLocBatch; ++i) {
            ++k;
            if (g.flags & kLocUseList) k_local_list<<<list_grid, kListBlock, 0, s>>>(d_ctrl, d_cnt, g, k);
            k_local_step<P><<<grid, kLocBlock, 0, s>>>(d_ctrl, d_cnt, g, p, k, full_thr); ++launches;
            if (g.nchunk && (g.flags & kLocDeferNotify)) k_local_giant_notify<P><<<(int)std::min<uint32_t>(g.nchunk, 1024u), kLocBlock, 0, s>>>(d_ctrl, g, p, k);
            if (per_sweep && sw.size() < 600) { hipEvent_t e2 = nullptr; if (evs.make(&e2) == hipSuccess) { (void)hipEventRecord(e2, s); sw.push_back(e2); } }
        }
        if (hipMemcpyAsync(&h[slot], d_ctrl, sizeof(LocalCtrl), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipEventRecord(ev[slot], s) != hipSuccess) { status = KOMB_ERR_DEVICE; break; }
        if (have_prev) {
            if (hipEventSynchronize(ev[slot ^ 1]) != hipSuccess) { status = KOMB_ERR_DEVICE; break; }
            // a batch of sweeps that neither finished nor counted a single sweep: the kernels are not running the fixed
            // point any more (the same guard as drive_peel's sequence check) -- fail instead of queueing launches for hours
            if (h[slot ^ 1].done) finished = true;
            else if ((int32_t)h[slot ^ 1].spare[0] == seen_iters) { stuck = true; finished = true; }
            seen_iters = (int32_t)h[slot ^ 1].spare[0];
        }
        have_prev = true;
        slot ^= 1;
    }
    hipError_t e = hipStreamSynchronize(s);
    if (per_sweep) {
        fprintf(stderr, "komb local sweeps (us):");
        for (size_t i = 1; i < sw.size(); ++i) { float ms = 0.f; (void)hipEventElapsedTime(&ms, sw[i - 1], sw[i]); fprintf(stderr, " %.0f", ms * 1e3f); }
        fprintf(stderr, "\n");
    }
    if (launches_out) *launches_out = launches;
    if (status != KOMB_OK || e != hipSuccess)
        KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "local fixed point: HIP failure (%s)", hipGetErrorString(e));
    if (stuck) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "local fixed point: the sweeps make no progress (sweep %d)", (int)seen_iters);
    return KOMB_OK;
}

// one launch instead of five memsets at the hand-over (each is ~15 us of host time while the GPU waits)
static __global__ __launch_bounds__(kBlock) void k_local_zero(uint32_t *a, uint32_t na, uint32_t *b, uint32_t nb, uint32_t *c, uint32_t nc,
                                                       uint32_t *d, uint32_t nd, uint32_t *e, uint32_t ne)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x, step = gridDim.x * kBlock;
    for (uint32_t i = t; i < na; i += step) a[i] = 0u;
    for (uint32_t i = t; i < nb; i += step) b[i] = 0u;
    for (uint32_t i = t; i < nc; i += step) c[i] = 0u;
    for (uint32_t i = t; i < nd; i += step) d[i] = 0u;
    for (uint32_t i = t; i < ne; i += step) e[i] = 0u;
}

struct LocalStats {
    uint32_t units = 0, heavy = 0, levels = 0;
    int32_t max_val = 0;
    int sweeps = 0, launches = 0;
    uint64_t items = 0, evals = 0;
    bool refused = false;            // the remainder has more items than the caller's limit: nothing was done,
    uint32_t new_limit = 0;          // ... the peel's control block now says done = 0, tail_limit = new_limit
};

// The whole hand-over: number -> scan -> collect (peel engine, `launch_collect`) -> check -> sweep (`run_fix`) ->
// scatter.  hc = the peel's control block as the host last read it (done == 3, or the initial state of a small
// input); marker / key = the peel's alive markers and live keys; `out` receives value + add for every unit of
// the remainder.  launch_collect(g, num, items, d_cctrl, index) issues ONE launch of k_peel_step<Collect>;
// run_fix(g, items, total, d_lctrl, d_cnt, &launches) = local_fixpoint<Local>; after_number(g) runs once the ids exist
// (k-core builds its bitmap of live vertices there).
template <class LaunchCollect, class RunFix, class AfterNumber>
int local_finish(komb_ctx *ctx, DevBufs &bufs, const PeelCtrl &hc, PeelCtrl *d_ctrl, uint32_t units, const int32_t *marker,
                 const int32_t *key, const int32_t *live_list, uint32_t light_max, size_t item_bytes, uint64_t item_limit, uint32_t max_density, bool use_list, int32_t add, int32_t *out,
                 LaunchCollect &&launch_collect, RunFix &&run_fix, LocalStats *ls, AfterNumber &&after_number)
{
    hipStream_t s = ctx->stream;
    const uint32_t n = hc.remaining;
    if (n == 0) KOMB_FAIL(ctx, KOMB_ERR_STATE, "local finish: nothing left to hand over");
    // option LOCAL_DEBUG=1: one stderr line per hand-over with the phase times (HIP events)
    const bool dbg = ctx_opt(ctx, "LOCAL_DEBUG") != nullptr;
    EventSet evs;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (dbg) for (auto &e : ev) (void)evs.make(&e);
    auto stamp = [&](int i) { if (dbg) (void)hipEventRecord(ev[i], s); };
    stamp(0);
    const int32_t *list = hc.live_mode ? live_list : nullptr;
    const uint32_t n_in = hc.live_mode ? hc.live_count : units;
    LocalGraph g{};
    int32_t *d_num = nullptr;
    LocalCtrl *d_lctrl = nullptr;
    PeelCtrl *d_cctrl = nullptr;
    uint32_t *d_present = nullptr;
    const size_t present_words = ((size_t)units >> 5) + 2 > kFinWords ? ((size_t)units >> 5) + 2 : (size_t)kFinWords;
    KOMB_HIP(ctx, bufs.alloc(&d_num, (size_t)units));
    KOMB_HIP(ctx, bufs.alloc(&g.off, (size_t)n + 1));
    KOMB_HIP(ctx, bufs.alloc(&g.len, (size_t)n + 1));
    KOMB_HIP(ctx, bufs.alloc(&g.val, (size_t)n));
    KOMB_HIP(ctx, bufs.alloc(&g.mark[0], (size_t)n));
    KOMB_HIP(ctx, bufs.alloc(&g.mark[1], (size_t)n));
    KOMB_HIP(ctx, bufs.alloc(&g.gid, (size_t)n));
    KOMB_HIP(ctx, bufs.alloc(&g.cur, (size_t)n));
    KOMB_HIP(ctx, bufs.alloc(&g.list, (size_t)n));
    KOMB_HIP(ctx, bufs.alloc(&g.giant, (size_t)n));                // at most n entries of 16 bytes; the pool hands out what is asked for
    KOMB_HIP(ctx, bufs.alloc(&g.khist, (size_t)kKeyBins));
    KOMB_HIP(ctx, bufs.alloc(&d_lctrl, 1));
    uint32_t *d_cnt = nullptr;
    KOMB_HIP(ctx, bufs.alloc(&d_cnt, (size_t)kCntWords + kCntTimerWords));
    KOMB_HIP(ctx, bufs.alloc(&d_cctrl, 1));
    KOMB_HIP(ctx, bufs.alloc(&d_present, present_words));
    static_assert(sizeof(LocalCtrl) % sizeof(uint32_t) == 0, "LocalCtrl is zeroed word by word");
    if (present_words > 0xFFFFFFFFull) KOMB_FAIL(ctx, KOMB_ERR_LIMIT, "local finish: %zu presence words", present_words);
    k_local_zero<<<64, kBlock, 0, s>>>(g.khist, kKeyBins, d_cnt, (uint32_t)(kCntWords + kCntTimerWords), (uint32_t *)d_lctrl, (uint32_t)(sizeof(LocalCtrl) / sizeof(uint32_t)),
                                       g.len + n, 1u, d_present, (uint32_t)present_words);
    void *d_items = nullptr;
    auto release_all = [&]() {
        bufs.release(d_items); bufs.release(d_num); bufs.release(g.off); bufs.release(g.len); bufs.release(g.val);
        bufs.release(g.mark[0]); bufs.release(g.mark[1]); bufs.release(g.gid); bufs.release(g.list); bufs.release(g.giant); bufs.release(g.gchunk); bufs.release(g.gacc); bufs.release(g.gwin); bufs.release(g.gnote);
        bufs.release(g.khist); bufs.release(g.cur); bufs.release(d_lctrl); bufs.release(d_cnt); bufs.release(d_cctrl); bufs.release(d_present);
    };
    g.n = n; g.nh = 0; g.ng = 0; g.nchunk = 0;
    int64_t gb = ((int64_t)n_in + kNumBlock - 1) / kNumBlock;
    k_local_number<<<(int)(gb < 1 ? 1 : (gb > 1024 ? 1024 : gb)), kNumBlock, 0, s>>>(list, n_in, n, marker, key, light_max, d_num, g, d_lctrl);
    // one round trip for everything the host decides on: the numbering's counters and the histogram of the live keys
    LocalCtrl hl{};
    std::vector<uint32_t> kh(kKeyBins);
    if (ctx->h_stage && sizeof(LocalCtrl) + kKeyBins * sizeof(uint32_t) <= kStageBytes) {
        char *st8 = (char *)ctx->h_stage;
        KOMB_HIP(ctx, hipMemcpyAsync(st8, d_lctrl, sizeof(LocalCtrl), hipMemcpyDeviceToHost, s));
        KOMB_HIP(ctx, hipMemcpyAsync(st8 + sizeof(LocalCtrl), g.khist, kKeyBins * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        KOMB_HIP(ctx, hipStreamSynchronize(s));
        memcpy(&hl, st8, sizeof(LocalCtrl));
        memcpy(kh.data(), st8 + sizeof(LocalCtrl), kKeyBins * sizeof(uint32_t));
    } else {
        KOMB_HIP(ctx, d2h(ctx, &hl, d_lctrl, sizeof(LocalCtrl)));
        KOMB_HIP(ctx, d2h(ctx, kh.data(), g.khist, kKeyBins * sizeof(uint32_t)));
    }
    if (hl.bad || hl.n_heavy + hl.n_light != n)
        KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "local finish: %u live units numbered, the peel counted %u", hl.n_heavy + hl.n_light, n);
    // max_density (0 = no rule): a remainder with more items per unit than this is left to the peel for good -- it only gets
    // denser as the peel goes on, and sweeping a dense remainder costs more than peeling it (k-truss, alpha = 2.1 shape: 280
    // triangles per edge, 6.3 ms in the fixed point for what the peel does in 2)
    const bool too_dense = max_density && hl.key_sum > (unsigned long long)max_density * n;
    if (hl.key_sum > item_limit || hl.key_sum >= 0xFFFFFFFFull || too_dense) {
        // too many items for the fixed point to pay (or for 32-bit slice offsets): the peel goes on, and offers the remainder
        // again once it is small enough that -- at this density or twice it -- its items fit
        const uint64_t lim = item_limit < 0xFFFFFFFFull ? item_limit : 0xFFFFFFFFull;
        uint64_t nl = (uint64_t)((double)n * ((double)lim / (double)hl.key_sum) * 0.5);
        if (nl >= n) nl = n / 2;
        if (too_dense) nl = 0;
        const uint32_t patch[2] = {(uint32_t)nl, 0u};
        KOMB_HIP(ctx, hipMemcpy(&d_ctrl->tail_limit, &patch[0], sizeof(uint32_t), hipMemcpyHostToDevice));
        KOMB_HIP(ctx, hipMemcpy(&d_ctrl->done, &patch[1], sizeof(uint32_t), hipMemcpyHostToDevice));
        if (dbg) fprintf(stderr, "komb local finish: refused %u units with %llu items (limit %llu); next offer at <= %u units\n", n, (unsigned long long)hl.key_sum, (unsigned long long)item_limit, (uint32_t)nl);
        if (ls) { ls->refused = true; ls->new_limit = (uint32_t)nl; }
        release_all();
        return KOMB_OK;
    }
    KOMB_TRY(prim_exclusive_sum_u32(ctx, g.len, g.off, (int64_t)n + 1));
    // (the offsets end at the sum of the live keys k_local_number has just reported -- the slice lengths ARE those keys --
    // so the total needs no round trip of its own; k_local_check compares every slice's fill with its length afterwards)
    const uint32_t total = (uint32_t)hl.key_sum;
    g.nh = hl.n_heavy;
    g.ng = hl.n_giant;
    g.nchunk = hl.n_chunk;
    // The list and the hubs' notification kernel are one more launch per sweep each (~5 us).  Measured at C2 / C3: the list
    // takes 0.3 ms off the k-truss fixed point at both sizes and adds 0.1-0.2 ms to k-core's (its units are long: few
    // groups have a single marked unit); the notification kernel pays where hubs are many (k-core at C3: 235 of them in
    // ~600 chunks, -0.13 ms; 3 x C3: -2.4 ms) and costs 0.2-0.4 ms where they are few (C2, k-truss).
    uint32_t defer_chunks = kLocDeferChunks;
    if (const char *e = ctx_opt(ctx, "LOCAL_DEFER_CHUNKS")) defer_chunks = (uint32_t)strtoul(e, nullptr, 10);     // (no result depends on it)
    g.flags = (use_list ? kLocUseList : 0u) | (g.nchunk >= defer_chunks && g.nchunk > 0 ? kLocDeferNotify : 0u);
#ifdef KOMB_DEBUG_SWITCHES
    if (const char *e = getenv("KOMB_LOCAL_MODE")) g.flags = (uint32_t)atoi(e);      // 1 = list, 2 = notification kernel
#endif
    if (g.ng) {
        KOMB_HIP(ctx, bufs.alloc(&g.gchunk, (size_t)g.nchunk));
        KOMB_HIP(ctx, bufs.alloc(&g.gacc, (size_t)g.ng));
        KOMB_HIP(ctx, bufs.alloc(&g.gwin, (size_t)g.ng * kGiantWin));
        KOMB_HIP(ctx, bufs.alloc(&g.gnote, (size_t)g.ng));
        k_local_chunks<<<(int)(g.ng > 1024u ? 1024u : g.ng), kBlock, 0, s>>>(g);
    }
    // K = h-index of the live keys: no level of the remainder is above it, so no value needs to start above it
    int32_t K = 0x7FFFFFFF;
    {
        uint64_t ge = 0;
        for (int32_t k = (int32_t)kKeyBins - 1; k >= 0; --k) {
            ge += kh[(size_t)k];
            if (ge >= (uint64_t)k) { if (k < (int32_t)kKeyBins - 1) K = k; break; }     // the last bin is open-ended: no bound from it
        }
    }
    stamp(1);
    after_number(g);
    KOMB_HIP(ctx, bufs.alloc((unsigned char **)&d_items, (size_t)total * item_bytes));

    // collect: SCAN (every live unit is a hit) + PROCESS of the peel engine on its own control block
    peel_collect_ctrl(s, d_cctrl, d_ctrl);
    PeelCtrl hcc{};
    int32_t cl = 1;                                        // launch indices of the collect pass (its state starts at seq 1)
    for (int guard = 0; guard < 16 && hcc.done == 0; ++guard) {
        for (int i = 0; i < 3; ++i) launch_collect(g, d_num, d_items, d_cctrl, cl++);
        KOMB_HIP(ctx, d2h(ctx, &hcc, d_cctrl, sizeof(PeelCtrl)));
        cl = hcc.seq;                                      // launches that found nothing to do do not move the state on
    }
    if (hcc.done != 1) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "local finish: collect pass did not complete (state %d)", hcc.done);
    int64_t gn = ((int64_t)n + kBlock - 1) / kBlock;
    const int grid_n = (int)(gn > 1024 ? 1024 : gn);
    k_local_check<<<grid_n, kBlock, 0, s>>>(g, d_lctrl);
    if (K != 0x7FFFFFFF) k_local_clamp<<<grid_n, kBlock, 0, s>>>(g, K);
    stamp(2);

    int launches = 0;
    KOMB_TRY(run_fix(g, d_items, (uint64_t)total, d_lctrl, d_cnt, &launches));
    stamp(3);
    k_local_finish<<<grid_n, kBlock, 0, s>>>(g, add, out, d_present, d_lctrl);
    k_local_levels<<<(int)(present_words / kBlock + 1 > 256 ? 256 : present_words / kBlock + 1), kBlock, 0, s>>>(d_present, (uint32_t)present_words, d_cnt, d_lctrl);
    stamp(4);
    KOMB_HIP(ctx, d2h(ctx, &hl, d_lctrl, sizeof(LocalCtrl)));
    if (dbg) {
        float t[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) (void)hipEventElapsedTime(&t[i], ev[i], ev[i + 1]);
        fprintf(stderr, "komb local finish: %u units (%u heavy, %u of them long; bound %d), %u items; number+scan %.1f us, collect %.1f us, %d sweeps (%d launches, %u evaluations) %.1f us, scatter %.1f us\n",
                n, g.nh, g.ng, K == 0x7FFFFFFF ? -1 : K, total, t[0] * 1e3f, t[1] * 1e3f, hl.iters, launches, hl.evals, t[2] * 1e3f, t[3] * 1e3f);
    }
#ifdef KOMB_LOCAL_TIMERS
    {
        unsigned long long tm[72];
        KOMB_HIP(ctx, d2h(ctx, tm, d_cnt + kCntWords, sizeof(tm)));
        for (int sw = 0; sw < 6; ++sw) {
            const unsigned long long *t = tm + sw * 12;
            const double nwv = t[10] ? (double)t[10] : 1.0;
            fprintf(stderr, "komb local timers, sweep %d (%llu waves), us per wave: hscan %.1f medium %.1f workgroup %.1f setup %.1f load+count %.1f search %.1f notify %.1f; "
                            "batches/wave %.2f groups/wave %.2f; slowest wave %.1f us\n",
                    sw == 0 ? 1 : sw * 4, t[10], t[0] / nwv / 100.0, t[1] / nwv / 100.0, t[2] / nwv / 100.0, t[3] / nwv / 100.0, t[4] / nwv / 100.0, t[5] / nwv / 100.0,
                    t[6] / nwv / 100.0, t[7] / nwv, t[8] / nwv, t[9] / 100.0);
        }
    }
#endif
    if (hl.bad) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "local finish: %u consistency failures in the compact index", hl.bad);
    if (!hl.done) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "local finish: launch budget exhausted before the fixed point");
    if (ls) {
        ls->units = n; ls->heavy = g.nh; ls->levels = hl.levels; ls->max_val = hl.max_val;
        ls->sweeps = hl.iters; ls->launches = launches; ls->items = total; ls->evals = hl.evals;
    }
    release_all();
    return KOMB_OK;
}

} // namespace komb
