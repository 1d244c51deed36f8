// peel_dev.h -- the level-synchronous peel engine shared by k-core and k-truss.
//
// A peel problem P has `units` (vertices / edges), each with a live integer key
// (degree / support) and a slice of `items` (CSR row / triangle incidence
// slice).  Peeling a unit walks its slice; each item may decrement other
// units' keys, and the decrement that lands a unit exactly on the current
// level "triggers" it into the next sub-round's frontier.
//
// One kernel, k_peel_step<P>, is launched over and over; what a launch does is
// decided on the device from the control block its predecessor finalised:
//   SCAN    : two passes over all units (later: over the compacted list of live
//             units, rewritten by every SCAN once <= 1/2 of the units is left).
//             Pass A classifies every unit (one byte each) and counts, per wave,
//             the live units with key <= level; one atomic per WORKGROUP
//             reserves queue space; pass B replays the bytes and writes the
//             queue entries at wave-private offsets (no append atomics at all).
//             Four units per thread per trip keep the sweep's loads overlapped.
//   PROCESS : the frontier lives in two queues.  "Light" units (<= kLight
//             items) are taken 64 per wavefront and their slices flattened
//             across the lanes (prefix sum in LDS + binary search), so lanes
//             stay busy on power-law slices; "heavy" units are pre-split into
//             kChunk-item chunks, each its own queue entry, so a hub row or a
//             hub edge's 10^4 triangles spread over hundreds of wavefronts
//             instead of serialising one.  Triggered units are staged in a
//             per-wave LDS buffer and flushed with one atomic per ~100 entries.
// The last workgroup to finish (two-level arrival ticket: 32 workgroups per
// group counter, then one top counter, so no word sees more than ~32 serial
// atomics; no cache-flushing fences, only atomics cross workgroups inside a
// launch) rewrites the control block for the next launch.  A step that fits
// one workgroup is finalised locally and the workgroup chains the next steps
// in-kernel while they stay that small.
#pragma once

#include "common.h"

namespace komb {

enum : int32_t { MODE_SCAN = 0, MODE_PROCESS = 1, MODE_RETIRE = 2 };   // RETIRE is a flag on top of the step that follows it (problems with byte states)

#ifndef KOMB_PEEL_BLOCK
#define KOMB_PEEL_BLOCK 1024
#endif
#ifndef KOMB_PEEL_PER_CU
#define KOMB_PEEL_PER_CU 1
#endif
constexpr int kPeelBlock = KOMB_PEEL_BLOCK;     // 16 wave64 per workgroup
constexpr int kPeelPerCu = KOMB_PEEL_PER_CU;   // workgroups per CU the grid is sized for
constexpr int kPeelWaves = kPeelBlock / kWave;
constexpr int kLight = 64;                     // units with <= kLight items are flattened 64 per wave
constexpr int kChunk = 128;                    // heavy units: one queue entry per kChunk items (= one trip of a wavefront)
constexpr int kScanU = 4;                      // units per thread per trip in SCAN
constexpr int kItemU = 2;                      // items per lane per trip in PROCESS
constexpr int kChainBudget = 16;               // batches a wave may peel from its own triggers (chainable problems)
constexpr int kStage = 192;                    // per-wave LDS staging of triggered light units
constexpr int kGroup = 32;                     // workgroups per first-level arrival counter
constexpr int kMaxGroups = 64;                 // grid <= kGroup * kMaxGroups
constexpr int kInitOff = 2 * (kMaxGroups + 1);  // the ticket buffer: kMaxGroups + 1 64-bit tickets, then the init kernel's two words

struct CtrlView {                              // launch-stable part of the control block
    int32_t mode, level, round, done, cur_sel;
    uint32_t cur_light, cur_heavy, remaining;
    uint32_t live_count;
    int32_t live_sel, live_mode;
    uint32_t tail_limit;
};

__device__ __forceinline__ int32_t alive_marker(uint32_t len)
{
    return kAlive - (int32_t)(len <= (uint32_t)kLight ? 0u : (len + kChunk - 1) / kChunk);
}
__device__ __forceinline__ bool marker_alive(int32_t m) { return m >= kAliveMin; }
__device__ __forceinline__ uint32_t marker_chunks(int32_t m) { return (uint32_t)(kAlive - m); }   // 0 = light

// One-byte shadow of a stamp, for the problems whose item visits gather other units' states at random (k-truss: two per
// triangle visit).  100 MB instead of 400 MB at C3: the gathers' lines come from the L2s / the 256 MB memory-side cache
// instead of HBM.  A byte says: alive (light / heavy slice), gone, or "entered the frontier of sub-round q" as q mod
// kStateWindow; state_rel() places such a q relative to the current sub-round r, which is exact while r - q < kStateWindow - 1:
// codes of sub-rounds that old are rewritten as gone by the engine's RETIRE step (every kRetireEvery sub-rounds).
enum : uint32_t { ST_ALIVE_LIGHT = 0, ST_ALIVE_HEAVY = 1, ST_GONE = 2, ST_ROUND0 = 3 };
enum : int { REL_GONE = 0, REL_NOW = 1, REL_LATER = 2 };
constexpr uint32_t kStateWindow = 253;          // 256 - ST_ROUND0
constexpr int32_t kRetireEvery = 120;           // sub-rounds between two RETIRE steps: codes in use span <= kRetireEvery + 1 sub-rounds (checked: PeelCtrl::max_retire_gap)
static_assert(2 * kRetireEvery + 2 < (int32_t)kStateWindow, "state codes must not wrap between two RETIRE steps, even if one were late by a whole period");
__device__ __forceinline__ uint8_t state_of_round(int32_t q) { return (uint8_t)(ST_ROUND0 + (uint32_t)q % kStateWindow); }
__device__ __forceinline__ uint8_t state_of_stamp(int32_t stamp)      // initial states: stamp = alive marker, 0 (triangle-free: gone) or 1 (first frontier)
{
    if (stamp >= kAliveMin) return (uint8_t)(stamp == kAlive ? ST_ALIVE_LIGHT : ST_ALIVE_HEAVY);
    return stamp == 0 ? (uint8_t)ST_GONE : state_of_round(stamp);
}
__device__ __forceinline__ int32_t marker_of_state(uint32_t c)         // what SCAN needs of a marker: alive or not, light or heavy
{
    return c == ST_ALIVE_LIGHT ? kAlive : (c == ST_ALIVE_HEAVY ? kAlive - 1 : 0);
}
__device__ __forceinline__ int state_rel(uint32_t c, int32_t r)
{
    if (c < ST_GONE) return REL_LATER;
    if (c == ST_GONE) return REL_GONE;
    uint32_t d = c - ST_ROUND0 + kStateWindow - (uint32_t)r % kStateWindow;   // (q - r) mod window
    d = d >= kStateWindow ? d - kStateWindow : d;
    return d == 0 ? REL_NOW : (d == 1 ? REL_LATER : REL_GONE);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Coherent read of a word other workgroups update with atomics in this launch.
__device__ __forceinline__ uint32_t coherent_load(uint32_t *p) { return atomicAdd(p, 0u); }
__device__ __forceinline__ int32_t coherent_load(int32_t *p) { return atomicAdd(p, 0); }

__device__ __forceinline__ int32_t wave_min(int32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = lane_id();
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// One atomic per WORKGROUP on a word every workgroup adds to (a word takes ~88 atomics per microsecond: a wavefront-level
// add from a 4096-workgroup grid queues 16 000 of them).  All threads of the workgroup must call these.
__device__ __forceinline__ void block_add_u64(unsigned long long v, unsigned long long *dst)
{
    __shared__ unsigned long long sh_blk64[16];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = (int)(threadIdx.x >> 6), nw = (int)((blockDim.x + 63) >> 6);
    __syncthreads();
    if (lane_id() == 0) sh_blk64[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int i = 0; i < nw; ++i) t += sh_blk64[i]; if (t) atomicAdd(dst, t); }
}
__device__ __forceinline__ void block_add_min(uint32_t add, int32_t mn, uint32_t *dst_add, int32_t *dst_min)
{
    __shared__ uint32_t sh_blk_a[16];
    __shared__ int32_t sh_blk_m[16];
    add = wave_sum(add);
    mn = wave_min(mn);
    const int w = (int)(threadIdx.x >> 6), nw = (int)((blockDim.x + 63) >> 6);
    __syncthreads();
    if (lane_id() == 0) { sh_blk_a[w] = add; sh_blk_m[w] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0; int32_t m = 0x7FFFFFFF;
        for (int i = 0; i < nw; ++i) { a += sh_blk_a[i]; m = min(m, sh_blk_m[i]); }
        if (a) atomicAdd(dst_add, a);
        if (m != 0x7FFFFFFF) atomicMin(dst_min, m);
    }
}

enum : uint8_t { SC_NONE = 0, SC_LIGHT = 1, SC_HEAVY = 2, SC_SURVIVOR = 4 };   // SCAN pass A -> pass B

struct PeelQueues {
    uint8_t *code;           // one classification byte per SCAN input position
    int32_t *light[2];       // unit ids
    int2 *heavy[2];          // (unit id, chunk index)
    int32_t *live[2];        // compacted ids of the units still live (SCAN's input once it pays off)
    int scalar_scan;         // debug: dense sweeps without the 16-byte loads
    int32_t *rlevel = nullptr;   // optional: rlevel[r] = the level sub-round r peeled at, written once per PROCESS step by its
                                 // finaliser -- a problem whose result is "the level a unit was peeled at" then needs no
                                 // result store per unit: the sub-round stamp it writes anyway says it (k-truss)
};

// ---------------------------------------------------------------- appenders
// All of these must be reached by all 64 lanes of the wave.

// `pred` lanes append `val` at wave-private position base + running offset (SCAN pass B).
__device__ __forceinline__ void wave_write_ordered(bool pred, int32_t val, int32_t *q, uint32_t base, uint32_t &run)
{
    const uint64_t m = __ballot(pred);
    if (pred) q[base + run + (uint32_t)__popcll(m & lanemask_lt())] = val;
    run += (uint32_t)__popcll(m);
}

// heavy units of `pred` lanes: every chunk becomes one (unit, chunk) entry, written by the whole wave
__device__ __forceinline__ void wave_write_chunks(bool pred, int32_t unit, uint32_t nchunks, int2 *q, uint32_t base, uint32_t &run)
{
    uint64_t m = __ballot(pred);
    const int lane = lane_id();
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int32_t u = __shfl(unit, src);
        const uint32_t n = (uint32_t)__shfl((int)nchunks, src);
        for (uint32_t c = (uint32_t)lane; c < n; c += kWave) q[base + run + c] = make_int2(u, (int)c);
        run += n;
    }
}

// PROCESS: staged append of triggered units.  Light units and heavy units (with their chunk counts) collect in
// the wave's LDS buffers; a flush reserves queue space for both kinds at once -- two atomics issued from two
// lanes of one instruction, ONE round trip -- and then writes the light ids and expands the heavy units into
// their (unit, chunk) entries.  (Reserving for the heavy units inside the item loop cost a returning atomic per
// trip, on the critical path of most k-core steps.)
constexpr int kStageH = 64;                     // staged heavy units per wave
struct WaveStage {
    int32_t *buf;            // [kStage] light unit ids, LDS, private to the wave
    int2 *hbuf;              // [kStageH] (heavy unit id, chunks)
    uint32_t n, hn;          // wave-uniform fills
};
__device__ __forceinline__ void stage_flush(WaveStage &st, int32_t *ql, uint32_t *tail_l, int2 *qh, uint32_t *tail_h)
{
    if (st.n == 0 && st.hn == 0) return;
    const int lane = lane_id();
    __builtin_amdgcn_wave_barrier();
    const int2 mine = (uint32_t)lane < st.hn ? st.hbuf[lane] : make_int2(0, 0);
    const uint32_t incl = wave_incl_scan((uint32_t)mine.y);
    const uint32_t total_h = (uint32_t)__shfl((int)incl, kWave - 1);
    uint32_t r = 0;
    if (lane == 0 && st.n) r = atomicAdd(tail_l, st.n);
    else if (lane == 1 && total_h) r = atomicAdd(tail_h, total_h);
    const uint32_t base_l = (uint32_t)__shfl((int)r, 0), base_h = (uint32_t)__shfl((int)r, 1);
    for (uint32_t i = (uint32_t)lane; i < st.n; i += kWave) ql[base_l + i] = st.buf[i];
    const uint32_t excl = incl - (uint32_t)mine.y;
    for (uint32_t k = 0; k < st.hn; ++k) {
        const int32_t u = __shfl(mine.x, (int)k);
        const uint32_t n = (uint32_t)__shfl(mine.y, (int)k);
        const uint32_t o = base_h + (uint32_t)__shfl((int)excl, (int)k);
        for (uint32_t c = (uint32_t)lane; c < n; c += kWave) qh[o + c] = make_int2(u, (int)c);
    }
    __builtin_amdgcn_wave_barrier();
    st.n = 0; st.hn = 0;
}
__device__ __forceinline__ void stage_push(WaveStage &st, bool is_light, bool is_heavy, int32_t unit, uint32_t nchunks,
                                           int32_t *ql, uint32_t *tail_l, int2 *qh, uint32_t *tail_h)
{
    const uint64_t ml = __ballot(is_light), mh = __ballot(is_heavy);
    if ((ml | mh) == 0) return;
    const uint32_t cl = (uint32_t)__popcll(ml), ch = (uint32_t)__popcll(mh);
    if (st.n + cl > (uint32_t)kStage || st.hn + ch > (uint32_t)kStageH) stage_flush(st, ql, tail_l, qh, tail_h);
    if (is_light) st.buf[st.n + (uint32_t)__popcll(ml & lanemask_lt())] = unit;
    if (is_heavy) st.hbuf[st.hn + (uint32_t)__popcll(mh & lanemask_lt())] = make_int2(unit, (int)nchunks);
    st.n += cl; st.hn += ch;
}

// ------------------------------------------------------- step planning / finalisation
constexpr uint32_t kSmallScan = 16384;          // a live list this short is scanned by one workgroup
#ifndef KOMB_SMALL_LIGHT
#define KOMB_SMALL_LIGHT 64
#endif
#ifndef KOMB_SMALL_HEAVY
#define KOMB_SMALL_HEAVY 4
#endif
constexpr uint32_t kSmallLight = KOMB_SMALL_LIGHT;   // a frontier this small (and <= kSmallHeavy chunks) is one workgroup's step
constexpr uint32_t kSmallHeavy = KOMB_SMALL_HEAVY;
constexpr uint32_t kMaxInKernelSteps = 8192;    // bound on steps one launch may chain

// How many workgroups take part in the step described by cv, and the light batch width.
__device__ __forceinline__ uint32_t plan_step(const CtrlView &cv, uint32_t grid, uint32_t &bsz)
{
    // light units per wave batch: 64 when the frontier is large, fewer (>= 2: two units of <= 64 items are
    // one trip of the item loop) when it would otherwise leave most of the grid's wavefronts without work
    bsz = kWave;
    const uint64_t grid_waves = (uint64_t)grid * kPeelWaves;
    while (bsz > 2 && (uint64_t)cv.cur_light < grid_waves * (bsz / 2)) bsz >>= 1;
    if (cv.mode & MODE_RETIRE) return grid;                 // a sweep over every unit's state byte
    if (cv.mode == MODE_SCAN) return (cv.live_mode != 0 && cv.live_count <= kSmallScan) ? 1u : grid;
    if (cv.cur_light <= kSmallLight && cv.cur_heavy <= kSmallHeavy) {
        // small frontier: one workgroup (its 16 wavefronts share the units) -- cheaper than a
        // grid-wide step, and it lets the workgroup chain the following steps in-kernel
        bsz = 4;
        while (bsz < (uint32_t)kWave && bsz * kPeelWaves < cv.cur_light) bsz <<= 1;
        return 1u;
    }
    const uint64_t wave_jobs = ((uint64_t)cv.cur_light + bsz - 1) / bsz + (uint64_t)cv.cur_heavy;
    const uint64_t need = (wave_jobs + kPeelWaves - 1) / kPeelWaves;
    return (uint32_t)(need < 1 ? 1 : (need > grid ? grid : need));
}

// Rewrite the control block after a step.  Called by all 64 lanes of ONE wavefront once every
// participating workgroup has arrived; the independent atomics are issued from different lanes
// so they cost one round trip, not seven.  The new state is also left in *out (LDS) so that a
// single workgroup can go on to the next step without re-reading global memory.
__device__ __forceinline__ void finalize_step(PeelCtrl *ctrl, const CtrlView &cv, uint32_t units, CtrlView *out, uint32_t acc, int32_t launch,
                                              int32_t *rlevel, int32_t retire_every)
{
    const int lane = lane_id();
    // from here on the state belongs to the NEXT launch: late workgroups of this one must not act on it
    if (lane == 0) { (void)atomicExch(&ctrl->seq, launch + 1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_wave_barrier();
    if (cv.mode & MODE_RETIRE) {                            // the state bytes are fresh: on to the step that was due
        if (lane == 0) { ctrl->mode = cv.mode & ~MODE_RETIRE; ctrl->last_retire = cv.round; *out = cv; out->mode = cv.mode & ~MODE_RETIRE; }
        return;
    }
    const bool scan = cv.mode == MODE_SCAN;
    const int sel = cv.cur_sel;
    const bool emitted = scan && (cv.live_mode != 0 || cv.remaining <= units / 2);
    const int next_q = scan ? sel : (sel ^ 1);
    uint32_t v = 0;
    if (lane == 1) v = coherent_load(&ctrl->tail_l[next_q]);
    else if (lane == 2) v = coherent_load(&ctrl->tail_h[next_q]);
    else if (lane == 3) v = (uint32_t)atomicExch(&ctrl->next_min, 0x7FFFFFFF);
    else if (lane == 4) v = emitted ? atomicExch(&ctrl->live_tail, 0u) : 0u;
    else if (lane == 5) { if (!scan) atomicExch(&ctrl->tail_l[sel], 0u); }
    else if (lane == 6) { if (!scan) atomicExch(&ctrl->tail_h[sel], 0u); }
    else if (lane == 7) atomicAdd(scan ? &ctrl->n_scans : &ctrl->n_rounds, 1);
    const uint32_t cur_l = (uint32_t)__shfl((int)v, 1), cur_h = (uint32_t)__shfl((int)v, 2);
    const int32_t nmin = __shfl((int)v, 3);
    const uint32_t live_n = (uint32_t)__shfl((int)v, 4);
    if (lane != 0) return;
    if (!scan && rlevel) rlevel[cv.round] = cv.level;       // the units stamped with this sub-round were peeled at this level
    int32_t mode = cv.mode, level = cv.level, round = cv.round, done = 0, nsel = sel;
    int32_t live_sel = cv.live_sel, live_mode = cv.live_mode;
    uint32_t live_count = cv.live_count;
    const uint32_t remaining = cv.remaining - acc;
    if (scan) {
        if (emitted) { live_count = live_n; live_sel ^= 1; live_mode = 1; }
        if (acc > 0) { atomicAdd(&ctrl->n_levels, 1); ctrl->max_level = level; }
        if (cur_l + cur_h > 0) mode = MODE_PROCESS;
        else if (remaining == 0) done = 1;
        else if (acc > 0) level += 1;                       // only item-less units at this level
        else if (nmin == 0x7FFFFFFF) done = 2;              // live units but no live key: inconsistent
        else level = nmin;                                  // jump to the first populated level
    } else {
        nsel = sel ^ 1; round += 1;
        const bool retire = retire_every > 0 && round % retire_every == 0;
        if (cur_l + cur_h == 0) {
            if (remaining == 0) done = 1;
            else {
                level += 1; mode = MODE_SCAN;
                if (remaining <= cv.tail_limit) done = 3;    // the rest goes to the problem's tail kernel
            }
        }
        // state codes of sub-rounds long gone must not wrap around (state_rel).  Also when the remainder is being offered to a
        // finish (done = 3): should the finish refuse it, the engine goes on from this state and must not have skipped a RETIRE
#ifdef KOMB_TEST_OLD_RETIRE_RULE                            // (the rule before commit 9b51a4a, for the regression test's negative control)
        if (retire && !done) mode |= MODE_RETIRE;
#else
        if (retire && done != 1 && done != 2) mode |= MODE_RETIRE;
#endif
        // the invariant the byte states rest on, made explicit: every increment of `round` passes through here, and a RETIRE
        // step runs every retire_every sub-rounds -- the codes in use never span more than retire_every + 1 sub-rounds
        // (< kStateWindow - 1).  The host fails the run if the gap ever exceeded the period (drive checks in ktruss.hip).
        if (retire_every > 0) {
            const int32_t gap = round - ctrl->last_retire;
            if (gap > ctrl->max_retire_gap) ctrl->max_retire_gap = gap;
        }
    }
    ctrl->mode = mode; ctrl->level = level; ctrl->round = round; ctrl->done = done;
    ctrl->cur_sel = nsel; ctrl->cur_light = cur_l; ctrl->cur_heavy = cur_h; ctrl->remaining = remaining;
    ctrl->live_count = live_count; ctrl->live_sel = live_sel; ctrl->live_mode = live_mode;
    out->mode = mode; out->level = level; out->round = round; out->done = done;
    out->cur_sel = nsel; out->cur_light = cur_l; out->cur_heavy = cur_h; out->remaining = remaining;
    out->live_count = live_count; out->live_sel = live_sel; out->live_mode = live_mode;
    out->tail_limit = cv.tail_limit;
}

// ------------------------------------------------------------ the step kernel
// Problem concept (all __device__):
//   static constexpr bool kChain;        triggered light units may be peeled in the same launch
//   static constexpr bool kSingleStep;   a launch never chains a second step (the sharded peel of shard_dev.h)
//   uint32_t units;
//   const int32_t *scan_marker();        [units] alive marker (common.h: it carries the unit's class) or round / level number
//   const int32_t *scan_key();           [units] live keys, or null when every live unit is a hit (collect passes)
//   void mark_scanned(u, cv)             unit enters the frontier through SCAN
//   void slice(u, uint32_t &begin, uint32_t &len)
//   Loaded item_load(unit, pos, cv)      the item's loads (no side effects)
//   void item_apply(ld, cv, int32_t &t0, int32_t &t1, uint32_t &c0, uint32_t &c1)
//                                        decrements; ids of triggered units or -1, and their classes
// (load and apply are split so that the loads of several items are in flight together)
//   static constexpr int32_t kRetireEvery (optional) + void retire(cv, block, nblocks)
//                                        problems with one-byte states (state_of_round): the sweep that rewrites stale codes
template <class P, class = void> struct PeelRetire { static constexpr int32_t every = 0; };
template <class P> struct PeelRetire<P, decltype((void)P::kRetireEvery)> { static constexpr int32_t every = P::kRetireEvery; };
template <class P> __host__ __device__ constexpr int32_t peel_retire_every() { return PeelRetire<P>::every; }

template <class P>
__global__ __launch_bounds__(kPeelBlock, (kPeelBlock / 64) * kPeelPerCu / 4) void k_peel_step(PeelCtrl *ctrl, uint32_t *grp_done, PeelQueues Q, P p, int32_t launch)
{
    int32_t retire_every = 0;                              // sub-rounds between two RETIRE steps (problems with one-byte states only)
    if constexpr (peel_retire_every<P>() > 0) retire_every = p.retire_every;
    __shared__ CtrlView sh_cv;
    __shared__ uint32_t sh_w[kPeelWaves][4];           // per-wave counts / bases
    __shared__ uint32_t sh_base[3];
    __shared__ uint32_t sh_acc;                        // units this workgroup moved into a frontier in this step
    __shared__ int32_t sh_min[kPeelWaves];
    __shared__ uint32_t sh_end[kPeelWaves][kWave];
    __shared__ uint32_t sh_beg[kPeelWaves][kWave];
    __shared__ int32_t sh_unit[kPeelWaves][kWave];
    __shared__ int32_t sh_stage[kPeelWaves][kStage];
    __shared__ int2 sh_stage_h[kPeelWaves][kStageH];

#ifdef KOMB_STEP_TIMERS
    unsigned long long tk[8] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0, 0, 0, 0};
#define KOMB_TK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) tk[i] = wall_clock64(); } while (0)
#else
#define KOMB_TK(i) do { } while (0)
#endif
    // Entry: the state and its sequence word are one 64-byte line, loaded by 16 lanes of ONE instruction -- one snapshot.
    // A workgroup acts only on the state written for THIS launch (PeelCtrl::seq, common.h): one that is dispatched after a
    // finaliser has moved on is late and leaves.
    __shared__ int32_t sh_raw[16];
    if (threadIdx.x < 16) sh_raw[threadIdx.x] = reinterpret_cast<const int32_t *>(ctrl)[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        const PeelCtrl *c = reinterpret_cast<const PeelCtrl *>(sh_raw);        // the first line only
        sh_cv.mode = c->mode; sh_cv.level = c->level; sh_cv.round = c->round; sh_cv.done = c->done;
        sh_cv.cur_sel = c->cur_sel; sh_cv.cur_light = c->cur_light; sh_cv.cur_heavy = c->cur_heavy;
        sh_cv.remaining = c->remaining; sh_cv.live_count = c->live_count;
        sh_cv.live_sel = c->live_sel; sh_cv.live_mode = c->live_mode;
        sh_cv.tail_limit = ctrl->tail_limit;
        if (c->seq != launch) sh_cv.done = 1;
    }
    __syncthreads();
    CtrlView cv = sh_cv;
    if (cv.done) return;
    KOMB_TK(1);
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);

    // A launch normally performs ONE step.  When the step fits a single workgroup, that
    // workgroup finalises it locally and goes straight on to the next step for as long as the
    // steps stay that small: the long tail of tiny sub-rounds costs no launches at all.
    // (Inside one workgroup, __syncthreads() makes the plain stores of one step visible to the
    // loads of the next: same CU, same L1.  State is carried in LDS, never re-read from global.)
    for (uint32_t chained = 0;; ++chained) {
    uint32_t bsz;
    const uint32_t nblk = plan_step(cv, gridDim.x, bsz);   // workgroups that take part; the rest leave
    if (blockIdx.x >= nblk) return;
    const int L = cv.level;
    const int sel = cv.cur_sel;

    if (cv.mode & MODE_RETIRE) {
        if constexpr (peel_retire_every<P>() > 0) p.retire(cv, blockIdx.x, nblk);
        if (threadIdx.x == 0) sh_acc = 0;
    } else if (cv.mode == MODE_SCAN) {
        // ---- input: every unit, or the compacted live list; survivors are compacted into the
        // other live buffer once at most half of the units is left
        const bool from_list = cv.live_mode != 0;
        const bool emit = from_list || cv.remaining <= p.units / 2;
        const uint32_t n_in = from_list ? cv.live_count : p.units;
        const int32_t *live_in = Q.live[cv.live_sel];
        int32_t *live_out = Q.live[cv.live_sel ^ 1];
        // ---- pass A: count this wave's light hits, its heavy units' chunks, all hits, survivors.
        // kScanU units per thread per trip: the trips are independent, so their loads overlap
        // (one unit per trip left the sweep bound by 2 x 380 sequential memory latencies).
        // Dense sweep (no live list yet): a lane takes kScanU CONSECUTIVE units and loads their markers and keys as
        // one 16-byte vector each, and stores the four classification bytes as one word.
        const bool vec = !from_list && !Q.scalar_scan;
        const int32_t *s_marker = p.scan_marker(), *s_key = p.scan_key();
        constexpr bool byte_marker = peel_retire_every<P>() > 0;   // problems with one-byte states are swept through those (1 byte per unit instead of 4)
        const uint8_t *s_state = nullptr;
        if constexpr (byte_marker) s_state = p.scan_state();
        uint32_t n_light = 0, n_chunks = 0, n_hits = 0, n_surv = 0;   // per lane until the sums after the sweep
        int32_t lmin = 0x7FFFFFFF;
        const uint64_t tile = (uint64_t)kPeelBlock * kScanU;
        auto index_of = [&](uint64_t tb, int k) -> uint64_t {
            return vec ? tb + ((uint64_t)w * kWave + (uint64_t)lane) * kScanU + (uint64_t)k
                       : tb + (uint64_t)k * kPeelBlock + (uint64_t)w * kWave + (uint64_t)lane;
        };
        // dense sweep: the vectors of the NEXT trip are requested before this trip is classified, so a workgroup's ~100 trips
        // are not a chain of ~100 exposed memory latencies
        int4 nm4 = make_int4(0, 0, 0, 0), nk4 = make_int4(0, 0, 0, 0);
        auto fetch = [&](uint64_t tb) {
            const uint64_t i0 = index_of(tb, 0);
            if (vec && tb < n_in && i0 + kScanU <= n_in) {
                if constexpr (byte_marker) nm4.x = (int32_t)*reinterpret_cast<const uint32_t *>(s_state + i0);
                else nm4 = *reinterpret_cast<const int4 *>(s_marker + i0);
                if (s_key) nk4 = *reinterpret_cast<const int4 *>(s_key + i0);
            }
        };
        fetch((uint64_t)blockIdx.x * tile);
        for (uint64_t tb = (uint64_t)blockIdx.x * tile; tb < n_in; tb += (uint64_t)nblk * tile) {
            uint8_t code[kScanU];
            int32_t mk[kScanU], ky[kScanU];
            const uint64_t i0 = index_of(tb, 0);
            const bool whole = vec && i0 + kScanU <= n_in;
            if (whole) {
                static_assert(kScanU == 4, "the dense sweep loads int4");
                if constexpr (byte_marker) {
#pragma unroll
                    for (int k = 0; k < kScanU; ++k) mk[k] = marker_of_state(((uint32_t)nm4.x >> (8 * k)) & 0xFFu);
                } else { mk[0] = nm4.x; mk[1] = nm4.y; mk[2] = nm4.z; mk[3] = nm4.w; }
                ky[0] = nk4.x; ky[1] = nk4.y; ky[2] = nk4.z; ky[3] = nk4.w;
            }
            fetch(tb + (uint64_t)nblk * tile);
#pragma unroll
            for (int k = 0; k < kScanU; ++k) {
                const uint64_t idx = index_of(tb, k);
                code[k] = SC_NONE;
                if (idx < n_in) {
                    // liveness, key and slice length are loaded unconditionally: independent loads (and, in
                    // the dense sweep, sequential ones) instead of a chain of three dependent round trips
                    if (!whole) {
                        const uint32_t u = from_list ? (uint32_t)live_in[idx] : (uint32_t)idx;
                        if constexpr (byte_marker) mk[k] = marker_of_state(s_state[u]);
                        else mk[k] = s_marker[u];
                        ky[k] = s_key ? s_key[u] : 0;
                    }
                    const bool live = marker_alive(mk[k]);
                    uint32_t nch = marker_chunks(mk[k]);
                    if (live && ky[k] <= L) {
                        if constexpr (byte_marker) if (nch) {           // (a state byte only says "heavy": the chunk count is the slice's)
                            const uint32_t u = from_list ? (uint32_t)live_in[idx] : (uint32_t)idx;
                            uint32_t b, len; p.slice(u, b, len); nch = (len + kChunk - 1) / kChunk;
                        }
                        if (nch == 0) code[k] = SC_LIGHT;
                        else { n_chunks += nch; code[k] = SC_HEAVY; }
                    } else if (live) { lmin = min(lmin, ky[k]); code[k] = SC_SURVIVOR; }
                }
            }
            if (whole) *reinterpret_cast<uchar4 *>(Q.code + i0) = make_uchar4(code[0], code[1], code[2], code[3]);
#pragma unroll
            for (int k = 0; k < kScanU; ++k) {
                const uint64_t idx = index_of(tb, k);
                if (!whole && idx < n_in) Q.code[idx] = code[k];      // pass B reads this byte instead of the unit's state
                // per-lane tallies, summed over the wavefront once at the end (three ballots + popcounts per unit made the
                // sweep issue-bound: ~190 instructions per 64 units, 1.6 TB/s)
                n_light += code[k] == SC_LIGHT ? 1u : 0u;
                n_hits += (code[k] == SC_LIGHT || code[k] == SC_HEAVY) ? 1u : 0u;
                n_surv += (emit && code[k] == SC_SURVIVOR) ? 1u : 0u;
            }
        }
        n_light = wave_sum(n_light); n_hits = wave_sum(n_hits); n_surv = wave_sum(n_surv);
        n_chunks = wave_sum(n_chunks);
        lmin = wave_min(lmin);
        if (lane == 0) { sh_w[w][0] = n_light; sh_w[w][1] = n_chunks; sh_w[w][2] = n_hits; sh_w[w][3] = n_surv; sh_min[w] = lmin; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tl = 0, th = 0, ts = 0, hits = 0;
            int32_t bmin = 0x7FFFFFFF;
            for (int i = 0; i < kPeelWaves; ++i) {
                const uint32_t a = sh_w[i][0], b = sh_w[i][1], c = sh_w[i][3];
                sh_w[i][0] = tl; sh_w[i][1] = th; sh_w[i][3] = ts;   // exclusive offsets inside the workgroup's reservations
                tl += a; th += b; ts += c; hits += sh_w[i][2];
                bmin = min(bmin, sh_min[i]);
            }
            sh_base[0] = tl ? atomicAdd(&ctrl->tail_l[sel], tl) : 0u;
            sh_base[1] = th ? atomicAdd(&ctrl->tail_h[sel], th) : 0u;
            sh_base[2] = ts ? atomicAdd(&ctrl->live_tail, ts) : 0u;
            sh_acc = hits;
            if (bmin != 0x7FFFFFFF) atomicMin(&ctrl->next_min, bmin);
        }
        __syncthreads();
        // ---- pass B: replay the classification bytes, write the entries
        const uint32_t base_l = sh_base[0] + sh_w[w][0], base_h = sh_base[1] + sh_w[w][1], base_s = sh_base[2] + sh_w[w][3];
        uint32_t run_l = 0, run_h = 0, run_s = 0;
        // (dense sweep: the wavefront replays the SAME 256 units it classified, now 64 consecutive ones per instruction, so
        // that its stores -- queue entries, markers, results -- are whole lines and the live list stays in ascending order)
        auto index_b = [&](uint64_t tb, int k) -> uint64_t {
            return vec ? tb + (uint64_t)w * (kWave * kScanU) + (uint64_t)k * kWave + (uint64_t)lane : index_of(tb, k);
        };
        for (uint64_t tb = (uint64_t)blockIdx.x * tile; tb < n_in; tb += (uint64_t)nblk * tile) {
            uint8_t code[kScanU];
            uint32_t u[kScanU];
#pragma unroll
            for (int k = 0; k < kScanU; ++k) {
                const uint64_t idx = index_b(tb, k);
                code[k] = idx < n_in ? Q.code[idx] : (uint8_t)SC_NONE;
            }
#pragma unroll
            for (int k = 0; k < kScanU; ++k) {
                const uint64_t idx = index_b(tb, k);
                u[k] = 0;
                if (code[k] != SC_NONE) u[k] = from_list ? (uint32_t)live_in[idx] : (uint32_t)idx;
            }
#pragma unroll
            for (int k = 0; k < kScanU; ++k) {
                const bool light = code[k] == SC_LIGHT, heavy = code[k] == SC_HEAVY, surv = code[k] == SC_SURVIVOR;
                uint32_t nch = 0;
                if (heavy) { uint32_t b, len; p.slice(u[k], b, len); nch = (len + kChunk - 1) / kChunk; }
                if (code[k] != SC_NONE && !surv) p.mark_scanned(u[k], cv);
                wave_write_ordered(light, (int32_t)u[k], Q.light[sel], base_l, run_l);
                if (__ballot(heavy)) wave_write_chunks(heavy, (int32_t)u[k], nch, Q.heavy[sel], base_h, run_h);
                if (emit) wave_write_ordered(surv, (int32_t)u[k], live_out, base_s, run_s);
            }
        }
    } else {
        // ---- PROCESS
        int32_t *qn_l = Q.light[sel ^ 1];
        int2 *qn_h = Q.heavy[sel ^ 1];
        uint32_t *tail_nl = &ctrl->tail_l[sel ^ 1], *tail_nh = &ctrl->tail_h[sel ^ 1];
        WaveStage st{sh_stage[w], sh_stage_h[w], 0u, 0u};
        uint32_t n_trig = 0;                                // wave-uniform
        const uint64_t gw = (uint64_t)blockIdx.x * kPeelWaves + (uint64_t)w;
        const uint64_t nw = (uint64_t)nblk * kPeelWaves;

        auto push_trigger = [&](int32_t t, uint32_t nch) {
            const bool trig = t >= 0;
            const uint64_t m = __ballot(trig);
            if (m == 0) return;
            n_trig += (uint32_t)__popcll(m);
            const bool is_light = trig && nch == 0;
            const bool is_heavy = trig && nch != 0;
            stage_push(st, is_light, is_heavy, t, nch, qn_l, tail_nl, qn_h, tail_nh);
        };
        // kItemU items per lane per trip: all loads first, then the decrements
        auto run_items = [&](const bool (&active)[kItemU], const int32_t (&unit)[kItemU], const uint32_t (&pos)[kItemU]) {
            typename P::Loaded ld[kItemU];
#pragma unroll
            for (int k = 0; k < kItemU; ++k)
                if (active[k]) ld[k] = p.item_load(unit[k], pos[k], cv);
            int32_t trg[2 * kItemU];
            uint32_t tch[2 * kItemU];
#pragma unroll
            for (int k = 0; k < kItemU; ++k) {
                trg[2 * k] = -1; trg[2 * k + 1] = -1;
                tch[2 * k] = 0; tch[2 * k + 1] = 0;
                if (active[k]) p.item_apply(ld[k], cv, trg[2 * k], trg[2 * k + 1], tch[2 * k], tch[2 * k + 1]);
            }
#pragma unroll
            for (int k = 0; k < 2 * kItemU; ++k) push_trigger(trg[k], tch[k]);
        };

        // light units: bsz per wave, slices flattened over the lanes
        const uint64_t n_batches = ((uint64_t)cv.cur_light + bsz - 1) / bsz;
        uint32_t *s_end = sh_end[w], *s_beg = sh_beg[w];
        int32_t *s_unit = sh_unit[w];
        auto run_batch = [&](int32_t unit, uint32_t beg, uint32_t len) {
            const uint32_t incl = wave_incl_scan(len);
            const uint32_t total = (uint32_t)__shfl((int)incl, kWave - 1);
            __builtin_amdgcn_wave_barrier();
            s_end[lane] = incl; s_beg[lane] = beg; s_unit[lane] = unit;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t it0 = 0; it0 < total; it0 += kWave * kItemU) {
                bool active[kItemU];
                int32_t me[kItemU];
                uint32_t pos[kItemU];
#pragma unroll
                for (int k = 0; k < kItemU; ++k) {
                    const uint32_t it = it0 + (uint32_t)(k * kWave + lane);
                    active[k] = it < total;
                    me[k] = -1; pos[k] = 0;
                    if (active[k]) {
                        int lo = 0;                         // smallest t with s_end[t] > it (branchless, 6 fixed steps)
#pragma unroll
                        for (int st = kWave / 2; st > 0; st >>= 1) lo += (s_end[lo + st - 1] <= it) ? st : 0;
                        const uint32_t first = lo ? s_end[lo - 1] : 0u;
                        me[k] = s_unit[lo];
                        pos[k] = s_beg[lo] + (it - first);
                    }
                }
                run_items(active, me, pos);
            }
            __builtin_amdgcn_wave_barrier();
        };
        // One job list: the light batches, then the heavy chunks; wave g takes jobs g, g + nw, ...  (with
        // separate strides for the two kinds the first waves got a batch AND a chunk -- three dependent
        // trips more on the step's critical path -- while the last ones had nothing to do)
        const uint64_t n_jobs = n_batches + (uint64_t)cv.cur_heavy;
        for (uint64_t job = gw; job < n_jobs; job += nw) {
            if (job < n_batches) {
                const uint64_t idx = job * bsz + (uint64_t)lane;
                int32_t unit = -1;
                uint32_t beg = 0, len = 0;
                if ((uint32_t)lane < bsz && idx < cv.cur_light) { unit = Q.light[sel][idx]; p.slice((uint32_t)unit, beg, len); }
#ifdef KOMB_STEP_TIMERS
                if (blockIdx.x == 0 && threadIdx.x == 0 && tk[2] == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tk[2] = wall_clock64(); }
#endif
                run_batch(unit, beg, len);
                continue;
            }
            // heavy chunk: kChunk items of one unit
            const int2 ent = Q.heavy[sel][job - n_batches];
            uint32_t beg, len;
            p.slice((uint32_t)ent.x, beg, len);
#ifdef KOMB_STEP_TIMERS
            if (blockIdx.x == 0 && threadIdx.x == 0 && tk[2] == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tk[2] = wall_clock64(); }
#endif
            const uint32_t c0 = (uint32_t)ent.y * kChunk;
            const uint32_t c1 = min(len, c0 + (uint32_t)kChunk);
            for (uint32_t it0 = c0; it0 < c1; it0 += kWave * kItemU) {
                bool active[kItemU];
                int32_t me[kItemU];
                uint32_t pos[kItemU];
#pragma unroll
                for (int k = 0; k < kItemU; ++k) {
                    const uint32_t it = it0 + (uint32_t)(k * kWave + lane);
                    active[k] = it < c1; me[k] = ent.x; pos[k] = beg + it;
                }
                run_items(active, me, pos);
            }
        }
        // Problems whose peel order is free (k-core: a decrement is owed per (peeled vertex, live
        // neighbour) pair, whenever it happens) let the wave peel the light units it has just
        // triggered itself, straight from its staging buffer, for a bounded number of batches:
        // cascades advance several hops per launch.  (The truss peel needs the sub-round barrier
        // for its tie-break, so it always defers to the next launch.)
        if (P::kChain) {
            for (int budget = kChainBudget; budget > 0 && st.n > 0; --budget) {
                const uint32_t take = st.n < (uint32_t)kWave ? st.n : (uint32_t)kWave;
                int32_t unit = -1;
                uint32_t beg = 0, len = 0;
                __builtin_amdgcn_wave_barrier();
                if ((uint32_t)lane < take) { unit = st.buf[st.n - take + (uint32_t)lane]; p.slice((uint32_t)unit, beg, len); }
                __builtin_amdgcn_wave_barrier();
                st.n -= take;
                run_batch(unit, beg, len);
            }
        }
        KOMB_TK(3);
        stage_flush(st, qn_l, tail_nl, qn_h, tail_nh);
#ifdef KOMB_STEP_TIMERS
        if (blockIdx.x == 0 && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tk[4] = wall_clock64(); }
#endif
        if (lane == 0) sh_w[w][3] = n_trig;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int i = 0; i < kPeelWaves; ++i) t += sh_w[i][3];
            sh_acc = t;
        }
    }

    // ---- end of step
    __syncthreads();
    KOMB_TK(5);
    if (nblk > 1) {
        // several workgroups: two-level arrival ticket; the last one rewrites the control block.
        // Only ATOMICS cross workgroups inside a launch (queue cursors, acc, next_min); every plain store
        // (queue entries, stamps, results) is consumed by the NEXT launch, behind the kernel boundary.  So
        // arrival needs no cache write-back / invalidate (a __threadfence() costs ~3.5 us here, and the
        // path used to have four): the __syncthreads() above has drained every wave's memory operations
        // (atomics are acknowledged after they execute at the memory side), and the arrival atomics
        // themselves are ordered by their data dependence.
        if (threadIdx.x >= kWave) return;
        int last = 0;
        uint32_t total = 0;
        if (lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // a ticket is 64 bits: arrivals in the high word, the arrivals' frontier counts summed in the low
            // word -- the step's total reaches the finaliser with the arrival itself, not through one more
            // atomic that every workgroup would have to wait out first
            unsigned long long *tick = reinterpret_cast<unsigned long long *>(grp_done);
            const uint32_t grp = blockIdx.x / kGroup;
            const uint32_t ngrp = (nblk + kGroup - 1) / kGroup;
            const uint32_t grp_size = (grp == ngrp - 1) ? (nblk - grp * kGroup) : (uint32_t)kGroup;
            const uint32_t mine = sh_acc;
            const unsigned long long old = atomicAdd(&tick[grp], (1ull << 32) | mine);
            if ((uint32_t)(old >> 32) == grp_size - 1) {
                atomicExch(&tick[grp], 0ull);
                const uint32_t grp_acc = (uint32_t)old + mine;
                if (ngrp == 1) { last = 1; total = grp_acc; }      // a single group: its last arrival is the last arrival
                else {
                    const unsigned long long old2 = atomicAdd(&tick[kMaxGroups], (1ull << 32) | grp_acc);
                    if ((uint32_t)(old2 >> 32) == ngrp - 1) {
                        atomicExch(&tick[kMaxGroups], 0ull);
                        last = 1; total = (uint32_t)old2 + grp_acc;
                    }
                }
            }
        }
#ifdef KOMB_STEP_TIMERS
        if (blockIdx.x == 0 && threadIdx.x == 0 && cv.mode == MODE_PROCESS && cv.cur_light + cv.cur_heavy < 16384 && tk[2] != 0) {
            tk[6] = wall_clock64();
            // sums of: ctrl read, queue + slice, items, flush, barrier, ticket; and the number of steps sampled
            for (int i = 0; i < 6; ++i) atomicAdd(&ctrl->pad1[i], (uint32_t)(tk[i + 1] - tk[i]));
            atomicAdd(&ctrl->pad1[6], 1u);
        }
#endif
        last = __shfl(last, 0);
        total = (uint32_t)__shfl((int)total, 0);
        if (last) finalize_step(ctrl, cv, p.units, &sh_cv, total, launch, Q.rlevel, retire_every);
        return;
    }
    // one workgroup did the whole step: finalise locally, chain the next step if it is small too
    if (threadIdx.x < kWave) finalize_step(ctrl, cv, p.units, &sh_cv, sh_acc, launch, Q.rlevel, retire_every);
    if (P::kSingleStep) return;                        // (shard_dev.h: the next frontier has to be exchanged first)
    __syncthreads();
    cv = sh_cv;
    if (cv.done || chained >= kMaxInKernelSteps) return;
    uint32_t next_bsz;
    if (plan_step(cv, gridDim.x, next_bsz) != 1u) return;
    }   // for (chained)
}

} // namespace komb
