// shard_dev.h -- the peel of peel_dev.h split over the ranks of one node ("sparse path"):
// every rank holds the whole read-only structure (CSR / incidence index) and the whole liveness state, and OWNS a
// contiguous range of the units (vertices for k-core, internal edge ids for k-truss): only the owner keeps a unit's
// live key (degree / support) and only the owner decrements it.
//
// One sub-round, on every rank:
//   1. the rank's own part of the frontier is a plain id list (the SCAN of its own range at a level's start, afterwards
//      the units its own decrements triggered);
//   2. EXCHANGE: the lists of all ranks are concatenated on every rank.  The C ABI gives the library one collective, an
//      in-place SUM all-reduce of 32-bit words (komb_allreduce_fn); a concatenation is that all-reduce over a buffer in
//      which every rank has filled its own segment and zeroed the others.  Two calls per sub-round: the ranks' counts
//      (and, after a SCAN, each rank's smallest live key above the level, for level skipping), then the ids;
//   3. every rank stamps ALL frontier units (so liveness -- and, for the truss peel, the sub-round stamps its tie-break
//      reads -- stay replicated), sorts them into the engine's light / heavy queues, and runs ONE PROCESS step of
//      k_peel_step over the WHOLE frontier with a problem type that applies a decrement only when the target is in the
//      rank's own range: every rank recomputes the same decisions and keeps the share it owns.  No atomics cross ranks.
// The control flow on every rank follows the exchanged totals only, so the ranks stay in lock step without any other
// message, and integer decrements are order independent: results are bit-identical to the single-GPU peel.
//
// What this costs is in DESIGN.md section 6: every rank still reads every frontier unit's items (only the atomics are
// divided), and a sub-round pays two collectives and two host round trips.  It is the partition the survey prescribes,
// built and tested; it is not the faster way to use N GPUs for one graph, and it is opt-in (komb_set_shard_peel).
#pragma once

#include "peel_dev.h"

#include <chrono>

namespace komb {

// Ordered-by-nobody append of a workgroup's hits to a global list: ONE atomic on the list's fill per workgroup and trip
// (U units per thread) -- a word takes ~88 atomics per microsecond, so a wavefront-level add per 64 units would spend
// 17 ms on the fill alone when 100 M units are swept.  All threads of the workgroup must call it.
template <int U, class T>
__device__ __forceinline__ void block_append(const bool (&hit)[U], const T (&val)[U], uint32_t *fill, T *__restrict__ out)
{
    __shared__ uint32_t sh_cnt[kBlock / kWave];
    __shared__ uint32_t sh_at;
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    uint64_t m[U];
    uint32_t wave_total = 0;
#pragma unroll
    for (int k = 0; k < U; ++k) { m[k] = __ballot(hit[k]); wave_total += (uint32_t)__popcll(m[k]); }
    __syncthreads();                                         // (the previous trip's readers of sh_cnt / sh_at are done)
    if (lane == 0) sh_cnt[w] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int i = 0; i < kBlock / kWave; ++i) t += sh_cnt[i];
        sh_at = t ? atomicAdd(fill, t) : 0u;
    }
    __syncthreads();
    uint32_t o = sh_at;
    for (int i = 0; i < w; ++i) o += sh_cnt[i];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        if (hit[k]) out[o + (uint32_t)__popcll(m[k] & lanemask_lt())] = val[k];
        o += (uint32_t)__popcll(m[k]);
    }
}

constexpr int kShardU = 8;                                   // units per thread and trip of the sweeps below

// SCAN of the rank's own range [lo, hi) at level L: live units with key <= L -> out[], words[0] = how many;
// words[1] = the smallest live key above L in the range (0x7FFFFFFF: none)
template <class P>
__global__ __launch_bounds__(kBlock) void k_shard_scan(P p, uint32_t lo, uint32_t hi, int32_t L, int32_t *__restrict__ out, uint32_t *words)
{
    const int32_t *mk = p.scan_marker(), *ky = p.scan_key();
    int32_t lmin = 0x7FFFFFFF;
    const uint64_t tile = (uint64_t)kBlock * kShardU;
    for (uint64_t base = (uint64_t)lo + (uint64_t)blockIdx.x * tile; base < hi; base += (uint64_t)gridDim.x * tile) {
        bool hit[kShardU];
        int32_t id[kShardU];
#pragma unroll
        for (int k = 0; k < kShardU; ++k) {
            const uint64_t u = base + (uint64_t)k * kBlock + threadIdx.x;
            hit[k] = false; id[k] = (int32_t)u;
            if (u < hi && marker_alive(mk[u])) {
                const int32_t key = ky[u];
                if (key <= L) hit[k] = true; else lmin = min(lmin, key);
            }
        }
        block_append<kShardU>(hit, id, &words[0], out);
    }
    lmin = wave_min(lmin);
    if (lane_id() == 0 && lmin != 0x7FFFFFFF) atomicMin(reinterpret_cast<int32_t *>(&words[1]), lmin);
}

static __global__ void k_shard_scan_reset(uint32_t *words)
{
    if (threadIdx.x == 0) { words[0] = 0u; words[1] = 0x7FFFFFFFu; }
}

// the rank's line of the header exchange: hdr[rank] = its count, hdr[world + rank] = its smallest live key above the level,
// hdr[2 world + rank] = its STATUS (0, or 1 once something failed on this rank: every rank learns of it with the next header and
// all of them leave together, instead of one returning and the others waiting in a collective that will never complete)
static __global__ void k_shard_header(uint32_t *hdr, int world, int rank, const uint32_t *cnt, const uint32_t *mn, uint32_t *words, uint32_t status)
{
    for (int i = (int)threadIdx.x; i < 3 * world; i += (int)blockDim.x) hdr[i] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) { hdr[rank] = cnt ? *cnt : 0u; hdr[world + rank] = mn ? *mn : 0x7FFFFFFFu; hdr[2 * world + rank] = status; }
    if (threadIdx.x == 1 && words) { words[2] = 0u; words[3] = 0u; }     // the fills of the queues the next k_shard_mark writes
}

// every rank, over the whole exchanged frontier: stamp the units (mark_scanned: liveness / sub-round / result) and sort
// them into the engine's queues -- light ids, heavy units as one (unit, chunk) entry per kChunk items.
// words[2] / words[3]: the queues' fills.
template <class P>
__global__ __launch_bounds__(kBlock) void k_shard_mark(P p, const uint32_t *__restrict__ list, uint32_t n, CtrlView cv,
                                                      int32_t *__restrict__ ql, int2 *__restrict__ qh, uint32_t *words)
{
    constexpr int U = 4;
    const int lane = lane_id();
    const uint64_t tile = (uint64_t)kBlock * U;
    for (uint64_t base = (uint64_t)blockIdx.x * tile; base < n; base += (uint64_t)gridDim.x * tile) {
        bool light[U], heavy[U];
        int32_t unit[U];
        uint32_t len[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint64_t i = base + (uint64_t)k * kBlock + threadIdx.x;
            const bool valid = i < n;
            uint32_t b = 0;
            unit[k] = 0; len[k] = 0;
            if (valid) { unit[k] = (int32_t)list[i]; p.slice((uint32_t)unit[k], b, len[k]); p.mark_scanned((uint32_t)unit[k], cv); }
            light[k] = valid && len[k] <= (uint32_t)kLight;
            heavy[k] = valid && len[k] > (uint32_t)kLight;
        }
        block_append<U>(light, unit, &words[2], ql);
#pragma unroll
        for (int k = 0; k < U; ++k) {
            uint64_t mh = __ballot(heavy[k]);
            while (mh) {                                    // a heavy unit's entries are written by the whole wavefront
                const int src = __ffsll((long long)mh) - 1;
                mh &= mh - 1;
                const int32_t u = __shfl(unit[k], src);
                const uint32_t nch = ((uint32_t)__shfl((int)len[k], src) + kChunk - 1) / kChunk;
                uint32_t o = 0;
                if (lane == 0) o = atomicAdd(&words[3], nch);
                o = (uint32_t)__shfl((int)o, 0);
                for (uint32_t c = (uint32_t)lane; c < nch; c += kWave) qh[o + c] = make_int2(u, (int)c);
            }
        }
    }
}

// control block of the one PROCESS step that follows (the engine reads it at entry; its finaliser leaves the number of
// units this rank's decrements triggered in cur_light, their ids in the other light queue)
static __global__ void k_shard_setup(PeelCtrl *ctrl, const uint32_t *words, int32_t level, int32_t round, int32_t sel, int32_t launch)
{
    if (threadIdx.x == 0) {
        PeelCtrl c{};
        c.mode = MODE_PROCESS; c.level = level; c.round = round; c.done = 0;
        c.cur_light = words[2]; c.cur_heavy = words[3]; c.cur_sel = sel;
        c.remaining = 0x7FFFFFFFu;                          // the host keeps the count; the device must never decide "done"
        c.next_min = 0x7FFFFFFF;
        c.seq = launch;
        *ctrl = c;
    }
}

// the rank's own unit range: contiguous, by count
inline void shard_bounds(uint64_t units, int rank, int world, uint32_t *lo, uint32_t *hi)
{
    *lo = (uint32_t)(units * (uint64_t)rank / (uint64_t)world);
    *hi = (uint32_t)(units * (uint64_t)(rank + 1) / (uint64_t)world);
}

// exchange buffer of a sub-round: zeros, except the rank's own segment [off, off + mine) = its list
static __global__ __launch_bounds__(kBlock) void k_shard_fill(uint32_t *__restrict__ x, uint64_t total, uint64_t off, uint32_t mine, const int32_t *__restrict__ own)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock)
        x[i] = (i - off < (uint64_t)mine) ? (uint32_t)own[i - off] : 0u;
}

// hand-over to a replicated finish: the live keys are owned by range, so every rank first publishes (id, key) of its own
// live units; after the exchange every rank writes all of them into its copy of the key array
template <class P>
__global__ __launch_bounds__(kBlock) void k_shard_live(P p, uint32_t lo, uint32_t hi, uint2 *__restrict__ out, uint32_t *words)
{
    const int32_t *mk = p.scan_marker(), *ky = p.scan_key();
    const uint64_t tile = (uint64_t)kBlock * kShardU;
    for (uint64_t base = (uint64_t)lo + (uint64_t)blockIdx.x * tile; base < hi; base += (uint64_t)gridDim.x * tile) {
        bool live[kShardU];
        uint2 e[kShardU];
#pragma unroll
        for (int k = 0; k < kShardU; ++k) {
            const uint64_t u = base + (uint64_t)k * kBlock + threadIdx.x;
            live[k] = u < hi && marker_alive(mk[u]);
            e[k] = make_uint2((uint32_t)u, live[k] ? (uint32_t)ky[u] : 0u);
        }
        block_append<kShardU>(live, e, &words[0], out);
    }
}
static __global__ __launch_bounds__(kBlock) void k_shard_fill2(uint2 *__restrict__ x, uint64_t total, uint64_t off, uint32_t mine, const uint2 *__restrict__ own)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock)
        x[i] = (i - off < (uint64_t)mine) ? own[i - off] : make_uint2(0u, 0u);
}
static __global__ __launch_bounds__(kBlock) void k_shard_put(const uint2 *__restrict__ x, uint64_t total, int32_t *__restrict__ key)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
        const uint2 e = x[i];
        key[e.x] = (int32_t)e.y;
    }
}
// the control block a finish expects from a peel that hands over: `remaining` live units, no live list, the level that starts
static __global__ void k_shard_handover(PeelCtrl *ctrl, uint32_t remaining, int32_t level, int32_t round, int32_t levels, int32_t max_level, uint32_t tail_limit)
{
    if (threadIdx.x == 0) {
        PeelCtrl c{};
        c.mode = MODE_SCAN; c.level = level; c.round = round; c.done = 3; c.remaining = remaining;
        c.n_levels = levels; c.max_level = max_level; c.next_min = 0x7FFFFFFF; c.tail_limit = tail_limit; c.seq = 1;
        *ctrl = c;
    }
}

struct ShardStats {
    int32_t levels = 0, rounds = 0, scans = 0, launches = 0, max_level = 0;
    int64_t exchanges = 0, words = 0;                        // collective calls; 32-bit words they carried
    uint32_t handed_over = 0;                                // units the replicated finish took (0: the sharded peel ran to the end)
    double ms_exchange = 0.0;                                // host time inside (drain + callback), all calls
};

// The host loop.  key = the live keys (what p.scan_key() reads); `zeros` units were peeled by the init kernel (level 0), `first_level` is the smallest live key;
// launch_step(i) issues k_peel_step<sharded problem> with launch index i.  Q.light[0..1] hold `units` ids each,
// Q.heavy[0..1] the chunk entries of any frontier.
// tail_limit / finish: the hand-over to the replicated local finish (local_dev.h), with the replicated peel's own rule -- a
// level that starts with at most tail_limit units left (0: never).  The live keys are made whole on every rank first (one
// exchange of the ranks' (id, key) lists); finish() then runs the caller's hand-over on ctx->h_ctrl[0] and leaves done = 1
// there, or done = 0 and a smaller tail_limit when it declines.
template <class P, class LaunchStep, class Finish>
int shard_peel(komb_ctx *ctx, DevBufs &bufs, const P &p, int32_t *key, uint32_t units, uint32_t zeros, int32_t first_level, int rank, int world,
               komb_allreduce_fn fn, void *user, PeelQueues &Q, PeelCtrl *d_ctrl, LaunchStep &&launch_step,
               uint32_t tail_limit, Finish &&finish, ShardStats *out)
{
    hipStream_t s = ctx->stream;
    const uint32_t lo = p.lo, hi = p.hi;                     // the rank's own units (shard_bounds)
    uint32_t *d_words = nullptr, *d_hdr = nullptr, *d_xbuf = nullptr;
    const size_t xwords = std::max<size_t>((size_t)units, 2 * (size_t)std::min<uint32_t>(tail_limit, units)) + 2;
    KOMB_HIP(ctx, bufs.alloc(&d_words, 8));
    KOMB_HIP(ctx, bufs.alloc(&d_hdr, (size_t)3 * world));
    KOMB_HIP(ctx, bufs.alloc(&d_xbuf, xwords));                      // (everything is allocated before the first collective)
    std::vector<uint32_t> hdr((size_t)3 * world);
    ShardStats ss;
    if (zeros) ss.levels = 1;
    // A failure on ONE rank between two collectives must not leave the others waiting: it is recorded here, the rank keeps
    // taking part in the collectives (their sizes come from the exchanged headers), and the next header carries the status
    // word that makes every rank return KOMB_ERR_DEVICE in the same iteration.
    int local_err = KOMB_OK;
    std::string local_msg;
    auto fail_later = [&](int code, const char *what) { if (local_err == KOMB_OK) { local_err = code; local_msg = what; } };
    auto exchange = [&](uint32_t *buf, int64_t count, bool is_header) -> int {
        const auto t0 = std::chrono::steady_clock::now();
        if (world > 1) KOMB_HIP(ctx, hipStreamSynchronize(s));   // the buffer is complete when the callback runs (one rank: no callback, no drain)
        if (world > 1 && fn(user, buf, count) != 0) {
            // (a header whose own exchange failed cannot be trusted for the sizes of what follows: this rank has to leave)
            if (is_header) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "sharded peel: all-reduce callback failed");
            fail_later(KOMB_ERR_DEVICE, "sharded peel: all-reduce callback failed");
        }
        ss.ms_exchange += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ++ss.exchanges; ss.words += count;
        return KOMB_OK;
    };
    // counts (and minima) of all ranks -> hdr[]; returns the total and this rank's offset in the concatenation
    auto exchange_counts = [&](const uint32_t *cnt, const uint32_t *mn, uint64_t *total, uint64_t *my_off, int32_t *gmin) -> int {
        k_shard_header<<<1, 64, 0, s>>>(d_hdr, world, rank, cnt, mn, d_words, local_err != KOMB_OK ? 1u : 0u);
        KOMB_TRY(exchange(d_hdr, (int64_t)3 * world, true));
        KOMB_HIP(ctx, d2h(ctx, hdr.data(), d_hdr, hdr.size() * sizeof(uint32_t)));
        *total = 0; *my_off = 0; *gmin = 0x7FFFFFFF;
        for (int r = 0; r < world; ++r) {
            if (r == rank) *my_off = *total;
            *total += hdr[(size_t)r];
            *gmin = std::min(*gmin, (int32_t)hdr[(size_t)world + r]);
        }
        for (int r = 0; r < world; ++r)
            if (hdr[(size_t)2 * world + r]) {
                if (r == rank || local_err != KOMB_OK) KOMB_FAIL(ctx, local_err != KOMB_OK ? local_err : KOMB_ERR_DEVICE, "%s", local_msg.empty() ? "sharded peel: failure on this rank" : local_msg.c_str());
                KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "sharded peel: rank %d reported a failure; all ranks stop", r);
            }
        return KOMB_OK;
    };
    auto grid_of = [](uint64_t n) { return (int)std::min<uint64_t>((n + kBlock - 1) / kBlock + 1, 2048); };
    auto sweep_grid = [](uint64_t n, int per_thread) { const uint64_t t = (uint64_t)kBlock * per_thread; return (int)std::min<uint64_t>((n + t - 1) / t + 1, 2048); };
    const int scan_grid = sweep_grid((uint64_t)(hi - lo), kShardU);
    const uint32_t *cur_light_word = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(d_ctrl) + offsetof(PeelCtrl, cur_light));
    uint64_t remaining = (uint64_t)units - zeros;
    int32_t L = first_level, round = 1, launch = 0;
    int own_q = 0;                                           // Q.light[own_q] holds this rank's part of the next frontier
    bool after_scan = false, level_counted = false;
    auto scan = [&]() -> int {
        k_shard_scan_reset<<<1, 64, 0, s>>>(d_words);
        k_shard_scan<P><<<scan_grid, kBlock, 0, s>>>(p, lo, hi, L, Q.light[own_q], d_words);
        ++ss.scans;
        after_scan = true; level_counted = false;
        return KOMB_OK;
    };
    // a level starts with `remaining` <= tail_limit units: make the live keys whole everywhere, then the caller's finish
    auto hand_over = [&](bool *finished, bool publish_keys) -> int {
        uint64_t total = 0, my_off = 0;
        int32_t gmin = 0;
        if (world > 1 && publish_keys) {
            uint2 *own = reinterpret_cast<uint2 *>(Q.light[own_q]);          // (`units` words = units / 2 pairs >= tail_limit >= the live units)
            k_shard_scan_reset<<<1, 64, 0, s>>>(d_words);
            k_shard_live<P><<<scan_grid, kBlock, 0, s>>>(p, lo, hi, own, d_words);
            KOMB_TRY(exchange_counts(d_words, nullptr, &total, &my_off, &gmin));
            if (total != remaining) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "sharded peel: %llu live units found at the hand-over, %llu expected", (unsigned long long)total, (unsigned long long)remaining);
            k_shard_fill2<<<grid_of(total), kBlock, 0, s>>>(reinterpret_cast<uint2 *>(d_xbuf), total, my_off, hdr[(size_t)rank], own);
            KOMB_TRY(exchange(d_xbuf, (int64_t)(2 * total), false));
            k_shard_put<<<grid_of(total), kBlock, 0, s>>>(reinterpret_cast<const uint2 *>(d_xbuf), total, key);
        }
        k_shard_handover<<<1, 64, 0, s>>>(d_ctrl, (uint32_t)remaining, L, round, ss.levels, ss.max_level, tail_limit);
        KOMB_HIP(ctx, d2h(ctx, &ctx->h_ctrl[0], d_ctrl, sizeof(PeelCtrl)));
        {
            // the finish runs on every rank (allocations, a fixed point): the ranks agree on its outcome before any of them goes on
            const int frc = finish();
            if (frc != KOMB_OK) fail_later(frc, ctx->err.c_str());
            if (world > 1) {
                uint64_t t2 = 0, o2 = 0; int32_t g2 = 0;
                KOMB_TRY(exchange_counts(nullptr, nullptr, &t2, &o2, &g2));
            } else if (frc != KOMB_OK) return frc;
        }
        const PeelCtrl &hc = ctx->h_ctrl[0];
        if (hc.done == 1) {
            ss.handed_over = (uint32_t)remaining;
            ss.levels = hc.n_levels; ss.max_level = hc.max_level;
            remaining = 0;
            *finished = true;
        } else {
            tail_limit = hc.tail_limit < tail_limit ? hc.tail_limit : 0u;     // declined: offered again at the smaller size it named (or never)
            *finished = false;
        }
        return KOMB_OK;
    };
    bool finished = false;
    // a small input goes to the finish whole: nothing has been decremented yet, every rank's keys are whole
    if (remaining && tail_limit && (uint64_t)units <= tail_limit) KOMB_TRY(hand_over(&finished, false));
    // (later hand-overs: the ranks' (id, key) pairs live in a light queue of `units` words and in d_xbuf)
    if (tail_limit > units / 2) tail_limit = units / 2;
    if (remaining) KOMB_TRY(scan());
    const uint64_t max_iter = 4ull * units + 4096;
    for (uint64_t it = 0; remaining > 0; ++it) {
        if (it > max_iter) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "sharded peel: no progress");
        // ---- exchange 1: counts (and the ranks' minima after a SCAN)
        uint64_t total = 0, my_off = 0;
        int32_t gmin = 0x7FFFFFFF;
        KOMB_TRY(exchange_counts(after_scan ? d_words : cur_light_word, after_scan ? d_words + 1 : nullptr, &total, &my_off, &gmin));
        if (total > remaining) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "sharded peel: %llu frontier units of %llu left", (unsigned long long)total, (unsigned long long)remaining);
        if (total == 0) {
            // the level is exhausted on every rank: the next one, or -- straight after a SCAN that found nothing -- the
            // smallest live key anywhere
            if (after_scan) {
                if (gmin == 0x7FFFFFFF) KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "sharded peel: %llu units left but none is live", (unsigned long long)remaining);
                L = gmin;
            } else {
                L += 1;
                if (tail_limit && remaining <= tail_limit) {
                    KOMB_TRY(hand_over(&finished, true));
                    if (finished) break;
                }
            }
            KOMB_TRY(scan());
            continue;
        }
        if (!level_counted) { ++ss.levels; ss.max_level = L; level_counted = true; }
        // ---- exchange 2: the ids, each rank's list in its own segment of a zeroed buffer
        k_shard_fill<<<grid_of(total), kBlock, 0, s>>>(d_xbuf, total, my_off, hdr[(size_t)rank], Q.light[own_q]);
        if (world > 1) KOMB_TRY(exchange(d_xbuf, (int64_t)total, false));
        // ---- every rank: stamp + classify the whole frontier, then one PROCESS step that keeps the rank's own decrements
        CtrlView cv{};
        cv.mode = MODE_PROCESS; cv.level = L; cv.round = round; cv.cur_sel = own_q;
        k_shard_mark<P><<<sweep_grid(total, 4), kBlock, 0, s>>>(p, d_xbuf, (uint32_t)total, cv, Q.light[own_q], Q.heavy[own_q], d_words);
        ++launch;
        k_shard_setup<<<1, 64, 0, s>>>(d_ctrl, d_words, L, round, own_q, launch);
        launch_step(launch);
        ++ss.launches; ++ss.rounds;
        own_q ^= 1;                                          // the units this rank's decrements triggered are in the other light queue
        after_scan = false;
        ++round;
        remaining -= total;
    }
    KOMB_HIP(ctx, hipStreamSynchronize(s));
    bufs.release(d_words); bufs.release(d_hdr); bufs.release(d_xbuf);
    *out = ss;
    return KOMB_OK;
}

} // namespace komb
