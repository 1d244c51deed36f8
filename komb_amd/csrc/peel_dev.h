// peel_dev.h -- device-side helpers shared by the k-core and k-truss peels.
#pragma once

#include "common.h"

namespace komb {

enum : int32_t { MODE_SCAN = 0, MODE_PROCESS = 1 };

struct CtrlView {                       // what one launch needs from the control block
    int32_t mode, level, round, done, cur_sel;
    uint32_t cur_count;
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ uint64_t lanemask_lt()
{
    return (1ull << lane_id()) - 1ull;
}

// Coherent read of a word other workgroups update with atomics in this launch.
__device__ __forceinline__ uint32_t coherent_load(uint32_t *p) { return atomicAdd(p, 0u); }
__device__ __forceinline__ int32_t coherent_load(int32_t *p) { return atomicAdd(p, 0); }

// Every workgroup loads the launch-stable fields once.
__device__ __forceinline__ CtrlView load_ctrl(const PeelCtrl *ctrl, CtrlView *sh)
{
    if (threadIdx.x == 0) {
        sh->mode = ctrl->mode; sh->level = ctrl->level; sh->round = ctrl->round;
        sh->done = ctrl->done; sh->cur_sel = ctrl->cur_sel; sh->cur_count = ctrl->cur_count;
    }
    __syncthreads();
    return *sh;
}

// Wave-aggregated append of `val` (for lanes with `pred`) to queue `q` whose
// cursor is `tail`: one atomic per wave, order inside the wave preserved.
// Must be reached by all 64 lanes.
__device__ __forceinline__ void wave_append(bool pred, int32_t val, int32_t *q, uint32_t *tail)
{
    const uint64_t m = __ballot(pred);
    if (m == 0) return;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane_id() == leader) base = atomicAdd(tail, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader);
    if (pred) q[base + (uint32_t)__popcll(m & lanemask_lt())] = val;
}

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

// The last workgroup to finish a launch rewrites the control block for the
// next launch.  remaining counts units that have not entered a frontier yet:
// SCAN subtracts what it collected, PROCESS subtracts what it triggered (the
// next queue's length plus ctrl->acc, the units a wave peeled on the spot).
__device__ __forceinline__ void finalize_launch(PeelCtrl *ctrl, const CtrlView &cv)
{
    __syncthreads();
    if (threadIdx.x != 0) return;
    __threadfence();
    const uint32_t ticket = atomicAdd(&ctrl->blocks_done, 1u);
    if (ticket != gridDim.x - 1) return;
    __threadfence();
    int32_t mode = cv.mode, level = cv.level, round = cv.round, done = 0, sel = cv.cur_sel;
    uint32_t cur = cv.cur_count, remaining = ctrl->remaining;
    if (mode == MODE_SCAN) {
        ctrl->n_scans += 1;
        const uint32_t cnt = coherent_load(&ctrl->tail[sel]);
        if (cnt > 0) {
            mode = MODE_PROCESS; cur = cnt; remaining -= cnt;
            ctrl->n_levels += 1; ctrl->max_level = level;
        } else if (remaining == 0) {
            done = 1;
        } else {
            level = coherent_load(&ctrl->next_min);        // first populated level above
            if (level == 0x7FFFFFFF) done = 2;             // live units but no live key: inconsistent state
        }
        atomicExch(&ctrl->next_min, 0x7FFFFFFF);
    } else {
        ctrl->n_rounds += 1;
        const uint32_t ncnt = coherent_load(&ctrl->tail[sel ^ 1]);
        remaining -= coherent_load(&ctrl->acc) + ncnt;
        atomicExch(&ctrl->acc, 0u);
        atomicExch(&ctrl->tail[sel], 0u);
        sel ^= 1; cur = ncnt; round += 1;
        if (ncnt == 0) {
            if (remaining == 0) done = 1;
            else { level += 1; mode = MODE_SCAN; }
        }
    }
    ctrl->mode = mode; ctrl->level = level; ctrl->round = round; ctrl->done = done;
    ctrl->cur_sel = sel; ctrl->cur_count = cur; ctrl->remaining = remaining;
    atomicExch(&ctrl->blocks_done, 0u);
    __threadfence();
}

} // namespace komb
