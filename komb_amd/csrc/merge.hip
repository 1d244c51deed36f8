// merge.hip -- rows a12 + a13 of the hot-path table: the indexed min-heap of the reference
// (src/HashIndexedMinHeap.h:10-238) and its only user, CombineCoreA::runMerge (src/CombineCoreA.h:45-219), the greedy
// densest-block peel over a "row" copy and a "column" copy of the graph.  Dead code in the reference (no caller); built
// here, last, because SURVEY section 8 lists it and its results can be pinned: the reference's heap class is std-only,
// so the tests run it compiled in place, and it decides every tie.
//
// What the algorithm is: every node starts on both sides with priority = suspiciousness + degree; each of the 2n steps
// removes the smaller of the two heap minima (the row side only when strictly smaller, or the column side is empty),
// subtracts its priority from the running sum, records density = sum / nodes left, and lowers by one the neighbours of
// the removed node that are still present on the OTHER side.  WHICH of several equal priorities is the minimum is decided
// by the heap's array layout -- and priorities are "score + integer", so ties are the rule: a result that is to equal
// the reference's must run the same sift operations in the same order.  Each step depends on the one before it; there is
// no parallelism to find that keeps the ties (the level-synchronous peels of this library settle UNIQUE integers, which
// is why they may reorder freely; this one does not).
//
// So the device version is what it can be: the exact heaps, operated by ONE lane, in LDS when both fit (n <= kMergeLds:
// a step costs a few LDS round trips) and in global memory otherwise (a step costs ~2 log n dependent L2 / HBM round
// trips: ~10 us at 10^5 nodes, ~25 us at 10^7 -- minutes for a 10 M-node graph; the reference's CPU loop is faster there,
// and nobody calls either).  The priorities are set up in parallel, with the same floating-point operations as the
// reference performs per node (score, then +1.0 once per incident slot).
#include "peel_dev.h"

namespace komb {

namespace {

constexpr int kMergeLds = 4096;                 // nodes up to which both heaps live in LDS (2 x (4 + 4 + 8) bytes each + flags)
constexpr int kMergeMaxNodes = 1 << 17;         // above it komb_densest_block returns KOMB_ERR_LIMIT (a few seconds of one lane's time at most)
static_assert(2 * kMergeLds * (8 + 4 + 4 + 1) <= 160 * 1024, "the LDS heaps are sized for gfx950's 160 KB of LDS per workgroup");

// priorities as the reference builds them (src/CombineCoreA.h:52-85): the score (or 0), then one +1.0 per incident slot
__global__ __launch_bounds__(kBlock) void k_merge_prio(const uint32_t *__restrict__ rowptr, int64_t nv, const double *__restrict__ susp,
                                                       double *__restrict__ prio)
{
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += (int64_t)gridDim.x * kBlock) {
        double p = susp ? susp[v] : 0.0;
        const uint32_t d = rowptr[v + 1] - rowptr[v];
        for (uint32_t k = 0; k < d; ++k) p += 1.0;
        prio[v] = p;
    }
}

struct Heap {                                   // src/HashIndexedMinHeap.h: array of ids, id -> position, id -> value
    int32_t *arr;
    int32_t *pos;
    double *val;
    int32_t size;
    __device__ bool down(int32_t p)             // minHeapfy (:171-217), iterative: the same swaps
    {
        bool moved = false;
        for (;;) {
            const int32_t l = 2 * (p + 1) - 1, r = 2 * (p + 1);
            const int32_t cur = arr[p];
            int32_t sp = p, sk = cur;
            if (l < size) { const int32_t kl = arr[l]; if (val[kl] < val[cur]) { sp = l; sk = kl; } }
            if (r < size) { const int32_t kr = arr[r]; if (val[kr] < val[sk]) { sp = r; sk = kr; } }
            if (sp == p) return moved;
            arr[p] = sk; pos[sk] = p;
            arr[sp] = cur; pos[cur] = sp;
            p = sp; moved = true;
        }
    }
    __device__ void refresh(int32_t key, double v)      // refreshPriority (:138-167)
    {
        val[key] = v;
        int32_t p = pos[key];
        if (down(p) || p <= 0) return;
        int32_t pp = (p + 1) / 2 - 1;
        while (p > 0 && val[arr[pp]] > val[key]) {
            const int32_t pe = arr[pp];
            arr[pp] = key; pos[key] = pp;
            arr[p] = pe; pos[pe] = p;
            p = pp; pp = (p + 1) / 2 - 1;
        }
    }
    __device__ void insert(int32_t key, double v)       // insert (:83-98)
    {
        const int32_t p = size++;
        arr[p] = key; pos[key] = p; val[key] = v;
        refresh(key, v);
    }
    __device__ int32_t poll(double &v)                  // poll (:55-81), size > 0
    {
        const int32_t top = arr[0];
        v = val[top];
        pos[top] = -1;
        if (size != 1) {
            const int32_t last = arr[size - 1];
            arr[0] = last; pos[last] = 0;
            --size;
            down(0);
        } else --size;
        arr[size] = 0;
        return top;
    }
};

struct MergeOut { int32_t n_block; int32_t pad; double max_density; };

// one workgroup of one wavefront; lane 0 runs the peel.  LDS: both heaps and the two "gone" flag arrays in LDS.
template <bool LDS>
__global__ __launch_bounds__(kWave) void k_merge_run(const uint32_t *__restrict__ rowptr, const int32_t *__restrict__ col, int32_t n,
                                                     const double *__restrict__ susp, const double *__restrict__ prio,
                                                     int32_t *g_arr, int32_t *g_pos, double *g_val, uint8_t *g_gone,
                                                     int32_t *__restrict__ order, int32_t *__restrict__ side, MergeOut *__restrict__ out)
{
    __shared__ double sh_val[LDS ? 2 * kMergeLds : 1];
    __shared__ int32_t sh_arr[LDS ? 2 * kMergeLds : 1];
    __shared__ int32_t sh_pos[LDS ? 2 * kMergeLds : 1];
    __shared__ uint8_t sh_gone[LDS ? 2 * kMergeLds : 1];
    const int32_t cap = LDS ? kMergeLds : n;
    int32_t *arr = LDS ? sh_arr : g_arr, *pos = LDS ? sh_pos : g_pos;
    double *val = LDS ? sh_val : g_val;
    uint8_t *gone = LDS ? sh_gone : g_gone;
    // (parallel) empty heaps, nobody gone
    for (int32_t i = (int32_t)threadIdx.x; i < 2 * cap; i += kWave) { if (i % cap < n) { pos[i] = -1; gone[i] = 0; } }
    __syncthreads();
    if (threadIdx.x != 0) return;
    Heap h[2] = {{arr, pos, val, 0}, {arr + cap, pos + cap, val + cap, 0}};
    // the running sum, in the reference's order of operations (src/CombineCoreA.h:63-67, :87)
    double sum = 0;
    if (susp) for (int32_t v = 0; v < n; ++v) sum += 2 * susp[v];
    sum += (double)(long)rowptr[n];
    for (int32_t v = 0; v < n; ++v) h[0].insert(v, prio[v]);        // :89-93
    for (int32_t v = 0; v < n; ++v) h[1].insert(v, prio[v]);        // :95-99 (the column degrees equal the row degrees: the graph is symmetric)
    double best = 0;
    int32_t best_left = 0;
    for (int32_t left = 2 * n; left >= 1;) {                        // :114-174
        const int s = (h[0].size > 0 && (h[1].size == 0 || h[0].val[h[0].arr[0]] < h[1].val[h[1].arr[0]])) ? 0 : 1;
        double pv;
        const int32_t node = h[s].poll(pv);
        sum -= pv;
        --left;
        order[left] = node; side[left] = s;
        if (left >= 1) { const double d = sum / left; if (d > best) { best = d; best_left = left; } }
        gone[s * cap + node] = 1;
        Heap &o = h[s ^ 1];
        const uint8_t *og = gone + (s ^ 1) * cap;
        for (uint32_t j = rowptr[node]; j < rowptr[node + 1]; ++j) {
            const int32_t w = col[j];
            if (!og[w]) o.refresh(w, o.val[w] - 1);
        }
    }
    out->n_block = best_left;
    out->max_density = best;
}

} // namespace

int merge_run(komb_ctx *ctx, const double *susp_host, int32_t *order, int32_t *side, int64_t *n_block, double *max_density)
{
    if (ctx->nv < 0) KOMB_FAIL(ctx, KOMB_ERR_STATE, "komb_densest_block: no graph loaded");
    const int64_t nv = ctx->nv;
    if (nv > 0 && (!order || !side)) KOMB_FAIL(ctx, KOMB_ERR_ARG, "komb_densest_block: null output");
    // Above kMergeLds nodes the two heaps live in global memory and ONE lane walks them: ~35 us per step, 2 nv steps -- 0.7 s for
    // 20 000 vertices, minutes for millions, on a stream nothing can cancel.  The exact tie order makes the peel sequential
    // (DESIGN.md section 9), so large graphs are refused rather than left to occupy the device.
    if (nv > kMergeMaxNodes) KOMB_FAIL(ctx, KOMB_ERR_LIMIT, "komb_densest_block: %lld nodes; the exact (sequential) densest-block peel is limited to %d", (long long)nv, kMergeMaxNodes);
    if (n_block) *n_block = 0;
    if (max_density) *max_density = 0.0;
    if (nv == 0) return KOMB_OK;
    Range r_all("komb_densest_block");
    hipStream_t s = ctx->stream;
    DevBufs bufs(ctx);
    double *d_susp = nullptr, *d_prio = nullptr, *d_val = nullptr;
    int32_t *d_arr = nullptr, *d_pos = nullptr, *d_order = nullptr, *d_side = nullptr;
    uint8_t *d_gone = nullptr;
    MergeOut *d_out = nullptr;
    const bool lds = nv <= kMergeLds;
    KOMB_HIP(ctx, bufs.alloc(&d_prio, (size_t)nv));
    KOMB_HIP(ctx, bufs.alloc(&d_order, (size_t)2 * nv));
    KOMB_HIP(ctx, bufs.alloc(&d_side, (size_t)2 * nv));
    KOMB_HIP(ctx, bufs.alloc(&d_out, 1));
    if (susp_host) {
        KOMB_HIP(ctx, bufs.alloc(&d_susp, (size_t)nv));
        KOMB_HIP(ctx, hipMemcpyAsync(d_susp, susp_host, (size_t)nv * sizeof(double), hipMemcpyHostToDevice, s));
    }
    if (!lds) {
        KOMB_HIP(ctx, bufs.alloc(&d_arr, (size_t)2 * nv));
        KOMB_HIP(ctx, bufs.alloc(&d_pos, (size_t)2 * nv));
        KOMB_HIP(ctx, bufs.alloc(&d_val, (size_t)2 * nv));
        KOMB_HIP(ctx, bufs.alloc(&d_gone, (size_t)2 * nv));
    }
    int64_t g = (nv + kBlock - 1) / kBlock;
    k_merge_prio<<<(int)(g > 4096 ? 4096 : g), kBlock, 0, s>>>(ctx->d_o_rowptr, nv, d_susp, d_prio);
    if (lds) k_merge_run<true><<<1, kWave, 0, s>>>(ctx->d_o_rowptr, ctx->d_o_col, (int32_t)nv, d_susp, d_prio, nullptr, nullptr, nullptr, nullptr, d_order, d_side, d_out);
    else k_merge_run<false><<<1, kWave, 0, s>>>(ctx->d_o_rowptr, ctx->d_o_col, (int32_t)nv, d_susp, d_prio, d_arr, d_pos, d_val, d_gone, d_order, d_side, d_out);
    MergeOut ho{};
    KOMB_HIP(ctx, d2h(ctx, &ho, d_out, sizeof(ho)));
    KOMB_HIP(ctx, hipMemcpy(order, d_order, (size_t)2 * nv * sizeof(int32_t), hipMemcpyDeviceToHost));
    KOMB_HIP(ctx, hipMemcpy(side, d_side, (size_t)2 * nv * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n_block) *n_block = ho.n_block;
    if (max_density) *max_density = ho.max_density;
    return KOMB_OK;
}

} // namespace komb
