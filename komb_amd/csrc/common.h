// common.h -- shared host-side definitions of libkomb_accel (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "komb_accel.h"

#include <roctracer/roctx.h>

namespace komb {

// roctx range around a phase of the path (rocprofv3 --marker-trace shows them)
struct Range {
    explicit Range(const char *name) { (void)roctxRangePushA(name); }
    ~Range() { (void)roctxRangePop(); }
    void next(const char *name) { (void)roctxRangePop(); (void)roctxRangePushA(name); }   // one phase ends, the next begins
    Range(const Range &) = delete;
    Range &operator=(const Range &) = delete;
};

constexpr int kBlock = 256;                 // 4 wave64 per workgroup
constexpr int kWave = 64;
// stamp / core value of a live edge / vertex: kAlive - c, where c = 0 for a unit with a short slice
// ("light") and c = its number of kChunk-item chunks otherwise.  Whoever reads the marker to test
// liveness (the SCAN sweep, the item that decrements the unit) learns the unit's class for free,
// instead of fetching its slice bounds.
constexpr int32_t kAlive = 0x7FFFFFFF;
constexpr int32_t kAliveMin = 0x40000000;   // every alive marker is >= this; every round / level number is below it

// Device-side control block of a peel loop.  One 128-byte record; the fields a
// launch reads at entry are written only by the workgroup that finalises a step.
// `seq` is the index of the launch the state is meant for: every launch carries its
// own index, a workgroup acts only when the two agree before AND after it has read
// the state, and a finaliser moves `seq` on before it touches anything else -- so a
// workgroup that is dispatched late (another process shares the GPU, say), after the
// state has already been rewritten for the next launch, leaves instead of taking part
// in a step that is not its launch's.
struct PeelCtrl {
    // ---- first 64-byte line: the state a launch reads at entry (one coalesced load = one snapshot) and its sequence word.
    // Written by the finalising workgroup only; the counters in it are statistics.
    int32_t  mode;          // 0 = SCAN, 1 = PROCESS
    int32_t  level;         // current peel level (degree k / support L)
    int32_t  round;         // sub-round id stamped on the current frontier
    int32_t  done;          // 1 once every unit is peeled (2 = inconsistent state, 3 = handed over, see tail_limit)
    uint32_t cur_light;     // entries in the current light queue (unit ids)
    uint32_t cur_heavy;     // entries in the current heavy queue ((unit, chunk) pairs)
    int32_t  cur_sel;       // which of the two queue pairs is current
    uint32_t remaining;     // units that have not entered a frontier yet
    int32_t  n_levels;      // stats: populated levels
    int32_t  n_rounds;      // stats: PROCESS launches
    int32_t  n_scans;       // stats: SCAN launches
    int32_t  max_level;     // stats: highest populated level
    uint32_t live_count;    // entries of the compacted live list (live_mode 1)
    int32_t  live_sel;      // which live-list buffer is current
    int32_t  live_mode;     // 0: SCAN sweeps all units; 1: SCAN sweeps the live list
    int32_t  seq;           // index of the launch this state is for (see above)
    // ---- second line: modified with atomics during a launch
    uint32_t tail_l[2];     // append cursors of the light queues
    uint32_t tail_h[2];     // append cursors of the heavy queues
    int32_t  next_min;      // min live key above the scanned level
    uint32_t live_tail;     // survivors appended by the running SCAN
    int32_t  last_retire;   // problems with one-byte states: the sub-round the last RETIRE step ran before (finaliser only)
    // ---- hand-over to a finish (local_dev.h, truss_tail.h, core_tail.h); constant while launches are queued
    uint32_t tail_limit;    // a level that starts with at most this many units left sets done = 3 (0: never)
    uint32_t pad1[7];       // (-DKOMB_STEP_TIMERS: stopwatch sums)
    int32_t  max_retire_gap;    // the most sub-rounds that ever passed without a RETIRE step: the host checks it against the period
};
static_assert(offsetof(PeelCtrl, seq) == 60 && offsetof(PeelCtrl, tail_l) == 64, "PeelCtrl: the entry state is one 64-byte line");
static_assert(sizeof(PeelCtrl) == 128, "PeelCtrl layout");

// Device-side control block of the local finish (local_dev.h): an h-index fixed point on the remainder
// the peel hands over.  64 bytes, device resident; the host keeps two pinned mirrors.
struct LocalCtrl {
    int32_t  done;           // 1 once a sweep changed nothing
    int32_t  iters;          // sweeps run (including the one that changed nothing)
    uint32_t spare[3];       // [0]: the sweep that is running (progress word for the host's guard); the per-sweep change counters live in their own array
    uint32_t bad;            // consistency failures (a compact slice whose fill differs from the live key, ...)
    int32_t  max_val;        // largest final value (k_local_finish)
    uint32_t levels;         // distinct final values (k_local_levels)
    uint32_t evals;          // unit evaluations, all sweeps
    uint32_t n_light, n_heavy;   // numbering counters (k_local_number)
    uint32_t n_giant;        // heavy units with more than kMedMax items
    uint32_t n_chunk;        // chunks of kGiantChunk items they are counted in
    uint32_t pad0;
    unsigned long long key_sum;  // sum of the live keys = items of the remainder (k_local_number)
};
static_assert(sizeof(LocalCtrl) == 64, "LocalCtrl layout");

// How a peel ends (option FINISH, komb_set_option): "local" hands the remainder to the h-index fixed point of local_dev.h
// once at most LOCAL_LIMIT units (default: a fraction of all units) are left at a level boundary and it has at most
// LOCAL_ITEMS items (else it is refused and offered again later); "lds" uses the single-workgroup LDS tails
// (truss_tail.h, core_tail.h; thresholds TAIL / CORE_TAIL); "none" keeps the whole peel in the general
// engine.  None of them changes a result.  Default for both peels: "local" (MI355X: k-core |V| = 1M 5.3 -> 3.0 ms,
// |V| = 10M 16.8 -> 10.7 ms; k-truss peel |E| = 10M 3.7 -> 2.9 ms, |E| = 100M 13.0 -> 11.1 ms against "lds").
enum FinishMode : int { FIN_LOCAL = 0, FIN_LDS = 1, FIN_NONE = 2 };

struct Timer {                               // HIP-event stopwatch on one stream
    hipEvent_t a = nullptr, b = nullptr;
    bool init() { return hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess; }
    void start(hipStream_t s) { (void)hipEventRecord(a, s); }
    double stop(hipStream_t s)
    {
        (void)hipEventRecord(b, s);
        (void)hipEventSynchronize(b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, a, b);
        return (double)ms;
    }
    void destroy() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); a = b = nullptr; }
};

// HIP events that are destroyed on every way out of the scope that made them (error returns included)
struct EventSet {
    std::vector<hipEvent_t> v;
    EventSet() = default;
    EventSet(const EventSet &) = delete;
    EventSet &operator=(const EventSet &) = delete;
    hipError_t make(hipEvent_t *out, unsigned flags = 0)
    {
        hipEvent_t e = nullptr;
        const hipError_t rc = flags ? hipEventCreateWithFlags(&e, flags) : hipEventCreate(&e);
        if (rc == hipSuccess) { v.push_back(e); *out = e; }
        return rc;
    }
    ~EventSet() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); }
};

} // namespace komb

// Caching device allocator: a step's scratch buffers are returned to the pool,
// not to the driver, so steady-state steps perform no hipMalloc / hipFree.
struct DevPool {
    struct Block { void *p; size_t bytes; bool used; };
    std::vector<Block> blocks;
    size_t n_malloc = 0, n_trim = 0;                 // statistics (KOMB_POOL_DEBUG)
    double ms_malloc = 0.0;                          // host time inside hipMalloc
    hipError_t get(void **out, size_t bytes)
    {
        if (bytes == 0) bytes = 16;
        int best = -1;
        for (int i = 0; i < (int)blocks.size(); ++i)
            if (!blocks[i].used && blocks[i].bytes >= bytes && blocks[i].bytes <= bytes + bytes / 4 + 4096 &&
                (best < 0 || blocks[i].bytes < blocks[best].bytes)) best = i;
        if (best >= 0) { blocks[best].used = true; *out = blocks[best].p; return hipSuccess; }
        void *q = nullptr;
        ++n_malloc;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&q, bytes);
        ms_malloc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (e != hipSuccess) {                       // give cached blocks back and retry once
            ++n_trim;
            trim();
            e = hipMalloc(&q, bytes);
            if (e != hipSuccess) return e;
        }
        blocks.push_back({q, bytes, true});
        *out = q;
        return hipSuccess;
    }
    void put(void *p)
    {
        if (!p) return;
        for (auto &b : blocks) if (b.p == p) { b.used = false; return; }
    }
    size_t unused_bytes() const                      // cached blocks nobody holds: a large request gets them back (trim) before it fails
    {
        size_t t = 0;
        for (const auto &b : blocks) if (!b.used) t += b.bytes;
        return t;
    }
    void trim()                                      // free every unused block
    {
        std::vector<Block> keep;
        for (auto &b : blocks) { if (b.used) keep.push_back(b); else (void)hipFree(b.p); }
        blocks.swap(keep);
    }
    void clear()
    {
        for (auto &b : blocks) (void)hipFree(b.p);
        blocks.clear();
    }
};

// Everything the k-truss path derives from a symmetric CSR before it enumerates a triangle (truss_prep.hip): the
// (degree,id) renumbering, the oriented CSR in those INTERNAL ids, the canonical edge list and the internal edge id of
// every canonical edge, the per-vertex lines and the task table of the enumeration.  Built on the first k-truss call
// of a graph (or by komb_truss_prepare), kept until the graph goes (or komb_truss_unprepare); an induced subgraph
// gets a temporary one of its own.  Every array is a block of the context's pool.
struct TrussPrep {
    bool valid = false;
    int64_t nv = 0, ne = 0;
    int32_t  *o2i = nullptr, *i2o = nullptr; // [nv] original -> internal id (rank in (degree, original id) order) and back
    uint32_t *orow = nullptr;                // [nv+1]  oriented CSR, INTERNAL ids, rows ascending (source below target): internal edge id = oriented slot
    int32_t  *ocol = nullptr, *osrc = nullptr;   // [ne + 8] target of every oriented slot; [ne] its source -- only once prep_sources was asked (LDS tail, moments)
    uint32_t *e2k = nullptr;                 // [ne] canonical edge id of every internal edge (oriented slot)
    uint4    *vline = nullptr;               // [4*nv] one 64-byte line per vertex: start, length, pivots and signature of its oriented row (truss_line.h)
    void     *wtasks = nullptr;              // [n_wtasks] task descriptors of the triangle enumeration (truss_line.h)
    int64_t   n_wtasks = 0;
    int64_t   own_bound = 0;                 // sum over the vertices of d+ (d+ - 1): bound on the own-role index entries (capacities of a k-truss run)
    double    ms = 0.0;                      // device time of the build (HIP events) ...
    double    ms_part[4] = {0, 0, 0, 0};     // ... and of its parts: vertex order | edges to rows + row pointers | row sort + lines + canonical map | tasks
};

struct komb_ctx {
    komb_opts opts{};
    DevPool pool;
    std::string err;
    bool device_ok = false;
    int device = 0;
    hipStream_t stream = nullptr;            // the context's own blocking stream (api.cpp)
    bool own_stream = false;
    komb::Timer timer;
    std::map<std::string, std::string> options;   // komb_set_option: tuning / test switches (none changes a result)

    // ---- resident simple graph: the symmetric CSR in the caller's (ORIGINAL) vertex ids, rows ascending -- what every result
    // is reported in, what k-core peels on, what komb_graph_get_csr returns
    int64_t nv = -1, ne = 0;
    uint32_t *d_o_rowptr = nullptr;          // [nv+1]
    int32_t  *d_o_col = nullptr;             // [2*ne]
    // ---- the k-truss side of the graph, made when a k-truss call first needs it
    TrussPrep prep;

    // ---- k-core results
    int32_t *d_deg = nullptr;                // [nv] degree (a2), ORIGINAL ids
    int32_t *d_core = nullptr;               // [nv] coreness (a3), ORIGINAL ids
    bool core_done = false;

    // ---- k-truss results (canonical order)
    int64_t t_ne = -1;                       // edges of the (sub)graph last run
    int32_t *d_t_eu = nullptr, *d_t_ev = nullptr, *d_t_truss = nullptr, *d_t_sup = nullptr;
    uint2 *d_t_slice = nullptr;              // [t_ne] (start, length) of every internal edge's index slice: the length is the support the peel started from; komb_truss_fetch_support puts them in canonical
    uint32_t t_k_lo = 0, t_k_hi = 0;         // the canonical edges the last run materialised (komb_truss_run_slice: this rank's slice)
    bool t_sup_ready = false;                // order on its first call (d_t_sup; igraph_trussness has no such output, and the timed step does not make it)
    bool t_own_edges = false;                // d_t_eu / d_t_ev are pool blocks of this result (induced subgraph), not the graph's cached list
    int32_t *d_ceu = nullptr, *d_cev = nullptr;   // the resident graph's canonical edge list, made by the first komb_truss_fetch that asks for endpoints (pool blocks)
    bool truss_done = false;
    int slice_rank = 0, slice_world = 1;     // komb_truss_run_slice: the canonical edges whose results this run materialises
    bool shard_peel = false;                 // komb_set_shard_peel: sharded runs split the peel too (shard_dev.h)
    // what the record stream of the last k-truss run on a graph of this size turned out to need (ktruss.hip): a graph with more
    // than two triangles per edge pays the second enumeration once per graph, not once per run
    struct { int64_t nv = -1, m = -1; unsigned long long own_cap = 0, rec_cap = 0; } cap_hint;

    // ---- pinned host mirrors of the control blocks (double buffered)
    komb::PeelCtrl *h_ctrl = nullptr;        // [2]
    komb::LocalCtrl *h_local = nullptr;      // [2]
    void *h_stage = nullptr;                 // pinned landing area of the small device-to-host reads (kStageBytes)
    struct H2DStager *stager = nullptr;      // pinned staging buffers + streams of the graph upload (graph_build.hip), made on first use

    komb_stats stats{};
};

// ---- options (komb_set_option): what the library used to read from KOMB_* environment variables.  The drop-in never sets
// any; tests and measurements do, explicitly, per context.
inline const char *ctx_opt(const komb_ctx *ctx, const char *name)
{
    const auto it = ctx->options.find(name);
    return it == ctx->options.end() ? nullptr : it->second.c_str();
}
inline bool ctx_flag(const komb_ctx *ctx, const char *name)
{
    const char *v = ctx_opt(ctx, name);
    return v && strcmp(v, "0") != 0;
}
namespace komb {
inline FinishMode finish_mode(const komb_ctx *ctx, FinishMode dflt)
{
    const char *e = ctx_opt(ctx, "FINISH");
    if (e && !strcmp(e, "local")) return FIN_LOCAL;
    if (e && !strcmp(e, "lds")) return FIN_LDS;
    if (e && !strcmp(e, "none")) return FIN_NONE;
    return dflt;
}
// The fixed point costs ~ sweeps x items, the peel ~ its items once + a latency per sub-round: a remainder with more items
// than this is not handed over (the peel goes on and offers a smaller one).  Option LOCAL_ITEMS overrides.
inline uint64_t local_item_limit(const komb_ctx *ctx, uint64_t dflt)
{
    if (const char *e = ctx_opt(ctx, "LOCAL_ITEMS")) return strtoull(e, nullptr, 10);
    return dflt;
}
inline uint32_t local_density_limit(const komb_ctx *ctx, uint32_t dflt)          // items per unit above which a remainder stays with the peel (0 = no rule)
{
    if (const char *e = ctx_opt(ctx, "LOCAL_DENSITY")) return (uint32_t)strtoul(e, nullptr, 10);
    return dflt;
}
inline uint32_t local_limit(const komb_ctx *ctx, uint64_t units, uint64_t divisor)
{
    uint64_t l = units / divisor;
    if (l < 4096) l = 4096;                  // small inputs go to the fixed point whole
    if (const char *e = ctx_opt(ctx, "LOCAL_LIMIT")) l = strtoull(e, nullptr, 10);
    if (l > units) l = units;
    return (uint32_t)l;
}
} // namespace komb

// Scratch buffers of one stage: everything still owned goes back to the context's pool on scope exit.
struct DevBufs {
    komb_ctx *ctx;
    std::vector<void *> owned;
    explicit DevBufs(komb_ctx *c) : ctx(c) {}
    DevBufs(const DevBufs &) = delete;
    DevBufs &operator=(const DevBufs &) = delete;
    template <class T> hipError_t alloc(T **out, size_t count)
    {
        void *q = nullptr;
        hipError_t e = ctx->pool.get(&q, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) { owned.push_back(q); *out = (T *)q; }
        return e;
    }
    void detach(void *q)                             // the caller keeps q (a pool block it will put() back itself)
    {
        for (auto &p : owned) if (p == q) p = nullptr;
    }
    void release(void *q)
    {
        for (auto &p : owned) if (p == q && q) { ctx->pool.put(q); p = nullptr; }
    }
    ~DevBufs() { for (void *p : owned) if (p) ctx->pool.put(p); }
};

// ---- error plumbing -------------------------------------------------------
#define KOMB_FAIL(ctx, code, ...)                                   \
    do {                                                            \
        char _b[512];                                               \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                      \
        (ctx)->err = _b;                                            \
        return (code);                                              \
    } while (0)

#define KOMB_HIP(ctx, call)                                                          \
    do {                                                                             \
        hipError_t _e = (call);                                                      \
        if (_e != hipSuccess) {                                                      \
            hipError_t _nomem = hipErrorOutOfMemory;                                 \
            KOMB_FAIL(ctx, _e == _nomem ? KOMB_ERR_NOMEM : KOMB_ERR_DEVICE,          \
                      "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e),         \
                      __FILE__, __LINE__);                                           \
        }                                                                            \
    } while (0)

#define KOMB_TRY(expr)                         \
    do {                                       \
        int _s = (expr);                       \
        if (_s != KOMB_OK) return _s;          \
    } while (0)

// small blocking device-to-host read, ordered on the context's stream.  It lands in pinned memory: a copy into pageable
// memory (a variable on the caller's stack) costs 50-100 us in the runtime alone, and a step makes a dozen of them while
// the GPU waits for the host's next decision.
constexpr size_t kStageBytes = 64 << 10;
inline hipError_t d2h(komb_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (ctx->h_stage && bytes <= kStageBytes && dst != ctx->h_stage) {
        hipError_t e = hipMemcpyAsync(ctx->h_stage, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) memcpy(dst, ctx->h_stage, bytes);
        return e;
    }
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    return e == hipSuccess ? hipStreamSynchronize(ctx->stream) : e;
}

// ---- primitives implemented in prims.hip (rocPRIM/hipCUB behind plain signatures)
namespace komb {
int prim_sort_u64(komb_ctx *ctx, uint64_t *keys, uint64_t *tmp_keys, int64_t n, int end_bit, uint64_t **sorted);
int prim_unique_u64(komb_ctx *ctx, const uint64_t *in, uint64_t *out, int64_t n, int64_t *n_out);
int prim_exclusive_sum_u32(komb_ctx *ctx, const uint32_t *in, uint32_t *out, int64_t n);   // out[n-1] valid; in/out may alias
int prim_exclusive_sum_u32_u64(komb_ctx *ctx, const uint32_t *in, unsigned long long *out, int64_t n);
int prim_sort_pairs_desc_i64(komb_ctx *ctx, int64_t *keys, int64_t *keys_tmp, uint32_t *vals, uint32_t *vals_tmp,
                             int64_t n, int end_bit, int64_t **sorted_keys, uint32_t **sorted_vals);

int prim_sort_pairs_u32_u64(komb_ctx *ctx, uint32_t *keys, uint32_t *keys_alt, unsigned long long *vals, unsigned long long *vals_alt,
                            int64_t n, int begin_bit, int end_bit, uint32_t **sorted_keys, unsigned long long **sorted_vals);
int prim_sort_pairs_u32_u32(komb_ctx *ctx, uint32_t *keys, uint32_t *keys_alt, uint32_t *vals, uint32_t *vals_alt,
                            int64_t n, int end_bit, uint32_t **sorted_keys, uint32_t **sorted_vals);
int prim_sort_pairs_u64_u32(komb_ctx *ctx, uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt,
                            int64_t n, int end_bit, uint64_t **sorted_keys, uint32_t **sorted_vals);

// ---- stages (each in its own translation unit)
int core_run(komb_ctx *ctx, int rank = 0, int world = 1, komb_allreduce_fn fn = nullptr, void *user = nullptr, bool sharded = false);
int truss_run(komb_ctx *ctx, const uint8_t *vmask_host, int rank, int world, komb_allreduce_fn fn, void *user);
int merge_run(komb_ctx *ctx, const double *susp_host, int32_t *order, int32_t *side, int64_t *n_block, double *max_density);
int corea_ranks(komb_ctx *ctx, const int32_t *deg, const int32_t *core, int64_t n, double *rank_deg, double *rank_key);
int graph_from_edges(komb_ctx *ctx, int64_t nv, int64_t n_raw, const int64_t *uv);
int graph_from_csr(komb_ctx *ctx, int64_t nv, const int64_t *rowptr, const int32_t *col);
void graph_free(komb_ctx *ctx);
void stager_free(komb_ctx *ctx);
// graph_build.hip: blocking copy pageable host memory <-> device through pinned staging buffers and a few host threads (>= 32 MB; else a plain copy)
hipError_t staged_copy(komb_ctx *ctx, void *dst, const void *src, size_t bytes, bool to_device);
void warm_up(komb_ctx *ctx);                 // graph_build.hip: first kernel launch of the library + the upload's staging buffers
// truss_prep.hip: the k-truss side of a symmetric CSR (device pointers; nv vertices, ns = 2 |E| slots)
int prep_build(komb_ctx *ctx, const uint32_t *rowptr, const int32_t *col, int64_t nv, int64_t ns, TrussPrep *out);
void prep_free(komb_ctx *ctx, TrussPrep *p);
int prep_ensure(komb_ctx *ctx);              // the resident graph's preparation, built if absent (ctx->prep)
int prep_sources(komb_ctx *ctx, TrussPrep *p);   // p->osrc, made on first use
// sum d^2, sum min(d,d), max d, (unused), sum d+ + d+ of the resident graph: the roofline model's inputs (measurement only)
int graph_moments(komb_ctx *ctx, int64_t out[5]);
// the subgraph induced by a vertex mask as a symmetric CSR of its own (new ids = ranks among the kept vertices)
struct InducedCsr { int64_t nv = 0, ns = 0; uint32_t *rowptr = nullptr; int32_t *col = nullptr; int32_t *vold = nullptr; };
int induce_csr(komb_ctx *ctx, const uint8_t *vmask_host, InducedCsr *out);
void induced_free(komb_ctx *ctx, InducedCsr *g);
// the canonical edge list (eu[k], ev[k]) of a symmetric CSR, through vold (new -> original ids) when it is an induced subgraph's
int edge_list(komb_ctx *ctx, const uint32_t *rowptr, const int32_t *col, int64_t nv, const int32_t *vold, int32_t *eu, int32_t *ev);
int truss_edges_canonical(komb_ctx *ctx);      // ktruss.hip: d_t_eu / d_t_ev of the last whole-graph run (komb_truss_fetch: on the first request per graph)
void truss_free(komb_ctx *ctx);
int truss_support_canonical(komb_ctx *ctx);    // ktruss.hip: d_t_sup from d_t_slice (whole-graph runs: on the first komb_truss_fetch_support)
void peel_ctrl_pre(hipStream_t s, uint32_t *d_grp_done);
void peel_collect_ctrl(hipStream_t s, PeelCtrl *d_collect, const PeelCtrl *d_from);
void peel_ctrl_init(hipStream_t s, PeelCtrl *d_ctrl, uint32_t *d_grp_done, uint32_t units, uint32_t tail_limit = 0);
int peel_grid(int64_t units);

// Issue `launch()` in batches until the device control block reports done.
// The host never decides what a launch does: every launch reads the control
// block its predecessor finalised (SCAN or PROCESS, or nothing once done), so
// there is no host round trip per sub-round; the host only polls a copy of the
// control block one batch behind the launches it keeps queued.
template <typename F>
int drive_peel(komb_ctx *ctx, PeelCtrl *d_ctrl, int64_t units, F &&launch, int *launches_out)
{
    // launch(i) issues the launch with index i; the state says which index it expects (PeelCtrl::seq)
    // Launches per batch.  What is queued behind the launch that finishes the peel is wasted (a launch that finds nothing to do
    // is ~5 us, and the host sees `done` one batch late): on average 1.5 batches.  The first batch is long (the giant steps of
    // the first levels keep the GPU busy while the host queues); the next few are short, because the graphs this path is
    // measured on hand over to a finish after 35-50 launches (C3: 37 of 72 launches did something with batches of 24; C2
    // k-core: 41 of 72); a peel that is still running after those has thousands of sub-rounds to go and gets long batches.
#ifndef KOMB_PEEL_BATCH
#define KOMB_PEEL_BATCH 24
#endif
#ifndef KOMB_PEEL_BATCH_SHORT
#define KOMB_PEEL_BATCH_SHORT 6
#endif
    constexpr int kBatch = KOMB_PEEL_BATCH, kBatchShort = KOMB_PEEL_BATCH_SHORT, kShortBatches = 6;
    PeelCtrl first;
    KOMB_HIP(ctx, d2h(ctx, &first, d_ctrl, sizeof(PeelCtrl)));
    int32_t next = first.seq;
#ifdef KOMB_DEBUG_SWITCHES
    if (const char *tr = getenv("KOMB_PEEL_TRACE")) {
        // debug: one launch at a time; append "mode level round light heavy live_mode live_count remaining us" per step
        FILE *f = fopen(tr, "a");
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
            if (f) fclose(f);
            if (a) (void)hipEventDestroy(a);
            KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "peel trace: hipEventCreate failed");
        }
        int launches = 0;
        if (f) fprintf(f, "# peel units=%lld\n", (long long)units);
        for (int64_t i = 0; i < 4 * units + 4096; ++i) {
            PeelCtrl before;
            if (d2h(ctx, &before, d_ctrl, sizeof(PeelCtrl)) != hipSuccess) break;
            if (before.done) break;
            (void)hipEventRecord(a, ctx->stream);
            launch(before.seq); ++launches;
            (void)hipEventRecord(b, ctx->stream);
            (void)hipEventSynchronize(b);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, a, b);
            if (f) fprintf(f, "%d %d %d %u %u %d %u %u %.1f\n", before.mode, before.level, before.round, before.cur_light,
                           before.cur_heavy, before.live_mode, before.live_count, before.remaining, ms * 1000.f);
        }
        if (f) fclose(f);
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        if (launches_out) *launches_out = launches;
        KOMB_HIP(ctx, d2h(ctx, &ctx->h_ctrl[0], d_ctrl, sizeof(PeelCtrl)));
        return KOMB_OK;
    }
#endif
    EventSet evs;
    hipEvent_t ev[2] = {nullptr, nullptr};
    KOMB_HIP(ctx, evs.make(&ev[0], hipEventDisableTiming));
    KOMB_HIP(ctx, evs.make(&ev[1], hipEventDisableTiming));
    const int64_t max_batches = (4 * units + 4096) / kBatch + 16 + kShortBatches;   // > 2 launches per unit: cannot be reached
    int launches = 0, slot = 0, status = KOMB_OK;
    bool have_prev = false, finished = false, stuck = false;
    int32_t seen_seq = first.seq - 1;
    for (int64_t batch = 0; batch < max_batches && !finished; ++batch) {
        const int nb = (batch >= 1 && batch <= kShortBatches) ? kBatchShort : kBatch;
        for (int i = 0; i < nb; ++i) { launch(next); ++next; ++launches; }
        if (hipMemcpyAsync(&ctx->h_ctrl[slot], d_ctrl, sizeof(PeelCtrl), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipEventRecord(ev[slot], ctx->stream) != hipSuccess) { status = KOMB_ERR_DEVICE; break; }
        if (have_prev) {
            if (hipEventSynchronize(ev[slot ^ 1]) != hipSuccess) { status = KOMB_ERR_DEVICE; break; }
            const PeelCtrl &h = ctx->h_ctrl[slot ^ 1];
            if (h.done) finished = true;
            else if (h.seq == seen_seq) { stuck = true; finished = true; }   // a whole batch of launches moved nothing
            seen_seq = h.seq;
        }
        have_prev = true;
        slot ^= 1;
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (launches_out) *launches_out = launches;
    if (status != KOMB_OK || e != hipSuccess)
        KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "peel driver: HIP failure (%s)", hipGetErrorString(e));
    if (d2h(ctx, &ctx->h_ctrl[0], d_ctrl, sizeof(PeelCtrl)) != hipSuccess)
        KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "peel driver: control block readback failed");
    if (!ctx->h_ctrl[0].done)
        KOMB_FAIL(ctx, KOMB_ERR_DEVICE, "peel driver: %s (level %d, remaining %u, launch %d of state %d)",
                  stuck ? "the launches make no progress" : "launch budget exhausted before completion",
                  ctx->h_ctrl[0].level, ctx->h_ctrl[0].remaining, (int)next, (int)ctx->h_ctrl[0].seq);
    return KOMB_OK;
}
} // namespace komb
