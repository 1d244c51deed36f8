/*
 * komb_accel.h -- C ABI of the MI355X-native k-core / k-truss / CoreA path.
 *
 * This is the drop-in boundary for KOMB's decomposition hot path: each entry
 * point replaces the igraph / CoreA call group named beside it (paths are
 * relative to the KOMB reference tree).  Plain pointers and sizes only; the
 * caller allocates every output; the library never frees caller memory; all
 * device state lives behind the opaque komb_ctx.  Every function returns
 * KOMB_OK (0) or a negative komb_status; komb_last_error() gives the text.
 * There is no CPU fallback: without a usable gfx950 device every compute
 * entry point fails with KOMB_ERR_DEVICE.
 *
 * Edge identity across the boundary is the canonical pair (min(u,v),max(u,v))
 * in lexicographic order -- igraph's internal edge ids are not observable in
 * any KOMB output (src/graph.cpp:519-534 only maps edges back to vertices).
 *
 * Threading: call from one host thread per context (the reference reaches this
 * seam from its main thread, src/graph.cpp:449 and src/komb2.cpp:132).
 */
#ifndef KOMB_ACCEL_H
#define KOMB_ACCEL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KOMB_ACCEL_ABI_VERSION 7

typedef enum komb_status {
    KOMB_OK          =  0,
    KOMB_ERR_ARG     = -1,   /* bad argument / graph not loaded / out-of-range id   */
    KOMB_ERR_DEVICE  = -2,   /* no HIP device, or a HIP call failed                 */
    KOMB_ERR_NOMEM   = -3,   /* host or device allocation failed                    */
    KOMB_ERR_LIMIT   = -4,   /* graph exceeds the 32-bit slot/edge-id design limits */
    KOMB_ERR_STATE   = -5    /* call order violated (e.g. fetch before run)         */
} komb_status;

typedef struct komb_ctx komb_ctx;

typedef struct komb_opts {
    int32_t device;        /* HIP device ordinal (LOCAL_RANK for one process per GPU) */
    int32_t verbosity;     /* 0 silent, 1 progress on stderr, 2 + phase times of create / graph build */
    int32_t reserved[2];   /* [0]: KOMB_CREATE_* flags below; [1]: 0                  */
} komb_opts;
/* komb_opts.reserved[0] (ABI version 7; until then the library read KOMB_NULL_STREAM / KOMB_NO_WARMUP from the environment) */
#define KOMB_CREATE_NULL_STREAM 1   /* work on the legacy default stream instead of a stream of the context's own      */
#define KOMB_CREATE_NO_WARMUP   2   /* do not load the code object (a first kernel launch) inside komb_create           */
#define KOMB_CREATE_WARM_UPLOAD 4   /* make the pinned staging buffers of a >= 64 MB graph upload inside komb_create   */

typedef struct komb_stats {
    int64_t nv, ne;                 /* simple graph: vertices, undirected edges      */
    int64_t triangles;              /* T (sum of supports / 3), after komb_truss_run */
    int64_t sum_deg_sq;             /* sum_v d(v)^2 = sum_{(u,v) in E} d(u)+d(v)      */
    int64_t wedge_items;            /* sum_{(u,v) in E} min(d(u),d(v))                */
    int64_t oriented_items;         /* sum_{(a->b)} d+(a)+d+(b), (degree,id) orientation */
    int32_t max_degree, max_coreness, max_trussness;
    int32_t core_levels, core_subrounds;    /* populated levels; PROCESS sub-rounds   */
    int32_t core_launches, truss_tail_runs; /* launches issued; hand-overs to the LDS tail kernel */
    int32_t truss_levels, truss_subrounds;  /* populated levels; PROCESS sub-rounds   */
    int32_t truss_scans, truss_launches;    /* SCAN launches; launches issued         */
    /* HIP-event times (ms), each measured on the stream the kernels run on */
    double  ms_build;               /* a1: edge list -> simple CSR                    */
    double  ms_core;                /* a2+a3: degree + k-core peel launches           */
    double  ms_orient;              /* truss: induced-subgraph filter + its per-vertex lines (0 for the whole graph: the oriented CSR comes with the graph) */
    double  ms_tri_count;           /* truss: triangle enumeration, support counting  */
    double  ms_tri_fill;            /* truss: triangle enumeration, incidence fill    */
    double  ms_compact;             /* truss: blocks + sorted records (or bounded slices) -> dense index */
    double  ms_support;             /* = ms_tri_count + ms_tri_fill + ms_sort + ms_compact */
    double  ms_allreduce;           /* truss: support all-reduce callback (sharded)   */
    double  ms_peel;                /* truss: all peel launches (SCAN + PROCESS)      */
    double  ms_gather;              /* truss: canonical-order result gather           */
    double  ms_corea;               /* a9/a10: CoreA rank kernels                     */
    double  ms_tail;                /* truss: setup + LDS tail kernel, part of ms_peel */
    /* local finish (h-index fixed point on the remainder the peel hands over; ABI version 2) */
    int64_t core_local_items, truss_local_items;   /* entries of the compact index            */
    int32_t core_local_units, core_local_sweeps;   /* vertices handed over; sweeps to the fixed point */
    int32_t truss_local_units, truss_local_sweeps; /* edges handed over; sweeps                */
    double  ms_core_local;          /* part of ms_core: numbering + collect + sweeps + scatter */
    double  ms_truss_local;         /* part of ms_peel                                         */
    /* index build by record stream (ABI version 3) */
    double  ms_sort;                /* truss: sort of the incidence records by destination edge */
    int64_t tri_records;            /* truss: record positions of the stream (incl. unused claim tails) */
    int32_t index_layout;           /* truss: 0 = record stream, 1 = bounded slices, 2 = exact two-pass */
    /* sharded peel (ABI version 4): the last komb_core_run_sharded / komb_truss_run_sharded with komb_set_shard_peel */
    int32_t shard_exchanges;        /* all-reduce callbacks made by the peel (two per sub-round + level changes) */
    double  ms_exchange;            /* host time inside them (stream drain + callback); part of ms_core / ms_peel */
    int64_t exchange_words;         /* 32-bit words they carried                                                 */
    /* graph build (ABI version 5): host wall-clock parts of ms_build (which is now wall-clock too: upload + device work) */
    double  ms_build_h2d;           /* staged host -> device copy of the raw pairs / the CSR                      */
    double  ms_build_relabel;       /* 0 since ABI version 7: the k-truss side is no longer made with the graph (ms_prepare) */
    /* k-truss preparation (ABI version 7): (degree,id) renumbering, oriented CSR, canonical edge map, the enumeration's lines
     * and tasks -- made by the first komb_truss_* call of a graph (or komb_truss_prepare), inside that call */
    double  ms_prepare;             /* device time of the preparation the last k-truss call (or komb_truss_prepare) made; 0 when it found one */
    int32_t truss_prepared;         /* 1 when the last k-truss call made a preparation (whole graph or induced subgraph)      */
    int32_t engine_flags;           /* which engines the last k-core / k-truss call ran: KOMB_ENGINE_* below                  */
    double  ms_prep_vertex;         /* parts of ms_prepare: vertices ordered by (degree, id) (k_prep_vertex, radix sort, scans)   */
    double  ms_prep_edges;          /* every canonical edge handed to its oriented row (k_prep_kept, _heavy, k_prep_dplus, scan)  */
    double  ms_prep_rows;           /* rows sorted, lines and canonical map written (k_prep_rows, _heavy); the rest: task table  */
    int32_t stream_retries;         /* record-stream build: 1 when the first attempt's capacities (two triangles per edge) ran out and the enumeration ran again with what it asked for */
    int32_t reserved0;
} komb_stats;
#define KOMB_ENGINE_LOCAL_FINISH 1   /* a remainder went to the local fixed point (local_dev.h)            */
#define KOMB_ENGINE_LDS_TAIL     2   /* ... to the single-workgroup LDS tail                                */
#define KOMB_ENGINE_SHARD_PEEL   4   /* the peel ran sharded by unit range (shard_dev.h)                    */
#define KOMB_ENGINE_TWO_PASS     8   /* the incidence index was built by the exact two-pass fallback        */

/* ---- lifetime ---------------------------------------------------------- */
komb_ctx   *komb_create(const komb_opts *opts);      /* NULL only on host OOM      */
void        komb_destroy(komb_ctx *ctx);
const char *komb_last_error(const komb_ctx *ctx);    /* "" when no error           */
int         komb_abi_version(void);
/* Tuning / test switches of one context (ABI version 7; until then KOMB_* environment variables, which the library no
 * longer reads: ambient environment cannot change which engine a drop-in runs).  None changes a result.  name: FINISH
 * (local | lds | none), LOCAL_LIMIT, LOCAL_ITEMS, LOCAL_DENSITY, LOCAL_DEFER_CHUNKS, TAIL, CORE_TAIL, INDEX (stream |
 * two_pass), REC_CAP, OWN_DENSE_CAP, NO_OWN_DENSE, NO_REC_SCRATCH, NO_FIRST_QUEUE, RETIRE_EVERY, SHARD_ENGINE, and the
 * stderr traces TRI_DEBUG, POOL_DEBUG, BUILD_DEBUG, LOCAL_DEBUG, TAIL_DEBUG.  value NULL unsets. */
int         komb_set_option(komb_ctx *ctx, const char *name, const char *value);

/* ---- graph construction ------------------------------------------------ */
/* Replaces igraph_create + igraph_simplify(multiple=true, loops=true)
 * (src/graph.cpp:418, src/graph.cpp:438): n_raw (u,v) pairs exactly as
 * generateGraph leaves them in `edges` (src/graph.cpp:379-389), vertex ids in
 * [0,nv).  Removes loops and parallel edges on the device and keeps the graph
 * resident in HBM as a symmetric CSR (rows ascending, the caller's ids).  What
 * only the k-truss path needs is made by the first k-truss call (below).
 * Every result is reported in the caller's vertex ids. */
int komb_graph_from_edges(komb_ctx *ctx, int64_t nv, int64_t n_raw,
                          const int64_t *uv_pairs);

/* Same, from an already simple, symmetric, row-sorted CSR (host pointers). */
int komb_graph_from_csr(komb_ctx *ctx, int64_t nv, const int64_t *rowptr,
                        const int32_t *col);

/* igraph_vcount / igraph_ecount (src/graph.cpp:443-444). */
int komb_graph_info(komb_ctx *ctx, int64_t *nv, int64_t *ne);

/* Copy the resident CSR back: rowptr[nv+1], col[2*ne]. */
int komb_graph_get_csr(komb_ctx *ctx, int64_t *rowptr, int32_t *col);

/* ---- k-core ------------------------------------------------------------ */
/* Replaces igraph_degree(ALL,NO_LOOPS) + igraph_coreness(ALL)
 * (src/graph.cpp:462-463).  komb_core_run computes on the device and leaves
 * degree/coreness in HBM (this is the timed region of bench.py);
 * komb_core_fetch copies them out; komb_degree_coreness = run + fetch. */
int komb_core_run(komb_ctx *ctx);
/* One process per GPU, every rank holding the same graph: rank r owns the vertices
 * [nv*r/world, nv*(r+1)/world) -- their live degrees and the decrements on them; every sub-round the ranks exchange
 * their parts of the frontier through `allreduce` (komb_allreduce_fn below: each rank fills its own segment of a zeroed
 * buffer, so the SUM is the concatenation), stamp the whole frontier and walk all of its rows, each applying the
 * decrements it owns.  All ranks end with identical, complete results, bit-equal to komb_core_run's.  world == 1 is
 * the same engine without a collective. */
typedef int (*komb_allreduce_fn)(void *user, void *device_u32, int64_t count);
int komb_core_run_sharded(komb_ctx *ctx, int32_t rank, int32_t world, komb_allreduce_fn allreduce, void *user);
int komb_core_fetch(komb_ctx *ctx, int32_t *degree /*[nv]*/, int32_t *coreness /*[nv]*/);
int komb_degree_coreness(komb_ctx *ctx, int32_t *degree, int32_t *coreness);

/* ---- k-truss ----------------------------------------------------------- */
/* Replaces igraph_induced_subgraph_map + igraph_trussness
 * (src/graph.cpp:502, src/graph.cpp:508).  vmask (host, nv bytes, nullable)
 * selects the induced subgraph exactly like the max-core vertex list built at
 * src/graph.cpp:470-473; NULL = whole graph.  Edges are reported with ORIGINAL
 * vertex ids (what invmap gives at src/graph.cpp:531-532), canonical order.
 * komb_truss_run computes on the device (timed region) and leaves the trussness
 * vector there in canonical edge order -- igraph_trussness's output, indexed by
 * edge id.  komb_truss_fetch copies (eu,ev,truss)[ne_sub] out; any of the three
 * may be NULL.  The ENDPOINTS of the canonical edges are what igraph_edge answers
 * afterwards (src/graph.cpp:529-532), not part of igraph_trussness: for a
 * whole-graph run they are made by the first fetch that asks for them and kept
 * with the graph (a run under a vmask makes its subgraph's before it returns).
 * Trussness of a triangle-free edge is 2. */
int komb_truss_run(komb_ctx *ctx, const uint8_t *vmask);
/* What igraph_trussness does to its argument before it lists a triangle (src/graph.cpp:508: vertices ordered by degree,
 * every edge oriented from its lower to its higher endpoint) is the k-truss PREPARATION here: (degree,id)-ranked internal
 * ids, the oriented CSR in them, the map back to the canonical edge order, the enumeration's per-vertex lines and task
 * table.  The first k-truss call of a graph makes it, inside the call and its time
 * (komb_stats.ms_prepare), and it stays with the graph: later calls find it.  komb_truss_prepare makes it ahead of time
 * (a no-op when it exists); komb_truss_unprepare drops it together with the last k-truss result (bench.py times
 * unprepare + komb_truss_run: nothing of a previous call is reused).  A vmask run prepares its induced subgraph every time. */
int komb_truss_prepare(komb_ctx *ctx);
int komb_truss_unprepare(komb_ctx *ctx);
/* One process per GPU, every rank holding the same graph: the triangle-support
 * phase is sharded by source-vertex range [rank/world) and the partial support
 * vectors (|E|+1 words) are summed across ranks by `allreduce` -- an in-place SUM
 * all-reduce over uint32[count] in device memory (the host implements it with RCCL; the
 * library has drained its stream when it calls, the reduction must be complete when the
 * callback returns; 32-bit two's-complement sums, so an int32 view of the words is fine;
 * return 0 on success).  Incidence index, peel and gather then run on every rank; all
 * ranks end with identical results.  world == 1 is komb_truss_run.
 * komb_set_shard_peel(ctx, 1) makes the following sharded runs split the PEEL as well: rank r owns the internal edge ids
 * [m*r/world, m*(r+1)/world) -- their live supports and the decrements on them -- and the ranks exchange their parts of
 * the frontier every sub-round through the same callback (as komb_core_run_sharded does for vertices).  Same results;
 * slower than the replicated peel on one node (DESIGN.md section 6 has the measurements), hence opt-in. */
int komb_truss_run_sharded(komb_ctx *ctx, const uint8_t *vmask, int32_t rank, int32_t world,
                           komb_allreduce_fn allreduce, void *user);
int komb_set_shard_peel(komb_ctx *ctx, int32_t on);
/* One process per GPU, every rank holding the same graph, NO exchange: every rank runs the whole k-truss path (support,
 * index and peel of one graph do not shard across GPUs at a profit: DESIGN.md section 6) and materialises the results of
 * ITS slice of the canonical edges only -- truss[k], support[k] for k in [ne*rank/world, ne*(rank+1)/world), zeros
 * elsewhere (a SUM all-reduce of the ranks' arrays is the whole result; eu / ev are complete on every rank).  What
 * bench.py --gpus N runs by default.  With a vmask the results are complete on every rank.  world == 1 is komb_truss_run. */
int komb_truss_run_slice(komb_ctx *ctx, const uint8_t *vmask, int32_t rank, int32_t world);
int komb_truss_count(komb_ctx *ctx, int64_t *ne_sub);
int komb_truss_fetch(komb_ctx *ctx, int32_t *eu, int32_t *ev, int32_t *truss);
/* per-edge triangle counts the peel started from (canonical order): an extra of this library (igraph_trussness has no
 * such output), put into canonical order by the first call after a run */
int komb_truss_fetch_support(komb_ctx *ctx, int32_t *support);
int komb_trussness(komb_ctx *ctx, const uint8_t *vmask, int64_t *ne_out,
                   int32_t *eu, int32_t *ev, int32_t *truss);

/* ---- CoreA ------------------------------------------------------------- */
/* Replaces CoreA::getAnomalyScore + CoreA::fractionalRank x2
 * (src/CoreA.h:109-140, 142-187): key = coreness*n + degree in 64-bit
 * (src/CoreA.h:122 is `int`, undefined once it overflows); fractional ranks by
 * device sort + run bounds; |ln r_deg - ln r_key| with the host libm so the
 * "%f" text of CombineCoreA::run (src/CombineCoreA.h:36-39) is unchanged.
 * degree/coreness are host arrays (what readKOMBOutput returns,
 * src/CoreA.h:24-56). */
int komb_corea_scores(komb_ctx *ctx, const int32_t *degree, const int32_t *coreness,
                      int64_t nv, double *score);
/* the two rank vectors themselves (exact half-integers), for tests */
int komb_corea_ranks(komb_ctx *ctx, const int32_t *degree, const int32_t *coreness,
                     int64_t nv, double *rank_degree, double *rank_key);

/* Replaces CombineCoreA::runMerge over its two HashIndexedMinHeap instances (src/CombineCoreA.h:45-219,
 * src/HashIndexedMinHeap.h:10-238; dead code in the reference -- nothing calls it): the greedy densest-block
 * peel over a row copy and a column copy of the resident graph.  suspiciousness: host array [nv] (what
 * getAnomalyScore returns) or NULL for plain degrees.  order[2*nv] / side[2*nv] are filled from the back, as the
 * reference fills `order` / `modes`; the first *n_block entries are the densest block (rows: side 0, columns:
 * side 1), *max_density its density.  Ties between equal priorities are resolved exactly as the reference's heap
 * resolves them (the same sift operations in the same order), which makes the peel sequential: the device runs
 * it on one lane (heaps in LDS up to 4096 nodes; graphs above 2^17 nodes are refused with KOMB_ERR_LIMIT).  Two defects of the reference are not reproduced: removed[][]
 * read uninitialised (:104-108) and `cols` sized by the number of rows (:191). */
int komb_densest_block(komb_ctx *ctx, const double *suspiciousness, int32_t *order, int32_t *side,
                       int64_t *n_block, double *max_density);

/* ---- instrumentation --------------------------------------------------- */
int komb_get_stats(komb_ctx *ctx, komb_stats *out);
/* Measurement only: fills komb_stats.sum_deg_sq / wedge_items / max_degree / oriented_items of the resident graph (the
 * inputs of the roofline model's algorithmic bytes; makes the k-truss preparation if absent).  No product path calls it
 * (until ABI version 7 every graph build ran this 2.2 ms kernel). */
int komb_graph_moments(komb_ctx *ctx);

/* ---- synthetic workload (host code, no device) ------------------------- */
/* Power-law "hybrid unitig graph" generator of SURVEY.md section 8(d): a union
 * of n_cliques small cliques (size min(1+Geom(0.45),6)) whose members are drawn
 * from w_i ~ (i+1)^(-1/(alpha-1)) and scattered by a seeded bijection; mirrors
 * how generateGraph expands read cliques (src/graph.cpp:310-352).  Two-call
 * pattern: uv_pairs==NULL returns the number of raw pairs; otherwise fills
 * uv_pairs[2*n_raw] (must be the size the first call returned). */
int64_t komb_gen_hug_edges(int64_t nv, int64_t n_cliques, double alpha,
                           uint64_t seed, int64_t *uv_pairs);

#ifdef __cplusplus
}
#endif
#endif /* KOMB_ACCEL_H */
