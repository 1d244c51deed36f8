#!/usr/bin/env python3
"""bench.py -- peeled edges/sec of the k-truss hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the k-truss path over the synthetic power-law unitig graph, from the resident simple graph
(the symmetric CSR igraph_simplify would leave) to per-edge trussness in canonical order -- what igraph_trussness does
(reference src/graph.cpp:508; SURVEY 8(d): "support computation + full peel, H2D of the CSR excluded"):
    preparation   (degree,id) renumbering + orientation + canonical edge map + the enumeration's lines and tasks
                  (truss_prep.hip; komb_stats.ms_prepare),
    support       triangle enumeration + incidence index,
    peel          level-synchronous sub-rounds + local fixed point,
    gather        results in canonical edge order.
Every timed step starts with komb_truss_unprepare: NOTHING a previous step computed is reused (rounds 1-4 built the
preparation with the graph, outside the timed region; `value_resident` is that older figure: the same step on a graph whose
preparation is kept, as a second KOMB call on the same graph would find it).  Rank 0 prints ONE JSON line.

N=1 workload = BASELINE.json configs[2] (|V|=10M, |E|~100M, full k-truss, the configuration the metric is quoted on).
--config c2 selects configs[1] (|V|=1M, |E|~10M); the k-core time of the same graph is reported alongside.

N>1 (plain `--gpus N`) = BASELINE.json configs[3] / north_star's partition: the SAME graph on every rank ("scaling":
"strong", `value` = |E| / the slowest rank's time); the triangle support is counted in shards (each rank its range of the
task table) and summed with one all-reduce of the |E|+1 support words (RCCL over xGMI), and the peel is sharded by edge
range: rank r owns the live supports of its internal edge ids and the decrements on them, the ranks' parts of the frontier
are exchanged every sub-round (komb_set_shard_peel).  Opt-in alternatives, each also timed after the region and printed
beside the headline under `alternatives`: --c4-allreduce (the support all-reduce only, the peel replicated: rounds 1-3's
default), --replicas-sliced (round 4's default: nothing exchanged, every rank runs the whole path and materialises its slice
of the canonical results), --replicas (nothing sliced), --batch (one graph per rank, seed + rank: weak scaling, its own
metric name).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s

CONFIGS = {
    # name: (nv, cliques, alpha, seed, description)
    "c3": (10_000_000, 24_250_000, 2.6, 42, "C3: synthetic power-law unitig graph |V|=10M |E|~100M, full k-truss peel"),
    "c2": (1_000_000, 2_450_000, 2.6, 42, "C2: synthetic power-law unitig graph |V|=1M |E|~10M, full k-truss peel"),
    "tiny": (100_000, 245_000, 2.6, 42, "tiny: |V|=100k |E|~1M (debug)"),
}
CPU_SAMPLE = "c2"              # the single-thread CPU figure is timed on the C2 graph (~25 s); the all-cores one on the workload itself
PHASES = ("ms_prepare", "ms_prep_vertex", "ms_prep_edges", "ms_prep_rows", "ms_orient", "ms_tri_count", "ms_allreduce", "ms_tri_fill", "ms_sort", "ms_compact", "ms_peel", "ms_tail",
          "ms_truss_local", "ms_gather", "ms_exchange")


def algorithmic_bytes(st):
    """Bytes each phase must move, per step (DESIGN.md 'Algorithmic bytes').
    V = vertices, E = edges, T = triangles, O = sum over oriented edges of d+(a)+d+(b), R = record positions of the stream."""
    V, E, T, O, R = st["nv"], st["ne"], st["triangles"], st["oriented_items"], st["tri_records"]
    # preparation: the symmetric CSR in once (rowptr + both directions of every edge); out once: oriented targets,
    # the canonical map, one 64-byte line and two id maps per vertex
    prep_vertex = 4 * V + 3 * 2 * 8 * V + 8 * V         # row pointers in; three radix passes over (degree, id) pairs, read + write; two id maps out
    prep_edges = (4 * V + 4 * E) + 8 * E                # the upper half of the CSR in; one (id, canonical id) pair per edge out
    prep_rows = 8 * E + 4 * E + 4 * E + 64 * V          # the pairs in; oriented targets, every slot's canonical id and one 64-byte line per vertex out
    prepare = prep_vertex + prep_edges + prep_rows
    tri_count = 12 * E + 4 * O + 24 * T                 # SURVEY 8(d) B_sup
    peel = 8 * E + 24 * T + 24 * T + 16 * T             # truss+stamp per edge; slice entries; two stamps per entry; 2 RMW per triangle
    survey_peel = 8 * E + 4 * st["sum_deg_sq"] + 16 * T # SURVEY 8(d) B_peel (merge re-intersection; not what we do)
    sort = 2 * 2 * 12 * R                               # the passes that run: two radix passes over 12-byte records, read + write
    finish = 16 * E + 48 * T + 4 * R                    # supports and slice pairs; every entry in once (record or block entry), out once
    gather = 12 * E                                     # one pass: stamp and canonical id in; the trussness (at its canonical id) out
    return {"prepare": prepare, "prep_vertex": prep_vertex, "prep_edges": prep_edges, "prep_rows": prep_rows, "tri_count": tri_count, "peel": peel, "survey_peel": survey_peel,
            "sort": sort, "finish": finish, "gather": gather}


def measured_traffic(config, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE, separate runs of this same command; scripts/prof_summary.py), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{config}_*traffic.json")))
    if not files:
        return None, None, None
    data = json.load(open(files[-1]))
    rec = data.get(kernel)
    if not rec:
        return None, None, os.path.basename(files[-1])
    return rec["hbm_bytes_per_launch"], rec.get("hbm_bytes_per_launch_max"), os.path.basename(files[-1])


def cpu_baseline(workload_csr, workload_ne):
    """The CPU restatement of igraph's trussness (oracle/, test infrastructure) timed on this box's host cores, next to
    the GPU figure: single thread pinned to one core (igraph is single-threaded and the reference calls it from one
    thread, src/graph.cpp:508) on a bounded sample of the same generator (the C2 graph, |E| ~ 10M), and the OpenMP
    all-cores variant on the WORKLOAD ITSELF (the full graph the GPU figure is quoted on).  The library is compiled on this
    machine for its own instruction set (oracle/Makefile target native)."""
    import komb_amd
    from oracle import oracle as O
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    native = O.native_lib() is not None
    run1 = (lambda rp, c: O.trussness_native(rp, c, 1)) if native else O.trussness
    nv, ncl, alpha, seed = CONFIGS[CPU_SAMPLE][:4]
    uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
    rowptr, col = O.simplify(nv, uv)
    del uv
    ne = len(col) // 2
    cpus = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, {cpus[0]})                    # taskset: one core for the single-thread figure
    try:
        t0 = time.perf_counter()
        run1(rowptr, col)
        dt = time.perf_counter() - t0
    finally:
        os.sched_setaffinity(0, set(cpus))
    out = {"value": ne / dt, "unit": "edges/s", "cores": 1, "kind": "port",
           "sample": f"oracle orc_trussness (support + bucket peel, 1 thread pinned to cpu {cpus[0]}, "
                     f"{'-march=native' if native else 'portable build'}) on the C2 graph |V|={nv} |E|={ne} of the same generator: {dt:.1f} s",
           "cpu_model": model, "host_cpus": len(cpus),
           "full_graph_recorded": "the full C3 graph (|E|=100.1M) took 335 s = 3.0e5 edges/s on one thread of a GPU box "
                                  "(profiles/r01_c3_full_parity.log)"}
    if native and workload_csr is not None:
        # the GPU boxes show every CPU of the host in the affinity mask but grant a share of 16 per GPU: more threads
        # than that only thrash (256 threads: 116 s against 15 s on one)
        nthr = min(len(cpus), 16)
        rp_w, col_w = workload_csr
        t0 = time.perf_counter()
        O.trussness_native(rp_w, col_w, nthr)
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": workload_ne / dt2, "unit": "edges/s", "cores": nthr, "kind": "port",
                            "sample": f"oracle orc_trussness_omp (parallel supports + level-synchronous parallel peel, OpenMP, "
                                      f"{nthr} threads) on the WHOLE workload graph |E|={workload_ne}: {dt2:.1f} s"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", action="store_true",
                    help="N > 1, opt-in: every rank decomposes a graph of its own (seed + rank) -- KOMB's one-graph-per-sample "
                         "shape (KOMB.py --file-list); nothing is exchanged, weak scaling, reported under its own metric name")
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1, opt-in: the same graph on every rank, every rank runs the whole single-GPU path, no exchange")
    ap.add_argument("--replicas-sliced", action="store_true",
                    help="N > 1, opt-in (round 4's default): the same graph on every rank, every rank runs the whole path and materialises "
                         "only its slice of the canonical results (komb_truss_run_slice); no exchange")
    ap.add_argument("--same-graph", action="store_true", help="(the default for N > 1; accepted for older command lines)")
    ap.add_argument("--c4-allreduce", "--shard", dest="shard", action="store_true",
                    help="N > 1, opt-in: the support-counting enumeration split over the ranks + one all-reduce of the |E|+1 support words, "
                         "then index build, peel and gather replicated (rounds 1-3's default)")
    ap.add_argument("--shard-peel", action="store_true",
                    help="(the default for N > 1 since round 5; accepted for older command lines) the support all-reduce AND the peel "
                         "sharded by edge range, the ranks' parts of the frontier exchanged every sub-round")
    ap.add_argument("--no-alternatives", action="store_true", help="N > 1: do not time the other flows after the region")
    ap.add_argument("--no-build", action="store_true",
                    help="load the prebuilt libkomb_accel.so, spawn no compiler (use under rocprofv3)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed workload (+ the k-core of the same graph): no runTruss-faithful variant, no first-call "
                         "figures, no C2 block, no CoreA -- for rocprofv3 runs, whose per-kernel averages should see the timed launches only")
    ap.add_argument("--faithful", action="store_true", help="(accepted for older command lines)")
    args = ap.parse_args()
    picked = [n for n in ("batch", "replicas", "replicas_sliced", "shard") if getattr(args, n)]
    if args.batch and (args.replicas or args.same_graph or args.shard or args.replicas_sliced or args.shard_peel):
        raise SystemExit("--batch runs one graph per rank: it cannot be combined with --replicas / --replicas-sliced / --same-graph / --shard")
    if args.replicas and (args.shard or args.shard_peel):
        raise SystemExit("--replicas runs the unsharded path on every rank: it cannot be combined with --shard")
    if len(picked) > 1:
        raise SystemExit(f"pick one flow for N > 1, not {picked}")
    if args.shard_peel and args.shard:
        args.shard = False                                  # (--shard-peel implies the all-reduce; it is the default flow)
    mode = "single"
    if args.gpus > 1:
        mode = "batch" if args.batch else "replicas" if args.replicas else "sliced" if args.replicas_sliced else "c4" if args.shard else "sharded"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: start one rank per GPU ourselves -- as a CHILD process and
        # before anything in this one has touched the GPU -- and leave with its exit code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        raise SystemExit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # the native pieces are compiled first: nothing has initialised the GPU yet, so the compiler children are
    # not spawned from a GPU-holding (or, under rocprofv3, profiler-preloaded: use --no-build there) process.
    # torch is imported (not initialised) before libkomb_accel.so is loaded, so that one HIP runtime serves both.
    import datetime
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    if rank == 0 and not args.no_build:
        entry.compile_native()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the komb_accel path has no CPU fallback")
    # rehearsal hook (tests only): several ranks on ONE GPU, collectives over gloo
    one_device = os.environ.get("KOMB_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    exchange = "single GPU"
    data_group = None                       # the group the support vectors / frontier parts are summed over
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane (barriers, the max over the ranks' times) on gloo; the data exchange on RCCL when every
        # rank can use it -- the choice is made collectively (MIN over the ranks' flags), never rank by rank
        # (gloo announces its connections on fd 1: stdout carries the one JSON line and nothing else)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)   # ranks != 0 wait here for rank 0's build
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        ok = 0
        why = "one-device rehearsal"
        if mode in ("batch", "replicas"):
            why = "no exchange on the data path"
        elif not one_device:                                # (the sliced flow exchanges nothing either, but its alternatives do)
            try:
                data_group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
                probe = torch.ones(1, dtype=torch.int32, device="cuda")
                dist.all_reduce(probe, group=data_group)
                torch.cuda.synchronize()
                ok = 1 if int(probe.item()) == world else 0
                why = "" if ok else f"probe all-reduce returned {int(probe.item())}"
            except (RuntimeError, dist.DistBackendError) as exc:
                why = f"{type(exc).__name__}: {exc}"
                print(f"bench.py rank {rank}: RCCL unavailable ({why})", file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            exchange = f"RCCL all-reduce over {world} ranks, in place on the device buffers"
        else:
            data_group = None
            exchange = f"gloo through host memory ({why or 'RCCL unavailable on another rank'})"
        dist.barrier()
    import komb_amd

    nv, ncl, alpha, seed, desc = CONFIGS[args.config]
    t0 = time.perf_counter()
    uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed + (rank if mode == "batch" else 0))
    t_gen = time.perf_counter() - t0
    acc = komb_amd.KombAccel(device=local_rank)
    t0 = time.perf_counter()
    acc.from_edges(nv, uv)                       # a1 on the device; the symmetric CSR stays resident in HBM (nothing else does)
    t_build = time.perf_counter() - t0
    build_stats = acc.stats()
    extras = rank == 0 and world == 1 and not args.no_extras
    if not extras:
        del uv
    ne = acc.ne

    def barrier_sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        from komb_amd import distributed as kd

    def run_flow(flow, cold):
        if cold:
            acc.truss_unprepare()                # nothing of the previous step survives: the preparation is part of the step
        if flow == "sharded":                    # support all-reduce + the peel by edge range, one frontier exchange per sub-round
            kd.truss_run_sharded(acc, group=data_group, shard_peel=True)
        elif flow == "c4":                       # support all-reduce only
            kd.truss_run_sharded(acc, group=data_group, shard_peel=False)
        elif flow == "sliced":
            kd.truss_run_slice(acc)
        else:
            acc.truss_run()

    def timed(flow, cold, steps, with_phases=False):
        barrier_sync()
        t0 = time.perf_counter()
        ph = {k: 0.0 for k in PHASES}
        for _ in range(steps):
            run_flow(flow, cold)
            if with_phases:
                s = acc.stats()
                for k in ph:
                    ph[k] += s[k]
        barrier_sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, {k: v / steps for k, v in ph.items()}

    flow = mode if world > 1 and mode in ("sharded", "c4", "sliced") else "whole"
    for _ in range(args.warmup):
        run_flow(flow, True)
    dt, phase = timed(flow, True, args.steps, with_phases=True)
    st = acc.stats()
    ne_total = ne
    if mode == "batch":                                   # the job's units: the edges of all the ranks' graphs
        t = torch.tensor([ne], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ne_total = int(t.item())
    # the same step on a graph whose preparation is resident (what rounds 1-4 reported as the headline)
    dt_res, phase_res = timed(flow, False, args.steps, with_phases=True)
    alternatives = None
    if world > 1 and mode in ("sharded", "c4", "sliced") and not args.no_alternatives:
        alternatives = {}
        for name, f in (("shard_peel_and_support_allreduce", "sharded"), ("support_allreduce_only", "c4"), ("replicas_sliced", "sliced")):
            if f == flow:
                continue
            run_flow(f, True)
            d_alt, _ = timed(f, True, args.steps)
            alternatives[name] = {"ms_per_step": d_alt / args.steps * 1e3, "value": ne * args.steps / d_alt}
        acc.set_shard_peel(flow == "sharded")
    slices_ok = None
    if flow == "sliced":
        # outside the timed region: this rank's slice against a whole run of its own (values in the slice, zeros elsewhere)
        kd.truss_run_slice(acc)
        _, _, tr_s, sup_s = acc.truss_fetch(with_support=True)
        _, _, tr_w, sup_w = acc.run_truss(with_support=True)
        lo, hi = ne * rank // world, ne * (rank + 1) // world
        good = (np.array_equal(tr_s[lo:hi], tr_w[lo:hi]) and np.array_equal(sup_s[lo:hi], sup_w[lo:hi])
                and not tr_s[:lo].any() and not tr_s[hi:].any() and not sup_s[:lo].any() and not sup_s[hi:].any())
        t = torch.tensor([1 if good else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        slices_ok = bool(int(t.item()))
        del tr_s, sup_s, tr_w, sup_w

    # the roofline model's graph statistics (sum d^2, oriented items, max degree): a measurement-only call
    acc.graph_moments()
    gm = acc.stats()
    for k in ("sum_deg_sq", "wedge_items", "oriented_items", "max_degree"):
        st[k] = gm[k]

    # k-core of the same graph, reported alongside (BASELINE config C2's op)
    # (three calls, the fastest reported: the first one of a context allocates its scratch -- 0.2-0.5 ms of hipMalloc between kernels)
    core_first_ms = None
    core_stats = None
    for _ in range(3):
        if flow == "sharded":
            kd.core_run_sharded(acc, group=data_group)
        else:
            acc.core_run()
        cs = acc.stats()
        if core_first_ms is None:
            core_first_ms = cs["ms_core"]
        if core_stats is None or cs["ms_core"] < core_stats["ms_core"]:
            core_stats = cs
    core_ms = core_stats["ms_core"]

    faithful = first_call = c2_block = corea = fetch_block = None
    workload_csr = None
    if extras:
        # the results leaving the device -- not in the step: igraph_trussness leaves its vector in memory, and the endpoints of the
        # canonical edges are igraph_edge's answer afterwards (src/graph.cpp:529-532).  The first fetch of a graph makes the endpoint
        # list on the device (a pass over the symmetric CSR), every fetch copies 12 bytes per edge over PCIe
        bufs3 = [np.ones(ne, dtype=np.int32) for _ in range(3)]         # (touched: no page faults inside the copies)
        tf = []
        for args3 in ((None, None, bufs3[2]), tuple(bufs3), tuple(bufs3), (None, None, bufs3[2])):
            t1 = time.perf_counter(); acc.truss_fetch_into(*args3); tf.append((time.perf_counter() - t1) * 1e3)
        fetch_block = {"truss_only_ms": min(tf[0], tf[3]), "eu_ev_truss_first_ms": tf[1], "eu_ev_truss_again_ms": tf[2], "bytes_d2h_all": 12 * ne,
                       "note": "komb_truss_fetch after the timed steps, into touched host arrays, through the pinned staging buffers; the first fetch "
                               "that asks for endpoints makes the canonical edge list on the device (a pass over the symmetric CSR) and keeps it with the graph"}
        del bufs3
        deg_h, core_h = acc.core_fetch()
        # CoreA on the workload's own degrees / coreness (a9 + a10: komb_corea_scores): device time of the rank kernels, wall time of the call
        acc.get_anomaly_score(deg_h, core_h)
        t1 = time.perf_counter()
        score = acc.get_anomaly_score(deg_h, core_h)
        t_ca = time.perf_counter() - t1
        key_bits = int(core_h.max()) * nv + int(deg_h.max())
        corea = {"ms_device_rank_kernels": acc.stats()["ms_corea"], "ms_call_wall": t_ca * 1e3, "vertices": nv,
                 "max_score": float(score.max()), "key_max": key_bits,
                 "reference_int_key_overflows": key_bits >= 2**31,      # src/CoreA.h:122 computes the key in `int`
                 "alg_bytes_sorts": 2 * (2 * 12 * nv) * 8}               # two sorts of (8-byte key, 4-byte index) pairs, <= 8 radix passes, read + write
        del score
        # the runTruss-faithful variant (reference src/graph.cpp:470-473,502,508): trussness of the subgraph
        # induced by the max-coreness vertices, reported alongside (after the timed region)
        mask = (core_h == core_h.max()).astype(np.uint8)
        acc.truss_run(mask)                       # warm
        t_f = 1e9
        for _ in range(3):                        # best of three (a shared box's host hiccups have cost this 6 ms call 80 ms once)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            acc.truss_run(mask)
            torch.cuda.synchronize()
            t_f = min(t_f, time.perf_counter() - t1)
        fs = acc.stats()
        feu, fev, ftr = acc.truss_fetch()
        faithful = {"max_core_vertices": int(mask.sum()), "subgraph_edges": int(len(feu)),
                    "max_trussness": int(ftr.max()) if len(ftr) else 0, "ms": t_f * 1e3,
                    "ms_induce": fs["ms_orient"], "ms_prepare": fs["ms_prepare"]}
        del deg_h, core_h, mask, feu, fev, ftr
        # what KOMB would see: komb2 decomposes a graph ONCE -- a fresh context (every buffer still to be allocated),
        # graph build, then the first k-core and the first k-truss call (which makes the preparation)
        # (the timed context stays alive meanwhile: a hipMalloc that follows the hipFree of tens of GB waits for the driver to
        # finish releasing them -- 1.7 s measured -- which a process that decomposes one graph never sees)
        trials = []
        for _ in range(2):                        # two fresh contexts, the faster one reported (a one-shot wall time on a shared host)
            fresh = komb_amd.KombAccel(device=local_rank)
            t1 = time.perf_counter()
            fresh.from_edges(nv, uv)
            t_b2 = time.perf_counter() - t1
            b2 = fresh.stats()
            t1 = time.perf_counter(); fresh.core_run(); torch.cuda.synchronize(); t_c1 = time.perf_counter() - t1
            t1 = time.perf_counter(); fresh.truss_run(); torch.cuda.synchronize(); t_t1 = time.perf_counter() - t1
            p1 = fresh.stats()["ms_prepare"]
            t1 = time.perf_counter(); fresh.core_run(); torch.cuda.synchronize(); t_c2 = time.perf_counter() - t1
            t1 = time.perf_counter(); fresh.truss_run(); torch.cuda.synchronize(); t_t2 = time.perf_counter() - t1
            fresh.close()
            trials.append((t_b2 + t_t1, t_b2, b2, t_c1, t_t1, p1, t_c2, t_t2))
        both_ms = [round(t[0] * 1e3, 2) for t in trials]
        _, t_b2, b2, t_c1, t_t1, p1, t_c2, t_t2 = min(trials, key=lambda t: t[0])
        first_call = {"context": "fresh komb_ctx in this process (HIP runtime and kernels already loaded), every buffer still to be allocated",
                      "graph_build_ms": t_b2 * 1e3,
                      "graph_build_parts_ms": {"h2d": b2["ms_build_h2d"], "renumber_orient_lines": b2["ms_build_relabel"]},
                      "kcore_first_ms": t_c1 * 1e3, "kcore_second_ms": t_c2 * 1e3,
                      "ktruss_first_ms": t_t1 * 1e3, "ktruss_first_prepare_ms": p1, "ktruss_second_ms_resident": t_t2 * 1e3,
                      "build_plus_first_ktruss_ms": (t_b2 + t_t1) * 1e3, "build_plus_first_ktruss_ms_both_trials": both_ms}
        del uv
        if not args.no_cpu_baseline:
            rp_w, col_w = acc.get_csr()                    # the CPU baseline's all-cores leg runs on this very graph
            workload_csr = (rp_w, col_w)
        # BASELINE configs[1] (C2: |V|=1M, |E|~10M, k-core only) -- and the k-truss of the same graph
        if args.config != "c2":
            nv2, ncl2, alpha2, seed2, desc2 = CONFIGS["c2"]
            uv2 = komb_amd.gen_hug_edges(nv2, ncl2, alpha2, seed2)
            with komb_amd.KombAccel(device=local_rank) as a2:
                a2.from_edges(nv2, uv2)
                del uv2
                t1 = time.perf_counter(); a2.core_run(); torch.cuda.synchronize(); c2_first = time.perf_counter() - t1
                reps = 10
                c2_core = 1e9
                for _ in range(3):                # best of three batches of ten
                    t1 = time.perf_counter()
                    for _ in range(reps):
                        a2.core_run()
                    torch.cuda.synchronize()
                    c2_core = min(c2_core, (time.perf_counter() - t1) / reps)
                s2c = a2.stats()
                a2.truss_run()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    a2.truss_unprepare()
                    a2.truss_run()
                torch.cuda.synchronize()
                c2_truss = (time.perf_counter() - t1) / reps
                s2 = a2.stats()
                t1 = time.perf_counter()
                for _ in range(reps):
                    a2.truss_run()
                torch.cuda.synchronize()
                c2_truss_res = (time.perf_counter() - t1) / reps
                bc2 = 16 * nv2 + 24 * a2.ne
                c2_block = {"workload": "C2: synthetic power-law unitig graph |V|=1M |E|~10M (BASELINE configs[1]: k-core only; k-truss alongside)",
                            "nv": nv2, "ne": a2.ne, "triangles": s2["triangles"],
                            "kcore": {"ms": c2_core * 1e3, "first_call_ms": c2_first * 1e3, "edges_per_s": a2.ne / c2_core,
                                      "levels": s2c["core_levels"], "launches": s2c["core_launches"], "max_coreness": s2c["max_coreness"],
                                      "alg_bytes": bc2, "GBps": bc2 / c2_core / 1e9, "frac_of_hbm_peak": bc2 / c2_core / 1e9 / HBM_PEAK_GBS},
                            "ktruss": {"ms_per_step": c2_truss * 1e3, "ms_per_step_resident": c2_truss_res * 1e3, "edges_per_s": a2.ne / c2_truss,
                                       "max_trussness": s2["max_trussness"],
                                       "phases_ms": {k: s2[k] for k in ("ms_prepare", "ms_tri_fill", "ms_sort", "ms_compact", "ms_peel", "ms_truss_local", "ms_gather")}}}

    if rank == 0:
        ab = algorithmic_bytes(st)
        shard = mode in ("sharded", "c4")
        sliced = mode == "sliced"
        # the peel = the launches of k_peel_step<Truss> (all of them, including the no-op launches of the last blind
        # batch, as rocprofv3 counts them) + the single-workgroup LDS tail (setup kernels + k_truss_tail)
        kernels = {"k_peel_step<Truss>": (phase["ms_peel"] - phase["ms_tail"] - phase["ms_truss_local"] - phase["ms_exchange"], st["truss_launches"], ab["peel"])}
        # the preparation's kernels (truss_prep.hip), each under its own symbol: the "dominant kernel" rule looks at symbols
        kernels["k_prep_kept (+ _heavy, k_prep_dplus, scan): canonical edges -> oriented rows"] = (phase["ms_prep_edges"], 4, ab["prep_edges"])
        kernels["k_prep_rows (+ _heavy): row sort, lines, canonical map"] = (phase["ms_prep_rows"], 2, ab["prep_rows"])
        kernels["k_prep_vertex + (degree,id) radix sort + scans + task table"] = (
            phase["ms_prepare"] - phase["ms_prep_edges"] - phase["ms_prep_rows"], 10, ab["prep_vertex"])
        if st["truss_tail_runs"]:
            kernels["k_truss_tail"] = (phase["ms_tail"], st["truss_tail_runs"], 0)
        if st["truss_local_units"]:
            kernels["local finish (number + collect + k_local_step sweeps)"] = (phase["ms_truss_local"], st["truss_local_sweeps"], 0)
        if phase["ms_tri_count"] > 0:            # the counting enumeration (sharded runs: this rank's share; two-pass: all of it)
            kernels["k_wedges<count>" if shard else "k_triangles<count>"] = (phase["ms_tri_count"], 1, ab["tri_count"] // (world if shard else 1))
        layout = st["index_layout"]
        if layout == 0:                          # ONE enumeration: dense own-role blocks + record stream, sort, merge
            # the enumeration is priced at SURVEY 8(d)'s B_sup verbatim (its stores -- 8 bytes per own-role entry, 12 per
            # record -- are not counted); the sort at the 2 radix passes it runs over 12-byte records, read + write
            kernels["k_wedges<stream>"] = (phase["ms_tri_fill"], 1, ab["tri_count"])
            kernels["record sort (rocPRIM radix, by bin: 2 passes)"] = (phase["ms_sort"], 1, ab["sort"])
            kernels["k_bin_offsets + k_bin_count + k_bin_finish"] = (phase["ms_compact"], 3, ab["finish"])
        else:
            kernels["k_triangles<single> (exact slices)"] = (phase["ms_tri_fill"], 1, ab["tri_count"])
        # (sliced runs: the pass covers every edge, its stores this rank's slice)
        kernels["k_truss_results"] = (phase["ms_gather"], 1, ab["gather"])
        dom = max(kernels, key=lambda k: kernels[k][0])
        ms, launches, nbytes = kernels[dom]
        achieved = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic, traffic_max, traffic_src = measured_traffic(args.config, dom)
        # the longest single launch (the other reading of "dominant"): the enumeration's one launch per step
        by_launch = max(kernels, key=lambda k: kernels[k][0] / max(kernels[k][1], 1))
        bl_ms, bl_n, bl_bytes = kernels[by_launch]
        roofline = {
            "bound": "hbm", "kernel": dom,
            "dominant_rule": "the kernel symbol with the largest time per step (all its launches together), as in rounds 1-4; "
                             "by_launch names the kernel with the longest single launch",
            "by_launch": {"kernel": by_launch, "avg_launch_us": bl_ms * 1e3 / max(bl_n, 1), "launches_per_step": bl_n,
                          "achieved": (bl_bytes / (bl_ms * 1e-3) / 1e9 if bl_ms > 0 else 0.0),
                          "frac": (bl_bytes / (bl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if bl_ms > 0 else 0.0)}, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            # FETCH_SIZE is exact for line gathers and 1/2 for coalesced streams (calibrated on this box):
            # traffic = FETCH+WRITE is the lower bound, traffic_max = 2*FETCH+WRITE the upper bound
            "traffic_max": traffic_max, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": nbytes / max(launches, 1),
            "launches_per_step": launches, "avg_launch_us": ms * 1e3 / max(launches, 1),
            "algorithmic_bytes_per_step": nbytes,
            "per_kernel": {k: {"ms_per_step": v[0], "launches": v[1], "alg_bytes": v[2],
                               "GBps": (v[2] / (v[0] * 1e-3) / 1e9 if v[0] > 0 else 0.0)} for k, v in kernels.items()},
            "survey_formula_peel_bytes": ab["survey_peel"],
            # SURVEY 8(d) verbatim: B_sup over the enumeration kernel's own event time, and the whole step's algorithmic bytes
            # (the preparation's included) over the whole step
            "survey_B_sup_bytes": ab["tri_count"],
            "survey_B_sup_frac": (ab["tri_count"] / ((phase["ms_tri_fill"] + phase["ms_tri_count"]) * 1e-3) / 1e9 / HBM_PEAK_GBS
                                  if phase["ms_tri_fill"] + phase["ms_tri_count"] > 0 else None),
            "whole_step_bytes": sum(v[2] for v in kernels.values()),
            "whole_step_frac": sum(v[2] for v in kernels.values()) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
            # SURVEY 8(d)'s secondary ceilings.  The peel engines are bounded by their decrements, not by bytes: the chip performs
            # ~27 G scattered 32-bit atomics per second -- one memory-side request per 64-byte line a wave instruction touches,
            # returning or not, whatever the table's size (4 MB .. 400 MB): scripts/calib/atomic_rate.hip,
            # profiles/r05_atomic_rate.txt; lanes that share a line share the request, which is how the truss engine reaches
            # ~50 G decrements/s on slices whose edges are neighbours (profiles/r04_peel_first_step_ablation.txt).  A triangle
            # owes at most two decrements; k-core one per (peeled vertex, live neighbour) slot.
            "secondary_ceilings": {
                "atomic_line_requests_G_per_s_calibrated": 27.0,
                "truss_decrement_rate_G_per_s_calibrated": 50.0,
                "calibration": "profiles/r05_atomic_rate.txt (this round), profiles/r04_peel_first_step_ablation.txt",
                "truss_decrements_upper_bound": 2 * st["triangles"],
                "truss_decrement_floor_ms": 2 * st["triangles"] / 50e9 * 1e3,
                "truss_engine_ms": kernels["k_peel_step<Truss>"][0],
                "kcore_slot_visits_upper_bound": 2 * ne, "kcore_decrement_floor_ms": 2 * ne / 27e9 * 1e3,
                "kcore_launches": core_stats["core_launches"], "kcore_ms": core_ms,
            },
        }
        par = {
            "single": "single",
            "batch": f"batch: {world} independent graphs (seed + rank), one per GPU, no exchange on the data path",
            "replicas": f"same graph on {world} ranks, replicas: every rank runs the whole single-GPU path, no exchange",
            "sliced": (f"same graph on {world} ranks: preparation, support, index and peel on every rank, the canonical results gathered in "
                       f"{world} slices, one per rank; no exchange on the data path (slices verified after the timed region: {slices_ok})"),
            "c4": (f"same graph on {world} ranks: triangle-support counting sharded by task range + one all-reduce of the per-edge support "
                   f"vector ({exchange}); preparation, incidence fill, peel and gather replicated on every rank"),
            "sharded": (f"same graph on {world} ranks, edge-range partition: triangle-support counting sharded by task range + one all-reduce of "
                        f"the per-edge support vector; peel: live supports owned by internal-edge-id range, every rank walks the whole exchanged "
                        f"frontier and applies the decrements it owns, {st['shard_exchanges']} exchanges per step ({exchange}); preparation, "
                        f"incidence fill and gather replicated"),
        }[mode]
        workload = desc if world == 1 else {
            "batch": f"{world} x [{desc}] (one graph per rank)",
            "replicas": f"{world} x [{desc}] (replicas of one graph)",
            "sliced": f"the same graph as [{desc}] on {world} GPUs: every rank runs the whole k-truss path and materialises its slice of the canonical results",
            "c4": f"C4: the same graph as [{desc}] on {world} GPUs, k-truss with the triangle support sharded + all-reduce",
            "sharded": f"C4: the same graph as [{desc}] on {world} GPUs, k-truss with the triangle support sharded + all-reduce, the peel sharded by edge range + one frontier exchange per sub-round",
        }[mode]
        out = {
            "metric": ("peeled edges/sec (k-truss)" if mode != "batch" else
                       f"aggregate peeled edges/sec (k-truss) over {world} independent graphs"),
            "value": ne_total * args.steps / dt,
            "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if mode == "batch" else "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            # the same step with the graph's preparation resident (rounds 1-4's headline definition)
            "value_resident": ne_total * args.steps / dt_res, "ms_per_step_resident": dt_res / args.steps * 1e3,
            "step_definition": "komb_truss_unprepare + komb_truss_run per step: preparation ((degree,id) renumbering, orientation, canonical "
                               "map, lines, tasks) + support + index + peel + gather, from the resident symmetric CSR; *_resident: "
                               "komb_truss_run on a graph that kept its preparation",
            "config": {"workload": workload,
                       "nv": nv, "ne": ne, "triangles": st["triangles"], "alpha": alpha, "seed": seed,
                       "max_degree": st["max_degree"], "max_trussness": st["max_trussness"],
                       "max_coreness": core_stats["max_coreness"],
                       "truss_levels": st["truss_levels"], "truss_subrounds": st["truss_subrounds"],
                       "truss_scans": st["truss_scans"], "truss_launches": st["truss_launches"],
                       "index_layout": {0: "record stream", 2: "exact two-pass"}[st["index_layout"]],
                       "tri_records": st["tri_records"], "engine_flags": st["engine_flags"],
                       **({"shard_peel": {"exchanges": st["shard_exchanges"], "exchange_words": st["exchange_words"]}} if mode == "sharded" else {}),
                       "truss_local": {"edges": st["truss_local_units"], "index_entries": st["truss_local_items"],
                                       "sweeps": st["truss_local_sweeps"]},
                       "parallelism": par},
            "phases_ms": phase, "phases_ms_resident": phase_res,
            **({"alternatives": alternatives,
                "multi_gpu_note": "never measured on more than one GPU by the builder (the pool grants one); DESIGN.md section 6 has the predicted N = 2/4/8 times"}
               if world > 1 else {}),
            "kcore": {"ms": core_ms, "first_call_ms": core_first_ms, "edges_per_s": ne / (core_ms * 1e-3) if core_ms > 0 else None,
                      "levels": core_stats["core_levels"], "launches": core_stats["core_launches"],
                      **({"sharded": {"exchanges": core_stats["shard_exchanges"], "ms_exchange": core_stats["ms_exchange"],
                                      "exchange_words": core_stats["exchange_words"]}} if mode == "sharded" else {}),
                      "local": {"vertices": core_stats["core_local_units"], "index_entries": core_stats["core_local_items"],
                                "sweeps": core_stats["core_local_sweeps"], "ms": core_stats["ms_core_local"]},
                      "alg_bytes": 16 * nv + 24 * ne,
                      "GBps": (16 * nv + 24 * ne) / (core_ms * 1e-3) / 1e9 if core_ms > 0 else None,
                      "frac_of_hbm_peak": (16 * nv + 24 * ne) / (core_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if core_ms > 0 else None},
            "corea": corea,
            # the library reads no environment variable (ABI 7): its switches are per-context options nobody sets here
            "library_options_set": [],
            "runtruss_faithful": faithful,
            "results_fetch": fetch_block,
            "first_call": first_call,
            "c2": c2_block,
            "setup_s": {"generate": t_gen, "graph_build_incl_h2d": t_build, "device_build_ms": build_stats["ms_build"],
                        "h2d_ms": build_stats["ms_build_h2d"],
                        "note": "first graph build of the process: includes the HIP runtime's first large allocations and first launches of "
                                "every build kernel (komb_create has loaded the code object); first_call.graph_build_ms is a second build in the "
                                "same process.  The graph object holds the symmetric CSR only; the k-truss preparation is made inside the first "
                                "k-truss call (and inside every timed step here)"},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed at N=1 only
            out["cpu_baseline"] = cb = cpu_baseline(workload_csr, ne)
            # context, not credit: the GPU figure over the CPU port's, edges/s against edges/s (north_star asks for >= 10x)
            cb["gpu_over_cpu"] = {"1_thread": out["value"] / cb["value"],
                                  **({"all_cores": out["value"] / cb["all_cores"]["value"]} if "all_cores" in cb else {})}
        print(json.dumps(out))
    acc.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
