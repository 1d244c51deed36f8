#!/usr/bin/env python3
"""bench.py -- peeled edges/sec of the k-truss hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the k-truss path (triangle support, incidence
index, level-synchronous peel, canonical gather: komb_truss_run) over the
synthetic power-law unitig graph, with the graph already resident in HBM when
the timed region starts (the resident graph object includes the oriented CSR in
(degree,id)-ranked internal ids; nothing a previous komb_truss_run computed is
reused).  Rank 0 prints ONE JSON line.

N=1 workload = BASELINE.json configs[2] (|V|=10M, |E|~100M, full k-truss, the
configuration the metric is quoted on).  --config c2 selects configs[1]
(|V|=1M, |E|~10M); the k-core time of the same graph is reported alongside.

N>1: the SAME graph on every rank ("scaling": "strong", `value` = |E| / the slowest
rank's time).  Support, index and peel of ONE graph do not shard across GPUs at a
profit (DESIGN.md section 6 has the byte counts), so every rank runs them whole and
materialises only ITS slice of the canonical results (komb_truss_run_slice): no
collective on the data path, never slower than N = 1 by construction.  Opt-in:
--c4-allreduce = BASELINE.json configs[3] to the letter (the triangle support counted
in shards and summed with one RCCL all-reduce over xGMI, then everything replicated:
the count is redundant work, 0.80-0.84 x by arithmetic), --shard-peel on top of it,
--replicas (nothing sliced) and --batch (one graph per rank, seed + rank, nothing
exchanged: weak scaling, its own metric name).
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
CPU_SAMPLE = (3_000_000, 7_350_000, 2.6, 42)  # ~30M edges: ~80 s of single-thread CPU work, ~13 s on 16 threads (one graph for both)


def algorithmic_bytes(st):
    """Bytes each phase must move, per step (DESIGN.md 'Algorithmic bytes').
    E = edges, T = triangles, O = sum over oriented edges of d+(a)+d+(b), R = record positions of the stream."""
    E, T, O, R = st["ne"], st["triangles"], st["oriented_items"], st["tri_records"]
    tri_count = 12 * E + 4 * O + 24 * T                 # SURVEY 8(d) B_sup
    tri_fill = 12 * E + 4 * O + 24 * T + 24 * T         # same reads, 3 cursor RMW + 3 pair stores per triangle
    peel = 8 * E + 24 * T + 24 * T + 16 * T             # truss+stamp per edge; slice entries; two stamps per entry; 2 RMW per triangle
    survey_peel = 8 * E + 4 * st["sum_deg_sq"] + 16 * T # SURVEY 8(d) B_peel (merge re-intersection; not what we do)
    sort = 2 * 2 * 12 * R                               # the passes that run: two radix passes over 12-byte records, read + write
    finish = 16 * E + 48 * T + 4 * R                    # supports and slice pairs; every entry in once (record or block entry), out once
    gather = 20 * E + 20 * E                            # resolve: stamp + slice pair in, (trussness, support) out; gather: map + pair in, two words out
    return {"tri_count": tri_count, "tri_fill": tri_fill, "peel": peel, "survey_peel": survey_peel,
            "sort": sort, "finish": finish, "gather": gather}


def measured_traffic(config, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE, separate runs of this same command; scripts/prof_summary.py), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{config}_*traffic.json")))
    if not files:
        return None, None, None
    data = json.load(open(files[-1]))
    name = kernel
    rec = data.get(name)
    if not rec:
        return None, None, os.path.basename(files[-1])
    return rec["hbm_bytes_per_launch"], rec.get("hbm_bytes_per_launch_max"), os.path.basename(files[-1])


def cpu_baseline():
    """The CPU restatement of igraph's trussness (oracle/, test infrastructure) timed on this box's host cores, next to
    the GPU figure: single thread pinned to one core (igraph is single-threaded and the reference calls it from one
    thread, src/graph.cpp:508) on a bounded sample of the same generator (|E| ~ 30M), and the OpenMP all-cores variant on
    the same sample.  The library is compiled on this machine for its own instruction set (oracle/Makefile target native)."""
    import numpy as np  # noqa: F401
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
    nv, ncl, alpha, seed = CPU_SAMPLE
    uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
    rowptr, col = O.simplify(nv, uv)
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
                     f"{'-march=native' if native else 'portable build'}) on |V|={nv} |E|={ne} of the same generator: {dt:.1f} s",
           "cpu_model": model, "host_cpus": len(cpus),
           "full_graph_recorded": "the full C3 graph (|E|=100.1M) took 335 s = 3.0e5 edges/s on one thread of a GPU box "
                                  "(profiles/r01_c3_full_parity.log)"}
    if native:
        # the GPU boxes show every CPU of the host in the affinity mask but grant a share of 16 per GPU: more threads
        # than that only thrash (256 threads: 116 s against 15 s on one)
        nthr = min(len(cpus), 16)
        nv2, ne2 = nv, ne                                  # the same sample graph
        t0 = time.perf_counter()
        O.trussness_native(rowptr, col, nthr)
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": ne2 / dt2, "unit": "edges/s", "cores": nthr, "kind": "port",
                            "sample": f"oracle orc_trussness_omp (parallel supports + level-synchronous parallel peel, OpenMP, "
                                      f"{nthr} threads) on |V|={nv2} |E|={ne2}: {dt2:.1f} s"}
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
    ap.add_argument("--same-graph", action="store_true", help="(the default for N > 1; accepted for older command lines)")
    ap.add_argument("--c4-allreduce", "--shard", dest="shard", action="store_true",
                    help="N > 1, opt-in: BASELINE configs[3] to the letter -- the support-counting enumeration split by source-vertex range "
                         "+ one all-reduce of the |E|+1 support words, then index build, peel and gather replicated (rounds 1-3's default)")
    ap.add_argument("--shard-peel", action="store_true",
                    help="N > 1, opt-in, on top of --c4-allreduce (implies it): the peel sharded by edge range too, the ranks' parts of the frontier "
                         "exchanged every sub-round (SURVEY 8(e)'s partition, komb_set_shard_peel); slower than the replicated peel "
                         "on one node (DESIGN.md section 6), so never what the plain command runs.  The k-core reported alongside "
                         "then runs komb_core_run_sharded")
    ap.add_argument("--no-build", action="store_true",
                    help="load the prebuilt libkomb_accel.so, spawn no compiler (use under rocprofv3)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed workload (+ the k-core of the same graph): no runTruss-faithful variant, no first-call "
                         "figures, no C2 block -- for rocprofv3 runs, whose per-kernel averages should see the timed launches only")
    ap.add_argument("--faithful", action="store_true", help="(the default since round 4; accepted for older command lines)")
    args = ap.parse_args()
    if args.batch and (args.replicas or args.same_graph or args.shard):
        raise SystemExit("--batch runs one graph per rank: it cannot be combined with --replicas / --same-graph / --shard")
    if args.replicas and args.shard:
        raise SystemExit("--replicas runs the unsharded path on every rank: it cannot be combined with --shard")
    if args.shard_peel and (args.batch or args.replicas or args.gpus < 2):
        raise SystemExit("--shard-peel shards one graph over N > 1 ranks: it cannot be combined with --batch / --replicas / --gpus 1")
    args.batch = args.gpus > 1 and args.batch
    # N > 1: --c4-allreduce = BASELINE configs[3] to the letter (support counting sharded + one all-reduce);
    # default = every rank peels the whole graph and materialises its slice of the results, no exchange
    args.shard = args.gpus > 1 and (args.shard or args.shard_peel) and not args.batch and not args.replicas
    args.slice = args.gpus > 1 and not args.shard and not args.batch and not args.replicas

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
    data_group = None                       # the group the support vectors are summed over
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
        if not args.shard:
            why = "no exchange on the data path"
        elif not one_device:
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
            exchange = "RCCL all-reduce in place on the device buffer"
        else:
            data_group = None
            exchange = f"gloo through host memory ({why or 'RCCL unavailable on another rank'})"
    host_exchange = world > 1
    if world > 1:
        dist.barrier()
    import komb_amd

    nv, ncl, alpha, seed, desc = CONFIGS[args.config]
    t0 = time.perf_counter()
    uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed + (rank if args.batch else 0))
    t_gen = time.perf_counter() - t0
    acc = komb_amd.KombAccel(device=local_rank)
    t0 = time.perf_counter()
    acc.from_edges(nv, uv)                       # a1 on the device; the graph stays resident in HBM
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

    # N > 1 by default: the same graph on every rank, the whole path on every rank, each rank's slice of the canonical
    # results materialised (komb_truss_run_slice; no exchange).  --c4-allreduce: the support-counting enumeration split by
    # source-vertex range + one all-reduce of the |E|+1 support words, then everything replicated (BASELINE configs[3] to
    # the letter; DESIGN.md section 6 has the arithmetic).  --replicas / --batch are opt-in too.
    shard = world > 1 and args.shard
    sliced = world > 1 and args.slice

    def step():
        if shard:
            # support phase sharded by vertex range + all-reduce; with --shard-peel the peel by edge range + one exchange per sub-round
            kd.truss_run_sharded(acc, group=data_group, shard_peel=args.shard_peel)
        elif sliced:
            kd.truss_run_slice(acc)
        else:
            acc.truss_run()

    for _ in range(args.warmup):
        step()
    barrier_sync()
    t0 = time.perf_counter()
    phase = {k: 0.0 for k in ("ms_orient", "ms_tri_count", "ms_allreduce", "ms_tri_fill", "ms_sort", "ms_compact", "ms_peel", "ms_tail",
                              "ms_truss_local", "ms_gather", "ms_exchange")}
    for _ in range(args.steps):
        step()
        s = acc.stats()
        for k in phase:
            phase[k] += s[k]
    barrier_sync()
    dt = time.perf_counter() - t0
    ne_total = ne
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if args.batch:                                    # the job's units: the edges of all the ranks' graphs
            t = torch.tensor([ne], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            ne_total = int(t.item())
    st = acc.stats()
    for k in phase:
        phase[k] /= args.steps
    slices_ok = None
    if sliced:
        # outside the timed region: this rank's slice against a whole run of its own (values in the slice, zeros elsewhere)
        import numpy as np
        _, _, tr_s, sup_s = acc.truss_fetch(with_support=True)
        _, _, tr_w, sup_w = acc.run_truss(with_support=True)
        lo, hi = ne * rank // world, ne * (rank + 1) // world
        good = (np.array_equal(tr_s[lo:hi], tr_w[lo:hi]) and np.array_equal(sup_s[lo:hi], sup_w[lo:hi])
                and not tr_s[:lo].any() and not tr_s[hi:].any() and not sup_s[:lo].any() and not sup_s[hi:].any())
        t = torch.tensor([1 if good else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        slices_ok = bool(int(t.item()))
        del tr_s, sup_s, tr_w, sup_w

    # k-core of the same graph, reported alongside (BASELINE config C2's op)
    if shard and args.shard_peel:
        kd.core_run_sharded(acc, group=data_group)
    else:
        acc.core_run()
    core_ms = acc.stats()["ms_core"]
    core_stats = acc.stats()

    # the runTruss-faithful variant (reference src/graph.cpp:470-473,502,508): trussness of the subgraph
    # induced by the max-coreness vertices, reported alongside (after the timed region)
    faithful = first_call = c2_block = None
    if extras:
        deg_h, core_h = acc.core_fetch()
        mask = (core_h == core_h.max()).astype(np.uint8)
        acc.truss_run(mask)                       # warm
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        acc.truss_run(mask)
        torch.cuda.synchronize()
        t_f = time.perf_counter() - t1
        feu, fev, ftr = acc.truss_fetch()
        faithful = {"max_core_vertices": int(mask.sum()), "subgraph_edges": int(len(feu)),
                    "max_trussness": int(ftr.max()) if len(ftr) else 0, "ms": t_f * 1e3}
        del deg_h, core_h, mask, feu, fev, ftr
        # what KOMB would see: komb2 decomposes a graph ONCE -- a fresh context (every buffer still to be allocated),
        # graph build, then the first k-core and the first k-truss call
        # (the timed context stays alive meanwhile: a hipMalloc that follows the hipFree of tens of GB waits for the driver to
        # finish releasing them -- 1.7 s measured -- which a process that decomposes one graph never sees)
        fresh = komb_amd.KombAccel(device=local_rank)
        t1 = time.perf_counter()
        fresh.from_edges(nv, uv)
        t_b2 = time.perf_counter() - t1
        b2 = fresh.stats()
        t1 = time.perf_counter(); fresh.core_run(); torch.cuda.synchronize(); t_c1 = time.perf_counter() - t1
        t1 = time.perf_counter(); fresh.truss_run(); torch.cuda.synchronize(); t_t1 = time.perf_counter() - t1
        t1 = time.perf_counter(); fresh.core_run(); torch.cuda.synchronize(); t_c2 = time.perf_counter() - t1
        t1 = time.perf_counter(); fresh.truss_run(); torch.cuda.synchronize(); t_t2 = time.perf_counter() - t1
        fresh.close()
        first_call = {"context": "fresh komb_ctx in this process (HIP runtime and kernels already loaded), every buffer still to be allocated",
                      "graph_build_ms": t_b2 * 1e3,
                      "graph_build_parts_ms": {"h2d": b2["ms_build_h2d"], "renumber_orient_lines": b2["ms_build_relabel"]},
                      "kcore_first_ms": t_c1 * 1e3, "kcore_second_ms": t_c2 * 1e3,
                      "ktruss_first_ms": t_t1 * 1e3, "ktruss_second_ms": t_t2 * 1e3,
                      "build_plus_first_ktruss_ms": (t_b2 + t_t1) * 1e3}
        del uv
        # BASELINE configs[1] (C2: |V|=1M, |E|~10M, k-core only) -- and the k-truss of the same graph
        if args.config != "c2":
            nv2, ncl2, alpha2, seed2, desc2 = CONFIGS["c2"]
            uv2 = komb_amd.gen_hug_edges(nv2, ncl2, alpha2, seed2)
            with komb_amd.KombAccel(device=local_rank) as a2:
                a2.from_edges(nv2, uv2)
                del uv2
                t1 = time.perf_counter(); a2.core_run(); torch.cuda.synchronize(); c2_first = time.perf_counter() - t1
                reps = 10
                t1 = time.perf_counter()
                for _ in range(reps):
                    a2.core_run()
                torch.cuda.synchronize()
                c2_core = (time.perf_counter() - t1) / reps
                s2c = a2.stats()
                a2.truss_run()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    a2.truss_run()
                torch.cuda.synchronize()
                c2_truss = (time.perf_counter() - t1) / reps
                s2 = a2.stats()
                bc2 = 16 * nv2 + 24 * a2.ne
                c2_block = {"workload": "C2: synthetic power-law unitig graph |V|=1M |E|~10M (BASELINE configs[1]: k-core only; k-truss alongside)",
                            "nv": nv2, "ne": a2.ne, "triangles": s2["triangles"],
                            "kcore": {"ms": c2_core * 1e3, "first_call_ms": c2_first * 1e3, "edges_per_s": a2.ne / c2_core,
                                      "levels": s2c["core_levels"], "launches": s2c["core_launches"], "max_coreness": s2c["max_coreness"],
                                      "alg_bytes": bc2, "GBps": bc2 / c2_core / 1e9, "frac_of_hbm_peak": bc2 / c2_core / 1e9 / HBM_PEAK_GBS},
                            "ktruss": {"ms_per_step": c2_truss * 1e3, "edges_per_s": a2.ne / c2_truss, "max_trussness": s2["max_trussness"],
                                       "phases_ms": {k: s2[k] for k in ("ms_tri_fill", "ms_sort", "ms_compact", "ms_peel", "ms_truss_local", "ms_gather")}}}

    if rank == 0:
        ab = algorithmic_bytes(st)
        # the peel = the launches of k_peel_step<Truss> (all of them, including the no-op launches of the last blind
        # batch, as rocprofv3 counts them) + the single-workgroup LDS tail (setup kernels + k_truss_tail)
        kernels = {"k_peel_step<Truss>": (phase["ms_peel"] - phase["ms_tail"] - phase["ms_truss_local"], st["truss_launches"], ab["peel"])}
        if st["truss_tail_runs"]:
            kernels["k_truss_tail"] = (phase["ms_tail"], st["truss_tail_runs"], 0)
        if st["truss_local_units"]:
            kernels["local finish (number + collect + k_local_step sweeps)"] = (phase["ms_truss_local"], st["truss_local_sweeps"], 0)
        if phase["ms_tri_count"] > 0:            # the counting enumeration (sharded runs: this rank's share; two-pass: all of it)
            kernels["k_triangles<count>"] = (phase["ms_tri_count"], 1, ab["tri_count"] // (world if shard else 1))
        layout = st["index_layout"]
        if layout == 0:                          # ONE enumeration: dense own-role blocks + record stream, sort, merge
            # the enumeration is priced at SURVEY 8(d)'s B_sup verbatim (its stores -- 8 bytes per own-role entry, 12 per
            # record -- are not counted); the sort at the 2 radix passes it runs over 12-byte records, read + write
            enum_name = "k_triangles<stream>" if os.environ.get("KOMB_ENUM") == "probe" else "k_wedges<stream>"
            kernels[enum_name] = (phase["ms_tri_fill"], 1, ab["tri_count"])
            kernels["record sort (rocPRIM radix, by bin: 2 passes)"] = (phase["ms_sort"], 1, ab["sort"])
            kernels["k_bin_offsets + k_bin_count + k_bin_finish"] = (phase["ms_compact"], 3, ab["finish"])
        elif layout == 1:                        # ONE enumeration into bounded slices + dense compaction
            kernels["k_triangles<single>"] = (phase["ms_tri_fill"], 1, ab["tri_count"])
            kernels["k_compact_inc"] = (phase["ms_compact"], 1, 16 * st["ne"] + 48 * st["triangles"])
        else:
            kernels["k_triangles<single> (exact slices)"] = (phase["ms_tri_fill"], 1, ab["tri_count"])
        # (sliced runs: the resolve pass covers every edge, the gather this rank's slice)
        kernels["k_truss_resolve + k_gather_canonical"] = (phase["ms_gather"], 2, ab["gather"] // 2 + ab["gather"] // 2 // (world if sliced else 1))
        if phase["ms_orient"] > 0.01:               # induced-subgraph runs only: the slot filter + the subgraph's vertex lines
            kernels["k_slot_filter<PredMask> + k_vertex_lines"] = (phase["ms_orient"], 3, 0)
        dom = max(kernels, key=lambda k: kernels[k][0])
        ms, launches, nbytes = kernels[dom]
        achieved = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic, traffic_max, traffic_src = measured_traffic(args.config, dom)
        # the longest single launch (the other reading of "dominant"): the enumeration's one launch per step
        by_launch = max(kernels, key=lambda k: kernels[k][0] / max(kernels[k][1], 1))
        bl_ms, bl_n, bl_bytes = kernels[by_launch]
        roofline = {
            "bound": "hbm", "kernel": dom,
            "dominant_rule": "the kernel symbol with the largest time per step (all its launches together), as in rounds 1-3; "
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
            # SURVEY 8(d) verbatim: B_sup over the enumeration kernel's own event time (since round 3 the headline figure is
            # priced the same way), and the whole step's algorithmic bytes over the whole step
            "survey_B_sup_bytes": ab["tri_count"],
            "survey_B_sup_frac": (ab["tri_count"] / ((phase["ms_tri_fill"] + phase["ms_tri_count"]) * 1e-3) / 1e9 / HBM_PEAK_GBS
                                  if phase["ms_tri_fill"] + phase["ms_tri_count"] > 0 else None),
            "whole_step_bytes": sum(v[2] for v in kernels.values()),
            "whole_step_frac": sum(v[2] for v in kernels.values()) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
            # SURVEY 8(d)'s secondary ceilings.  The peel engine is bounded by its decrements, not by bytes: a returning atomic on a
            # per-edge counter costs ~20 ps (50 G/s over the chip, measured by switching them off / adding them in a debug build:
            # profiles/r04_peel_first_step_ablation.txt; the same with the counters packed into a cache-resident array), and a
            # triangle owes at most two; k-core: one per (peeled vertex, live neighbour) slot, launches x the ~15 us step floor
            "secondary_ceilings": {
                "decrement_rate_G_per_s_measured": 50.0,
                "truss_decrements_upper_bound": 2 * st["triangles"],
                "truss_decrement_floor_ms": 2 * st["triangles"] / 50e9 * 1e3,
                "truss_engine_ms": kernels["k_peel_step<Truss>"][0] if "k_peel_step<Truss>" in kernels else None,
                "kcore_slot_visits_upper_bound": 2 * ne, "kcore_decrement_floor_ms": 2 * ne / 50e9 * 1e3,
                "kcore_launches": core_stats["core_launches"], "kcore_ms": core_ms,
            },
        }
        out = {
            "metric": ("peeled edges/sec (k-truss)" if not args.batch else
                       f"aggregate peeled edges/sec (k-truss) over {world} independent graphs"),
            "value": ne_total * args.steps / dt,
            "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if args.batch else "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": desc if world == 1 else
                       (f"C4: the same graph as [{desc}] on {world} GPUs, k-truss with the triangle support sharded + all-reduce"
                        + (", the peel sharded by edge range + one frontier exchange per sub-round" if args.shard_peel else "") if shard else
                        (f"the same graph as [{desc}] on {world} GPUs: every rank runs the whole k-truss path and materialises its slice of the canonical results"
                         if sliced else f"{world} x [{desc}] ({'one graph per rank' if args.batch else 'replicas of one graph'})")),
                       "nv": nv, "ne": ne, "triangles": st["triangles"], "alpha": alpha, "seed": seed,
                       "max_degree": st["max_degree"], "max_trussness": st["max_trussness"],
                       "max_coreness": core_stats["max_coreness"],
                       "truss_levels": st["truss_levels"], "truss_subrounds": st["truss_subrounds"],
                       "truss_scans": st["truss_scans"], "truss_launches": st["truss_launches"],
                       "index_layout": ["record stream", "bounded slices", "exact two-pass"][st["index_layout"]],
                       "tri_records": st["tri_records"],
                       **({"shard_peel": {"exchanges": st["shard_exchanges"], "exchange_words": st["exchange_words"]}} if args.shard_peel else {}),
                       "truss_local": {"edges": st["truss_local_units"], "index_entries": st["truss_local_items"],
                                       "sweeps": st["truss_local_sweeps"]},
                       "parallelism": "single" if world == 1 else
                       (f"batch: {world} independent graphs (seed + rank), one per GPU, no exchange on the data path" if args.batch else
                        (f"same graph on {world} ranks: triangle-support counting sharded by source-vertex range + one all-reduce of the "
                         f"per-edge support vector ({exchange}); incidence fill"
                         + (", gather replicated; peel: supports owned by edge range, every rank walks the whole exchanged frontier and applies "
                            f"its own decrements, {st['shard_exchanges']} exchanges per step" if args.shard_peel else ", peel and gather replicated on every rank")) if shard else
                        (f"same graph on {world} ranks: support, index and peel on every rank (they do not shard at a profit: DESIGN.md section 6), "
                         f"the canonical results gathered in {world} slices, one per rank; no exchange on the data path (slices verified after the timed region: {slices_ok})"
                         if sliced else f"same graph on {world} ranks, replicas: every rank runs the whole single-GPU path, no exchange"))},
            "phases_ms": phase,
            "kcore": {"ms": core_ms, "edges_per_s": ne / (core_ms * 1e-3) if core_ms > 0 else None,
                      "levels": core_stats["core_levels"], "launches": core_stats["core_launches"],
                      **({"sharded": {"exchanges": core_stats["shard_exchanges"], "ms_exchange": core_stats["ms_exchange"],
                                      "exchange_words": core_stats["exchange_words"]}} if shard and args.shard_peel else {}),
                      "local": {"vertices": core_stats["core_local_units"], "index_entries": core_stats["core_local_items"],
                                "sweeps": core_stats["core_local_sweeps"], "ms": core_stats["ms_core_local"]},
                      "alg_bytes": 16 * nv + 24 * ne,
                      "GBps": (16 * nv + 24 * ne) / (core_ms * 1e-3) / 1e9 if core_ms > 0 else None},
            # every KOMB_* switch the run saw (none changes a result; some change which engine runs, e.g. KOMB_SHARD_PEEL, KOMB_ENUM)
            "env_switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("KOMB_")},
            "runtruss_faithful": faithful,
            "first_call": first_call,
            # nothing a k-truss / k-core call computes is skipped after the first call: the graph moments (sum d^2, max degree, ...)
            # that rounds 1-3 computed on the first call only are made with the graph since round 4
            "statistics_skipped_after_first_call": [],
            "c2": c2_block,
            "setup_s": {"generate": t_gen, "graph_build_incl_h2d": t_build, "device_build_ms": build_stats["ms_build"],
                        "h2d_ms": build_stats["ms_build_h2d"], "renumber_orient_lines_ms": build_stats["ms_build_relabel"],
                        "note": "first graph build of the process: includes the HIP runtime's first large allocations and first launches of "
                                "every build kernel (komb_create has loaded the code object and made the pinned staging buffers); "
                                "first_call.graph_build_ms is a second build in the same process.  The graph object holds the "
                                "(degree,id)-renumbered oriented CSR, the canonical edge map, the per-vertex lines of the enumeration and the "
                                "graph moments (all functions of the graph alone, built once with it); no k-truss / k-core call reuses "
                                "anything a previous call computed"},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed at N=1 only
            out["cpu_baseline"] = cb = cpu_baseline()
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
