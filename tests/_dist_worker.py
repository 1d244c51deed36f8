"""Worker for the world_size-2 tests (launched with torch.distributed.run).

mode "cpu": gloo on CPU tensors -- each rank counts the triangle supports of its
own source-vertex shard with a pure-Python restatement of the sharded kernel's
role counting, the partial vectors are summed with komb_amd.distributed's
all-reduce, and the sum must equal the single-rank count and the oracle; then the
sharded PEELS' protocol (komb_amd/csrc/shard_dev.h: live keys owned by range,
owner-computes decrements, the frontier concatenated every sub-round by an
all-reduce over disjoint segments) restated in Python for k-core and k-truss,
against the oracle on every rank.
mode "gpu": the real komb_truss_run_sharded on the GPU box (ranks share GPU 0,
gloo backend, all-reduce staged through the host), checked against the oracle.
mode "peel": komb_core_run_sharded and komb_truss_run_sharded + komb_set_shard_peel
on the GPU box, 2 or 3 ranks.  mode "c3": all of it at full C3 size.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import komb_amd                                    # noqa: E402
import komb_amd.api                                # noqa: E402
from komb_amd import distributed as kd             # noqa: E402
komb_amd.api.FORWARD_ENV_OPTIONS = True            # this worker's KOMB_* switches reach the library as per-context options (api.py)
from oracle import oracle as O                     # noqa: E402


def oriented(rowptr, col):
    nv = len(rowptr) - 1
    deg = np.diff(rowptr)
    rows = []
    for a in range(nv):
        nb = col[rowptr[a]:rowptr[a + 1]]
        keep = [int(b) for b in nb if (deg[a], a) < (deg[b], b)]
        rows.append(keep)
    orow = np.zeros(nv + 1, dtype=np.int64)
    orow[1:] = np.cumsum([len(r) for r in rows])
    return orow, rows


def partial_support(orow, rows, v_lo, v_hi):
    """Per-edge triangle counts contributed by source vertices [v_lo, v_hi); internal edge id = oriented slot."""
    m = int(orow[-1])
    own = np.zeros(m + 1, dtype=np.int32)
    other = np.zeros(m + 1, dtype=np.int32)
    for a in range(v_lo, v_hi):
        ra = rows[a]
        pos_a = {w: i for i, w in enumerate(ra)}
        for ib, b in enumerate(ra):
            for jw, w in enumerate(rows[b]):
                if w in pos_a:
                    own[orow[a] + ib] += 1
                    own[orow[a] + pos_a[w]] += 1
                    other[orow[b] + jw] += 1
    return own + other                                   # what the library sums before the exchange


def concat_by_allreduce(own, world, rank):
    """The exchange of komb_amd/csrc/shard_dev.h, restated on CPU tensors: first the ranks' counts (each rank fills its own
    slot of a zeroed header), then the ids (each rank fills its own segment of a zeroed buffer); a SUM all-reduce of
    disjoint segments is the concatenation.  Returns the concatenated list (rank order)."""
    hdr = torch.zeros(world, dtype=torch.int32)
    hdr[rank] = len(own)
    kd.allreduce_sum_(hdr)
    counts = hdr.tolist()
    total, off = sum(counts), sum(counts[:rank])
    buf = torch.zeros(total, dtype=torch.int32)
    if len(own):
        buf[off:off + len(own)] = torch.tensor(own, dtype=torch.int32)
    if total:
        kd.allreduce_sum_(buf)
    return buf.tolist()


def sharded_core_python(rowptr, col, rank, world):
    """k-core with the live degrees owned by vertex range: every rank walks every frontier vertex's row and applies the
    decrements on the vertices it owns (ShardCore in kcore.hip), the frontier concatenated every sub-round."""
    nv = len(rowptr) - 1
    lo, hi = nv * rank // world, nv * (rank + 1) // world
    deg = np.diff(rowptr).astype(np.int64)
    core = np.full(nv, -1, dtype=np.int64)
    core[deg == 0] = 0
    remaining = int((deg > 0).sum())
    level = 0
    while remaining:
        own = [v for v in range(lo, hi) if core[v] < 0 and deg[v] <= level]
        while True:
            front = concat_by_allreduce(own, world, rank)
            if not front:
                break
            for v in front:
                core[v] = level                               # every rank stamps the whole frontier
            own = []
            for v in front:
                for u in col[rowptr[v]:rowptr[v + 1]]:
                    if lo <= u < hi and core[u] < 0:          # the owner's decrement
                        deg[u] -= 1
                        if deg[u] == level:
                            core[u] = level; own.append(int(u))
            remaining -= len(front)
        level += 1
    return core


def sharded_truss_python(orow, rows, rank, world):
    """k-truss peel with the supports owned by edge range (ShardTruss in ktruss.hip): internal edge id = oriented slot;
    every rank walks every frontier edge's triangles, decides from the replicated stamps (an edge of the triangle gone
    in an earlier sub-round: skip; one in the same sub-round: the smaller id keeps the triangle) and applies the
    decrements on the edges it owns.  Returns (support, trussness) by internal id."""
    nv = len(rows)
    m = int(orow[-1])
    inc = [[] for _ in range(m)]
    for a in range(nv):
        pos_a = {w: i for i, w in enumerate(rows[a])}
        for ib, b in enumerate(rows[a]):
            for jw, w in enumerate(rows[b]):
                if w in pos_a:
                    e, i, j = int(orow[a]) + ib, int(orow[a]) + pos_a[w], int(orow[b]) + jw
                    inc[e].append((i, j)); inc[i].append((e, j)); inc[j].append((e, i))
    sup0 = np.array([len(x) for x in inc], dtype=np.int64)
    sup = sup0.copy()
    lo, hi = m * rank // world, m * (rank + 1) // world
    ALIVE = 1 << 40
    stamp = np.full(m, ALIVE, dtype=np.int64)
    truss = np.full(m, 2, dtype=np.int64)
    stamp[sup == 0] = 0
    remaining = int((sup > 0).sum())
    level, rnd = 1, 1
    while remaining:
        own = [e for e in range(lo, hi) if stamp[e] == ALIVE and sup[e] <= level]
        while True:
            front = concat_by_allreduce(own, world, rank)
            if not front:
                break
            for e in front:
                stamp[e] = rnd; truss[e] = level + 2
            own = []
            for e in front:
                for x, y in inc[e]:
                    if stamp[x] < rnd or stamp[y] < rnd:
                        continue
                    xin, yin = stamp[x] == rnd, stamp[y] == rnd
                    for t, tin, oin, other in ((x, xin, yin, y), (y, yin, xin, x)):
                        if not tin and (not oin or e < other) and lo <= t < hi:
                            sup[t] -= 1
                            if sup[t] == level:
                                stamp[t] = rnd + 1; truss[t] = level + 2; own.append(t)
            remaining -= len(front)
            rnd += 1
        level += 1
    return sup0, truss


def rccl_main():
    """One rank, backend "nccl" (= RCCL): the in-place device branch of the all-reduce callback, called the way the
    library calls it (raw device pointer + count).  With one rank the sum is the identity."""
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    cb = kd.make_allreduce_callback(0)
    x = torch.arange(1, 100001, dtype=torch.int32, device="cuda:0")
    want = x.clone()
    torch.cuda.synchronize()
    rc = cb(None, x.data_ptr(), x.numel())
    assert rc == 0 and torch.equal(x, want)
    # and through the library: world 1 of 1 takes the single-GPU path, results equal the oracle's
    nv = 20000
    uv = komb_amd.gen_hug_edges(nv, int(2.6 * nv), 2.6, 11)
    rowptr, col = O.simplify(nv, uv)
    with komb_amd.KombAccel(device=0) as a:
        a.from_edges(nv, uv)
        kd.truss_run_sharded(a)
        assert np.array_equal(a.truss_fetch()[2], O.trussness(rowptr, col))
    print("DIST_OK rccl 1")
    dist.destroy_process_group()


def rccl_n_main():
    """N ranks on N GPUs, backend "nccl" (= RCCL over xGMI): the flow `bench.py --gpus N` runs by default -- the support
    all-reduce in place on the device buffer + the peel sharded by edge range, one frontier exchange per sub-round through the
    same callback -- and the sharded k-core, against the oracle on every rank.  Needs one GPU per rank: the test that launches
    it skips on a one-GPU box."""
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    rank, world = dist.get_rank(), dist.get_world_size()
    assert dist.get_backend() == "nccl"
    nv = 60000
    uv = komb_amd.gen_hug_edges(nv, int(2.6 * nv), 2.6, 11)
    rowptr, col = O.simplify(nv, uv)
    otr, ocore = O.trussness(rowptr, col), O.coreness(rowptr, col)
    with komb_amd.KombAccel(device=local) as a:
        a.from_edges(nv, uv)
        for shard_peel in (False, True):
            kd.truss_run_sharded(a, shard_peel=shard_peel)
            st = a.stats()
            assert np.array_equal(a.truss_fetch()[2], otr), f"rank {rank}: trussness (shard_peel={shard_peel})"
            assert st["ms_allreduce"] > 0 and (st["shard_exchanges"] > 0) == shard_peel
        kd.core_run_sharded(a)
        assert np.array_equal(a.core_fetch()[1], ocore), f"rank {rank}: coreness"
    dist.barrier()
    if rank == 0:
        print("DIST_OK rccl_n", world)
    dist.destroy_process_group()


def c3_main():
    """BASELINE configs[3]'s code path at the size it is quoted on: the C3 graph (|V|=10M, |E|=100.1M) on two ranks
    sharing GPU 0, komb_truss_run_sharded (support counted per shard, all-reduced over gloo), against the recorded
    known answer of the single-GPU path and -- on rank 0 -- against a single-rank run of the same build, value for value."""
    import hashlib
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nv = 10_000_000
    uv = komb_amd.gen_hug_edges(nv, 24_250_000, 2.6, 42)
    with komb_amd.KombAccel(device=0) as a:
        a.from_edges(nv, uv)
        del uv
        assert a.ne == 100_120_558
        kd.truss_run_sharded(a)
        st = a.stats()
        assert st["ms_allreduce"] > 0 and st["triangles"] == 88_336_441
        eu, ev, tr, sup = a.truss_fetch(with_support=True)
        assert int(sup.sum(dtype=np.int64)) == 3 * 88_336_441
        assert hashlib.sha256(tr.tobytes()).hexdigest()[:16] == "5970a467914854ea", "sharded C3 trussness differs from the known answer"
        # SURVEY 8(e)'s partition at the same size: supports owned by edge range, the frontier exchanged every sub-round
        # (komb_set_shard_peel), and the k-core with live degrees owned by vertex range -- every value, on every rank
        kd.truss_run_sharded(a, shard_peel=True)
        st = a.stats()
        assert st["shard_exchanges"] > 0 and st["exchange_words"] > a.ne // 2, st["shard_exchanges"]
        eu2, ev2, tr2, sup2 = a.truss_fetch(with_support=True)
        assert np.array_equal(eu, eu2) and np.array_equal(ev, ev2) and np.array_equal(tr, tr2) and np.array_equal(sup, sup2), "sharded-peel C3 trussness differs"
        a.set_shard_peel(False)
        kd.core_run_sharded(a)
        assert a.stats()["shard_exchanges"] > 0
        deg_s, core_s = a.core_fetch()
        assert hashlib.sha256(core_s.tobytes()).hexdigest()[:16] == "120d47bf172d8b8f", "sharded C3 coreness differs from the known answer"
        dist.barrier()
        if rank == 0:
            a.truss_run()
            eu1, ev1, tr1, sup1 = a.truss_fetch(with_support=True)
            assert np.array_equal(eu, eu1) and np.array_equal(ev, ev1) and np.array_equal(tr, tr1) and np.array_equal(sup, sup1)
            deg1, core1 = a.run_core()
            assert np.array_equal(core1, core_s) and np.array_equal(deg1, deg_s)
    dist.barrier()
    if rank == 0:
        print("DIST_OK c3", world)
    dist.destroy_process_group()


def _peel_graphs():
    """(name, nv, raw pairs): shapes whose frontiers hold light and heavy units, long cascades and empty levels."""
    rng = np.random.default_rng(5)
    out = []
    nv = 60000
    out.append(("hug60k", nv, np.asarray(komb_amd.gen_hug_edges(nv, int(2.6 * nv), 2.6, 11)).reshape(-1, 2)))
    nv = 20000
    out.append(("hug20k_a2.1", nv, np.asarray(komb_amd.gen_hug_edges(nv, int(2.7 * nv), 2.1, 3)).reshape(-1, 2)))
    # a clique on 120 vertices (every edge has 118 triangles: heavy units; one populated level far above the others),
    # a star over everything (a hub row of nv - 1 slots) and a long path (a cascade: one unit per sub-round)
    nv = 5000
    iu = np.triu_indices(120, 1)
    clique = np.stack(iu, axis=1) + 100
    star = np.stack([np.zeros(nv - 1, np.int64), np.arange(1, nv)], axis=1)
    path = np.stack([np.arange(2000, 4999), np.arange(2001, 5000)], axis=1)
    out.append(("clique+star+path", nv, np.concatenate([clique, star, path, rng.integers(0, nv, (3000, 2))])))
    out.append(("empty", 10, np.zeros((0, 2), np.int64)))
    out.append(("triangle-free", 1000, np.stack([np.arange(0, 999), np.arange(1, 1000)], axis=1)))
    return out


def peel_main():
    """SURVEY 8(e)'s sharded peel (komb_amd/csrc/shard_dev.h) on the GPU box: the ranks (sharing GPU 0, gloo) own vertex /
    edge ranges and exchange their parts of the frontier every sub-round; coreness, supports and trussness must equal the
    oracle's on every rank."""
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    for name, nv, uv in _peel_graphs():
        uv = np.ascontiguousarray(uv, dtype=np.int64)
        rowptr, col = O.simplify(nv, uv)
        with komb_amd.KombAccel(device=0) as a:
            a.from_edges(nv, uv)
            ocore, (osup, otri), otr = O.coreness(rowptr, col), O.support(rowptr, col), O.trussness(rowptr, col)
            # "none": sharded sub-rounds to the end; "local": the remainder handed to the replicated local finish after one
            # exchange of the live keys (a limit of 500 units: offered late, and on a dense remainder declined)
            for finish, limit in (("none", None), ("local", None), ("local", "500")):
                os.environ["KOMB_FINISH"] = finish
                if limit: os.environ["KOMB_LOCAL_LIMIT"] = limit
                else: os.environ.pop("KOMB_LOCAL_LIMIT", None)
                kd.core_run_sharded(a)
                deg, core = a.core_fetch()
                assert np.array_equal(core, ocore), f"{name}/{finish}: sharded coreness mismatch on rank {rank}"
                st = a.stats()
                if a.ne and finish == "none":            # (with the local finish a small graph is handed over whole, before any exchange)
                    assert st["shard_exchanges"] > 0 and st["exchange_words"] >= 2 * world, (name, st["shard_exchanges"])
                kd.truss_run_sharded(a, shard_peel=True)
                st = a.stats()
                tr_levels = st["truss_levels"]
                eu, ev, tr, sup = a.truss_fetch(with_support=True)
                assert np.array_equal(sup, osup), f"{name}/{finish}: support mismatch on rank {rank}"
                assert np.array_equal(tr, otr), f"{name}/{finish}: sharded-peel trussness mismatch on rank {rank}"
                if otri and finish == "none":
                    assert st["shard_exchanges"] > 0 and st["ms_exchange"] > 0
                if finish == "none":
                    assert st["truss_local_units"] == 0
            os.environ.pop("KOMB_FINISH", None); os.environ.pop("KOMB_LOCAL_LIMIT", None)
            # the replicated peel of the same context, after the sharded one
            kd.truss_run_sharded(a, shard_peel=False)
            st2 = a.stats()
            assert st2["shard_exchanges"] == 0
            assert np.array_equal(a.truss_fetch()[2], tr)
            assert st2["max_trussness"] == st["max_trussness"] and (tr_levels >= 1 or a.ne == 0), (name, st2["truss_levels"], tr_levels)
            if name == "hug60k":
                mask = (O.coreness(rowptr, col) >= 5).astype(np.uint8)
                kd.truss_run_sharded(a, mask, shard_peel=True)
                seu, sev, stra = a.truss_fetch()
                weu, wev, wtr = O.trussness_induced(rowptr, col, mask)
                assert np.array_equal(seu, weu) and np.array_equal(sev, wev) and np.array_equal(stra, wtr)
            if name == "hug60k" and world > 1:
                # a failure on ONE rank between two collectives must not strand the others.  Rank 1's callback carries
                # out the third data exchange of a sharded k-core and then reports failure: every rank has to come back with an
                # error from the same iteration (the status word of the next header, shard_dev.h) -- nobody hangs -- and the
                # context must work again afterwards.
                import ctypes
                good = kd.make_allreduce_callback(a.device, None)
                calls = {"data": 0}

                def _cb(user, dev_ptr, count):
                    rc = good(user, dev_ptr, count)
                    if count != 3 * world:
                        calls["data"] += 1
                        if rank == 1 and calls["data"] == 3:
                            return 1
                    return rc
                bad = komb_amd._lib.ALLREDUCE_FN(_cb)
                os.environ["KOMB_FINISH"] = "none"
                a._sync_env_options()
                rc = a._lib.komb_core_run_sharded(a._ctx, rank, world, ctypes.cast(bad, ctypes.c_void_p), None)
                os.environ.pop("KOMB_FINISH", None)
                assert rc == komb_amd._lib.KOMB_ERR_DEVICE, f"rank {rank}: the injected failure was not reported (rc {rc})"
                msg = a._lib.komb_last_error(a._ctx).decode()
                assert ("callback failed" in msg) if rank == 1 else ("rank 1 reported a failure" in msg), (rank, msg)
                dist.barrier()
                kd.core_run_sharded(a)
                assert np.array_equal(a.core_fetch()[1], ocore)
        dist.barrier()
    if rank == 0:
        print("DIST_OK peel", world)
    dist.destroy_process_group()


def main():
    mode = sys.argv[1]
    if mode == "rccl":
        return rccl_main()
    if mode == "rccl_n":
        return rccl_n_main()
    if mode == "c3":
        return c3_main()
    if mode == "peel":
        return peel_main()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nv = 3000 if mode == "cpu" else 60000
    uv = komb_amd.gen_hug_edges(nv, int(2.6 * nv), 2.6, 11)
    rowptr, col = O.simplify(nv, uv)
    if mode == "cpu":
        orow, rows = oriented(rowptr, col)
        n_tasks = kd.n_support_tasks(nv)
        lo, hi = kd.shard_range(n_tasks, rank, world)
        part = partial_support(orow, rows, min(nv, lo * 16), min(nv, hi * 16))
        t = torch.from_numpy(part.copy())
        kd.allreduce_sum_(t)
        full = partial_support(orow, rows, 0, nv)
        assert np.array_equal(t.numpy(), full), "sum of shards != single-rank count"
        m = int(orow[-1])
        sup_internal = full
        osup, otri = O.support(rowptr, col)
        assert int(sup_internal.sum()) == 3 * otri
        assert sorted(sup_internal[:m].tolist()) == sorted(osup.tolist())
        # shards tile the task range exactly
        spans = [kd.shard_range(n_tasks, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n_tasks and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        # the sharded PEELS' protocol (shard_dev.h), restated: owner-computes decrements + the frontier concatenated through
        # the all-reduce helper every sub-round; coreness and trussness must equal the oracle's on every rank
        assert np.array_equal(sharded_core_python(rowptr, col, rank, world), O.coreness(rowptr, col)), "sharded k-core protocol"
        psup, ptr_ = sharded_truss_python(orow, rows, rank, world)
        # internal ids -> canonical order: edge (a, b) with a < b, sorted by (a, b)
        ends = [(min(a, b), max(a, b)) for a in range(nv) for b in rows[a]]
        order = sorted(range(m), key=lambda e: ends[e])
        assert np.array_equal(psup[order], osup), "sharded k-truss protocol: supports"
        assert np.array_equal(ptr_[order], O.trussness(rowptr, col)), "sharded k-truss protocol: trussness"
    else:
        with komb_amd.KombAccel(device=0) as a:
            a.from_edges(nv, uv)
            kd.truss_run_sharded(a)
            eu, ev, tr, sup = a.truss_fetch(with_support=True)
            osup, _ = O.support(rowptr, col)
            assert np.array_equal(sup, osup), "sharded support mismatch"
            assert np.array_equal(tr, O.trussness(rowptr, col)), "sharded trussness mismatch"
            mask = (O.coreness(rowptr, col) >= 5).astype(np.uint8)
            kd.truss_run_sharded(a, mask)
            seu, sev, stra = a.truss_fetch()
            weu, wev, wtr = O.trussness_induced(rowptr, col, mask)
            assert np.array_equal(seu, weu) and np.array_equal(sev, wev) and np.array_equal(stra, wtr)
            st = a.stats()
            assert st["ms_allreduce"] > 0
    dist.barrier()
    if rank == 0:
        print("DIST_OK", mode, world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
