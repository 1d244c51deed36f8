"""N>1 path: world_size-2 runs (gloo).  CPU: shard arithmetic + all-reduce of the
support vectors against a Python restatement of the sharded role counting, and the sharded
peels' exchange protocol (owner-computes decrements, frontier concatenated by all-reduce)
restated in Python against the oracle.
GPU box: the real komb_truss_run_sharded with two ranks sharing GPU 0."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(mode, nproc, port, threads="2"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=threads)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py"), mode]
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)


def test_shard_ranges(built):
    from komb_amd import distributed as kd
    for n in (0, 1, 7, 625000):
        for world in (1, 2, 3, 8):
            spans = [kd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_world2_gloo_cpu(built):
    r = _launch("cpu", 2, 29611)
    assert r.returncode == 0 and "DIST_OK cpu 2" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_world2_sharded_on_gpu(built):
    r = _launch("gpu", 2, 29612)
    assert r.returncode == 0 and "DIST_OK gpu 2" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_world2_sharded_c3_size(built):
    """C4's code path at the size it is quoted on: two ranks (sharing GPU 0), the full C3 graph."""
    r = _launch("c3", 2, 29614, threads="8")
    assert r.returncode == 0 and "DIST_OK c3 2" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_peel_on_gpu(built, world):
    """SURVEY 8(e)'s partition: komb_core_run_sharded and komb_truss_run_sharded with komb_set_shard_peel -- vertex / edge
    ranges owned by the ranks, the frontier exchanged every sub-round -- against the oracle, 2 and 3 ranks (uneven ranges)."""
    r = _launch("peel", world, 29615 + world)
    assert r.returncode == 0 and f"DIST_OK peel {world}" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_rccl_inplace_branch_single_rank(built):
    """backend "nccl" on the GPU box: the in-place RCCL branch of komb_amd.distributed's callback runs (one rank)."""
    r = _launch("rccl", 1, 29613)
    assert r.returncode == 0 and "DIST_OK rccl 1" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_rccl_two_ranks_on_two_gpus(built):
    """The default multi-GPU flow (support all-reduce + edge-range sharded peel) and the sharded k-core over RCCL with one GPU
    per rank, against the oracle on both ranks.  Skips where there is one GPU (this pool's boxes): the one-GPU rehearsals
    above run the same library code over gloo."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    r = _launch("rccl_n", 2, 29616)
    assert r.returncode == 0 and "DIST_OK rccl_n 2" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_processes_sharing_one_gpu(built):
    """Two processes decomposing at the same time on GPU 0 (scripts/share_stress.py): workgroups of a launch get
    dispatched late when the CUs are busy with somebody else's kernels -- the peel's launch-to-launch hand-over must not
    depend on when they start (this caught stale-state steps before every launch carried its index)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "share_stress.py"), "2", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rc [0, 0]" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--c4-allreduce"], ["--same-graph", "--shard"], ["--replicas-sliced"], ["--replicas"], ["--batch"], ["--shard-peel"]])
def test_bench_two_ranks_on_one_gpu(built, extra):
    """`python bench.py --gpus 2` started plainly: it spawns its two ranks itself (before anything has touched the GPU) and
    rank 0 prints the one JSON line.  Both ranks share GPU 0 here (KOMB_BENCH_ONE_DEVICE=1, exchange over gloo).  Default
    (= --shard-peel): BASELINE configs[3] / north_star's partition -- the same graph on both ranks, the support count sharded +
    all-reduce, the peel sharded by edge range with one frontier exchange per sub-round, the other flows timed beside it;
    --c4-allreduce (--same-graph --shard is the older spelling): the support all-reduce only; --replicas-sliced: round 4's
    default (each rank's slice of the results, verified against a whole run after the timed region, no exchange); --replicas:
    nothing sliced; --batch: one graph per rank, weak scaling, own metric name."""
    import json
    env = dict(os.environ, KOMB_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-build"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["value_resident"] >= d["value"] * 0.8 and d["phases_ms"]["ms_prepare"] > 0
    par = d["config"]["parallelism"]
    if extra == ["--batch"]:
        assert d["scaling"] == "weak" and "independent graphs" in par and d["metric"].startswith("aggregate peeled edges/sec")
    else:
        assert d["scaling"] == "strong" and d["metric"] == "peeled edges/sec (k-truss)"
        if extra == ["--replicas-sliced"]:
            assert "slices verified after the timed region: True" in par and d["phases_ms"]["ms_allreduce"] == 0
        elif extra == ["--replicas"]:
            assert "replicas" in par
        else:
            assert "sharded" in par and d["config"]["workload"].startswith("C4") and d["phases_ms"]["ms_allreduce"] > 0
        if extra in ([], ["--shard-peel"]):              # the peel by edge range: exchanges counted, k-core sharded as well, the other flows beside
            assert d["config"]["shard_peel"]["exchanges"] > 0 and "edge range" in d["config"]["workload"] and "sharded" in d["kcore"]
            assert set(d["alternatives"]) == {"support_allreduce_only", "replicas_sliced"} and d["config"]["engine_flags"] & 4
        else:
            assert "shard_peel" not in d["config"] and d["phases_ms"]["ms_exchange"] == 0
    assert "cpu_baseline" not in d                       # timed at N = 1 only
