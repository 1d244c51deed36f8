"""bench.py argument handling that needs no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300)


def test_conflicting_multi_gpu_modes_are_refused():
    r = _run("--gpus", "2", "--batch", "--shard")
    assert r.returncode != 0 and "--batch runs one graph per rank" in (r.stdout + r.stderr)
    r = _run("--gpus", "2", "--replicas", "--shard")
    assert r.returncode != 0 and "--replicas runs the unsharded path" in (r.stdout + r.stderr)


def test_world_size_mismatch_is_reported():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-build"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stdout + r.stderr)


def test_no_gpu_is_an_error_not_a_fallback(built):
    """Without a GPU the bench refuses to run (no CPU path behind the same metric)."""
    import torch
    if torch.cuda.is_available():
        return
    r = _run("--gpus", "1", "--no-build", "--no-cpu-baseline", "--config", "tiny")
    assert r.returncode != 0 and "needs a GPU" in (r.stdout + r.stderr)
