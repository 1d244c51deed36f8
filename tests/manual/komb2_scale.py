"""Rows N1/N2 at scale: komb2's host pipeline (SAM -> edge list, writers) against oracle/sam_port, the
reference's pipeline restated with its own hash containers, on the same synthetic SAM pair; then, when a GPU
is present, the whole komb2 run with its own stage times.

    python tests/manual/komb2_scale.py [n_unitigs] [n_reads] [threads] [workdir]

Prints the graph comparison (must be identical) and the wall-clock of both.
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KOMB2 = os.path.join(ROOT, "komb_amd", "bin", "komb2")
PORT = os.path.join(ROOT, "oracle", "sam_port")


def write_inputs(d, n_unitigs, n_reads, seed=1):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    with open(os.path.join(d, "unitigs.fa"), "w") as f:
        seq = "ACGT" * 16
        for lo in range(0, n_unitigs, 1 << 16):
            f.write("".join(f">{u} LN:i:64 KC:i:7\n{seq}\n" for u in range(lo, min(n_unitigs, lo + (1 << 16)))))
    lines = 0
    for mate in (1, 2):
        # every read aligns to 1-3 unitigs drawn from a skewed distribution; 5% unmapped; a fifth of the
        # reads carry the /1 /2 suffix (the reference cuts the key at '/')
        k = rng.choice([1, 1, 1, 2, 3], n_reads)
        rid = np.repeat(rng.permutation(n_reads), k)
        uni = np.minimum((n_unitigs * rng.random(rid.size) ** 3).astype(np.int64), n_unitigs - 1)
        star = rng.random(rid.size) < 0.05
        flag = rng.choice([0, 16, 256], rid.size)
        with open(os.path.join(d, f"r{mate}.sam"), "w") as f:
            f.write("".join(f"@SQ\tSN:{u}\tLN:64\n" for u in range(0, n_unitigs, 997)))
            f.write("@PG\tID:bwa-mem2\tPN:bwa-mem2\n")
            for lo in range(0, rid.size, 1 << 18):
                hi = min(rid.size, lo + (1 << 18))
                f.write("".join(
                    f"r{r}{'/%d' % mate if r % 5 == 0 else ''}\t{fl}\t{'*' if st else u}\t7\t60\t50M\t*\t0\t0\tACGT\tIIII\n"
                    for r, fl, st, u in zip(rid[lo:hi].tolist(), flag[lo:hi].tolist(), star[lo:hi].tolist(), uni[lo:hi].tolist())))
        lines += rid.size
    print(f"inputs: {n_unitigs} unitigs, {lines} alignment lines in two SAM files, written in {time.time() - t0:.1f} s", flush=True)
    return lines


def load_pairs_by_name(path):
    import pandas as pd
    df = pd.read_csv(path, sep="\t", header=None, dtype=np.int64).to_numpy()
    return df


def canonical(pairs):
    a, b = np.minimum(pairs[:, 0], pairs[:, 1]), np.maximum(pairs[:, 0], pairs[:, 1])
    keep = a != b
    return np.unique(a[keep] * (1 << 32) + b[keep])


def main():
    n_unitigs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 8)
    d = sys.argv[4] if len(sys.argv) > 4 else "/tmp/komb2_scale"
    os.makedirs(d, exist_ok=True)
    lines = write_inputs(d, n_unitigs, n_reads)
    args = ["-t", str(threads), "-l", "-1", "-i", f"{d}/r1.sam", "-j", f"{d}/r2.sam", "-u", f"{d}/unitigs.fa"]

    t0 = time.time()
    r = subprocess.run([PORT, str(threads), f"{d}/r1.sam", f"{d}/r2.sam", f"{d}/port_pairs.txt"], capture_output=True, text=True)
    t_port = time.time() - t0
    assert r.returncode == 0, r.stderr
    print(f"sam_port ({threads} threads): {t_port:.2f} s wall\n  " + r.stdout.strip().replace("\n", "\n  "), flush=True)

    t0 = time.time()
    r = subprocess.run([KOMB2] + args + ["-o", f"{d}/host"], capture_output=True, text=True, env=dict(os.environ, KOMB_STOP_AFTER_EDGES="1"))
    t_host = time.time() - t0
    assert r.returncode == 0, r.stderr
    stages = [ln.strip() for ln in r.stdout.splitlines() if ln.startswith("Time elapsed")]
    print(f"komb2 host pipeline only ({threads} threads): {t_host:.2f} s wall  ({lines / t_host / 1e6:.2f} M lines/s)\n  " + "\n  ".join(stages[:5]), flush=True)

    # same graph? names are integers in this fixture: compare by name
    import pandas as pd
    names = pd.read_csv(f"{d}/host/vertex_names.txt", sep="\t", header=None, dtype=np.int64).to_numpy()
    name_of = np.zeros(names[:, 0].max() + 1, np.int64)
    name_of[names[:, 0]] = names[:, 1]
    ours = canonical(name_of[load_pairs_by_name(f"{d}/host/edgelist.txt")])
    theirs = canonical(load_pairs_by_name(f"{d}/port_pairs.txt"))
    same = ours.size == theirs.size and bool(np.all(ours == theirs))
    print(f"simple edges: komb2 {ours.size}, sam_port {theirs.size}, identical: {same}", flush=True)
    assert same

    import torch
    if torch.cuda.is_available():
        t0 = time.time()
        r = subprocess.run([KOMB2] + args + ["-o", f"{d}/full"], capture_output=True, text=True)
        t_full = time.time() - t0
        assert r.returncode == 0, r.stderr
        keep = [ln.strip() for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("\t")]
        print(f"komb2 end to end on the GPU ({threads} threads): {t_full:.2f} s wall\n  " + "\n  ".join(keep), flush=True)
    else:
        print("no GPU here: end-to-end run skipped")


if __name__ == "__main__":
    main()
