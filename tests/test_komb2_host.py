"""komb2 host logic on CPU: CLI contract (src/komb2.cpp:35-73) and SAM -> graph
construction (src/graph.cpp:166-393), checked against tests/samgraph.py with the
device stage switched off (KOMB_STOP_AFTER_EDGES=1)."""
import os
import subprocess

import pytest

import samgraph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KOMB2 = os.environ.get("KOMB2_BIN", os.path.join(ROOT, "komb_amd", "bin", "komb2"))   # `make -C komb_amd/csrc asan` points this at the sanitizer build


@pytest.fixture(scope="module")
def fixture(built, tmp_path_factory):
    d = tmp_path_factory.mktemp("c1")
    fasta, s1, s2 = samgraph.make_fixture(300, 3000, seed=1)
    (d / "unitigs.fa").write_bytes(fasta)
    (d / "r1.sam").write_bytes(s1)
    (d / "r2.sam").write_bytes(s2)
    return d, s1, s2


def run(args, env=None, **kw):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([KOMB2] + args, capture_output=True, text=True, env=e, **kw)


def test_cli_contract(built):
    r = run(["--version"])
    assert r.returncode == 0 and "2.0" in r.stdout
    r = run(["--help"])
    assert r.returncode == 0 and "--input-unitigs" in r.stdout
    r = run(["-i", "a.sam"])
    assert r.returncode == 1 and r.stderr.startswith("PARSE ERROR:")
    r = run(["-i", "a", "-j", "b", "-u", "c", "--bogus"])
    assert r.returncode == 1 and "PARSE ERROR" in r.stderr
    r = run(["-i", "a", "-j", "b", "-u", "c", "-t", "x"])
    assert r.returncode == 1 and "PARSE ERROR" in r.stderr
    # KOMB.py passes "-l -1" (KOMB.py:436-442): must parse as a value; then the missing file is reported
    r = run(["-t", "2", "-l", "-1", "-o", "/tmp/komb2_cli_test", "-i", "/nonexistent/a.sam", "-j", "b", "-u", "c"])
    assert r.returncode == 1 and "File /nonexistent/a.sam could not be opened. Exiting..." in r.stderr


def _graph_from_run(outdir):
    names = {}
    for ln in open(os.path.join(outdir, "vertex_names.txt")):
        v, nm = ln.rstrip("\n").split("\t")
        names[int(v)] = nm
    edges = set()
    for ln in open(os.path.join(outdir, "edgelist.txt")):
        a, b = ln.split()
        if a != b:
            edges.add(frozenset((names[int(a)], names[int(b)])))
    return set(names.values()), edges


@pytest.mark.parametrize("threads,strict", [(1, False), (4, False), (7, False), (3, True)])
def test_sam_to_graph(fixture, tmp_path, threads, strict):
    d, s1, s2 = fixture
    out = tmp_path / "out"
    env = {"KOMB_STOP_AFTER_EDGES": "1"}
    if strict:
        env["KOMB_STRICT_SAM"] = "1"
    r = run(["-t", str(threads), "-l", "-1", "-o", str(out), "-i", str(d / "r1.sam"), "-j", str(d / "r2.sam"),
             "-u", str(d / "unitigs.fa")], env=env)
    assert r.returncode == 0, r.stderr
    for line in ("Time elapsed for reading SAMs", "Time elapsed for edgeInfo", "Time elapsed for generateGraph"):
        assert line in r.stdout
    names, edges = _graph_from_run(str(out))
    want_names, want_edges = samgraph.build_graph(s1, s2, threads, strict)
    assert names == want_names
    assert edges == want_edges


def test_thread_count_changes_the_graph_like_the_reference(fixture):
    """the reference loses the line straddling each byte-chunk boundary."""
    _, s1, s2 = fixture
    strict = samgraph.parsed_lines(s1, 1, True)
    t1 = samgraph.parsed_lines(s1, 1, False)
    t4 = samgraph.parsed_lines(s1, 4, False)
    assert len(t1) == len(strict)                  # file ends with '\n': nothing lost at T=1
    assert len(strict) - 3 <= len(t4) < len(strict)
    assert set(t4) <= set(strict)


@pytest.mark.parametrize("case", ["no_trailing_nl", "empty_second", "header_only", "short_lines"])
def test_sam_edge_cases(fixture, tmp_path, case):
    """Unterminated last line (the reference never parses it), empty / header-only files, lines with fewer
    than three fields: same graph as the restatement for several thread counts, strict and faithful."""
    d, s1, s2 = fixture
    a, b = {"no_trailing_nl": (s1.rstrip(b"\n"), s2), "empty_second": (s1, b""),
            "header_only": (b"@HD\tVN:1.0\n", s2), "short_lines": (b"a\nb\tc\n\n\n" + s1, s2)}[case]
    (tmp_path / "a.sam").write_bytes(a)
    (tmp_path / "b.sam").write_bytes(b)
    for threads in (1, 3, 16):
        for strict in (False, True):
            env = {"KOMB_STOP_AFTER_EDGES": "1"}
            if strict:
                env["KOMB_STRICT_SAM"] = "1"
            out = tmp_path / f"o{threads}{int(strict)}"
            r = run(["-t", str(threads), "-o", str(out), "-i", str(tmp_path / "a.sam"), "-j", str(tmp_path / "b.sam"),
                     "-u", str(d / "unitigs.fa")], env=env)
            assert r.returncode == 0, r.stderr
            names, edges = _graph_from_run(str(out))
            want_names, want_edges = samgraph.build_graph(a, b, threads, strict)
            assert names == want_names and edges == want_edges, (case, threads, strict)


@pytest.mark.parametrize("n,mode", [(6, "1"), (7, "1"), (401, "1"), (400, "fixed"), (37, "fixed")])
def test_v1_outputs(built, tmp_path, n, mode):
    """combineFile + splitAnomalousUnitigs (src/graph.cpp:591-635,667-749; off at src/komb2.cpp:126,139),
    from an existing kcore.tsv / CoreA_anomaly.txt, against the restatement in samgraph.py."""
    import random
    rnd = random.Random(n)
    # unitig names: a permutation of 0..n-1 minus a few (so index-named and name-keyed lookups differ and some miss)
    ids = list(range(n))
    rnd.shuffle(ids)
    fasta = "".join(f">{u} LN:i:60 KC:i:9\n{''.join(rnd.choice('ACGT') for _ in range(70))}\n"
                    f"{''.join(rnd.choice('ACGT') for _ in range(11))}\n" for u in ids if u % 9 != 4)
    names = [str(u) for u in ids]
    kcore = "#VID\tName\tCoreness\tDegree\n" + "".join(
        f"{i}\t{names[i]}\t{rnd.randrange(1, 9)}\t{rnd.randrange(1, 40)}\n" for i in range(n))
    # heavy-tailed scores with ties, six decimals like "%f" prints them
    corea = "".join(f"{i}\t{(rnd.random() ** 4) * rnd.choice([1, 1, 1, 6]):.6f}\n" for i in range(n))
    (tmp_path / "u.fa").write_text(fasta)
    (tmp_path / "kcore.tsv").write_text(kcore)
    (tmp_path / "CoreA_anomaly.txt").write_text(corea)
    r = run(["-t", "3", "-o", str(tmp_path), "-i", "x", "-j", "y", "-u", str(tmp_path / "u.fa")],
            env={"KOMB_V1_ONLY": "1", "KOMB_V1_OUTPUTS": mode})
    assert r.returncode == 0, r.stderr
    unitigs = samgraph.read_unitigs(fasta.encode())
    assert (tmp_path / "combined.fasta").read_text() == samgraph.combined_fasta(kcore, unitigs)
    top, low, _ = samgraph.split_anomalous(corea, unitigs, names if mode == "fixed" else None)
    assert (tmp_path / "top_scoring_anomalous_unitigs.txt").read_text() == top
    assert (tmp_path / "low_scoring_anomalous_unitigs.txt").read_text() == low
    assert low and (top or n < 100)


def test_v1_split_known_answer_and_small_input(built, tmp_path):
    """Hand-worked case: 8 scores 0..7 -> q1 window [0,3) size 2 -> (s[0]+s[1])/2 = 0.5; q3 window
    start 4, size 2 -> (s[4]+s[5])/2 = 4.5; cutoff 10.5 -> nothing on top. With the last score 20 the sorted
    row 7 (and only it) goes on top and is labelled Unitig_7 although the outlier is vertex 2."""
    (tmp_path / "u.fa").write_text("".join(f">{i} x\nACGT\n" for i in range(8)))
    (tmp_path / "kcore.tsv").write_text("#VID\tName\tCoreness\tDegree\n" + "".join(f"{i}\t{i}\t1\t1\n" for i in range(8)))
    vals = [0, 1, 20, 3, 4, 5, 6, 2]
    (tmp_path / "CoreA_anomaly.txt").write_text("".join(f"{i}\t{float(v):.6f}\n" for i, v in enumerate(vals)))
    args = ["-o", str(tmp_path), "-i", "x", "-j", "y", "-u", str(tmp_path / "u.fa")]
    assert run(args, env={"KOMB_V1_ONLY": "1"}).returncode == 0
    assert (tmp_path / "top_scoring_anomalous_unitigs.txt").read_text() == "Unitig_7\nACGT\n"
    assert run(args, env={"KOMB_V1_ONLY": "1", "KOMB_V1_OUTPUTS": "fixed"}).returncode == 0
    assert (tmp_path / "top_scoring_anomalous_unitigs.txt").read_text() == "Unitig_2\nACGT\n"
    # fewer than 6 scores: the reference reads outside its vector; refused with a note, combined.fasta still written
    (tmp_path / "CoreA_anomaly.txt").write_text("0\t0.100000\n1\t0.200000\n2\t0.300000\n")
    r = run(args, env={"KOMB_V1_ONLY": "1"})
    assert r.returncode == 0 and "split skipped" in r.stderr


@pytest.mark.parametrize("threads", [1, 4, 7])
def test_sam_port_agrees(fixture, tmp_path, threads):
    """oracle/sam_port (the reference's pipeline restated with its own hash containers, used as the CPU
    baseline of the host pipeline at scale) gives the graph of the Python restatement and of komb2."""
    port = os.path.join(ROOT, "oracle", "sam_port")
    d, s1, s2 = fixture
    r = subprocess.run([port, str(threads), str(d / "r1.sam"), str(d / "r2.sam"), str(tmp_path / "pairs.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    names, edges = set(), set()
    for ln in open(tmp_path / "pairs.txt"):
        a, b = ln.rstrip("\n").split("\t")
        names.update((a, b))
        if a != b:
            edges.add(frozenset((a, b)))
    want_names, want_edges = samgraph.build_graph(s1, s2, threads, False)
    assert edges == want_edges
    assert names <= want_names               # the pair list does not show vertices whose cliques are singletons
    assert f"vertices {len(want_names)} " in r.stdout
