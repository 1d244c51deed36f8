"""komb2 host logic on CPU: CLI contract (src/komb2.cpp:35-73) and SAM -> graph
construction (src/graph.cpp:166-393), checked against tests/samgraph.py with the
device stage switched off (KOMB_STOP_AFTER_EDGES=1)."""
import os
import subprocess

import pytest

import samgraph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KOMB2 = os.path.join(ROOT, "komb_amd", "bin", "komb2")


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
