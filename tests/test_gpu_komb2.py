"""BASELINE config C1 (plumbing) on the GPU box: the drop-in komb2 binary run the
way KOMB.py runs it (KOMB.py:435-442), on the generated SAM + FASTA fixture,
with every output file checked against the oracle (keyed by unitig Name, since
the reference's vertex numbering is hash-order dependent)."""
import os
import subprocess

import numpy as np
import pytest

import samgraph

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KOMB2 = os.path.join(ROOT, "komb_amd", "bin", "komb2")


@pytest.fixture(scope="module")
def fixture(built, tmp_path_factory):
    d = tmp_path_factory.mktemp("c1gpu")
    fasta, s1, s2 = samgraph.make_fixture(2000, 20000, seed=1)
    (d / "unitigs.l-1.fasta").write_bytes(fasta)
    (d / "reads1.fastq.sam").write_bytes(s1)
    (d / "reads2.fastq.sam").write_bytes(s2)
    return d, fasta, s1, s2


def _fasta_map(fasta: bytes):
    m, cur = {}, None
    for ln in fasta.decode().split("\n"):
        if ln.startswith(">"):
            cur = ln[1:].split(" ")[0]
            m[cur] = ""
        elif cur is not None:
            m[cur] += ln
    return m


@pytest.mark.parametrize("threads", [1, 4])
def test_komb2_end_to_end(fixture, tmp_path, threads):
    from oracle import oracle as O
    d, fasta, s1, s2 = fixture
    out = tmp_path / "out"
    out.mkdir()                                              # KOMB.py creates it (KOMB.py:41-52)
    cmd = f"{KOMB2} -t {threads} -l -1 -o {out} -i {d}/reads1.fastq.sam -j {d}/reads2.fastq.sam -u {d}/unitigs.l-1.fasta"
    env = dict(os.environ, KOMB_TRUSS="1")
    r = subprocess.run(cmd, shell=True, executable="/bin/bash", capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    for f in ("kcore.tsv", "CoreA_anomaly.txt", "edgelist.txt", "truss_unitigs.fasta"):   # .github/workflows/build-run.yml:80
        assert (out / f).exists(), f

    # expected graph, by name
    want_names, want_edges = samgraph.build_graph(s1, s2, threads)
    rows = [ln.rstrip("\n").split("\t") for ln in open(out / "kcore.tsv")]
    assert rows[0] == ["#VID", "Name", "Coreness", "Degree"]
    rows = rows[1:]
    assert [int(x[0]) for x in rows] == list(range(len(rows)))
    names = [x[1] for x in rows]
    assert set(names) == want_names and len(names) == len(want_names)
    vid = {nm: i for i, nm in enumerate(names)}
    uv = np.array([[vid[a], vid[b]] for a, b in (tuple(e) for e in want_edges)], dtype=np.int64).reshape(-1, 2)
    rowptr, col = O.simplify(len(names), uv)
    deg, core = O.degree(rowptr), O.coreness(rowptr, col)
    assert [int(x[3]) for x in rows] == deg.tolist()
    assert [int(x[2]) for x in rows] == core.tolist()

    # stdout contract
    assert f"\tNumber of vertices: {len(names)}\n" in r.stdout
    assert f"\tNumber of edges: {len(want_edges)}\n" in r.stdout
    for line in ("Time elapsed doing K-core decomposition", "Created Kcore", "Time elapsed for combineFile",
                 "Time elapsed for anomalyDetection", "Identified anomalous unitigs", "Created anomalouss unitigs file",
                 "Time elapsed for KOMB", "Time elapsed for analysis (sec) ="):
        assert line in r.stdout, line
    score = O.corea_scores(deg, core, faithful=True)
    assert "Dense Ratio: %f\n" % float(int(core.max()) // 2) in r.stdout
    assert "Max CoreA score: %f\n" % score.max() in r.stdout

    # CoreA_anomaly.txt byte for byte (src/CombineCoreA.h:36-39)
    want = "".join("%d\t%f\n" % (i, s) for i, s in enumerate(score))
    assert open(out / "CoreA_anomaly.txt").read() == want

    # runTruss: vertices of the max-trussness edges of the max-core subgraph (src/graph.cpp:519-556)
    mask = (core == core.max()).astype(np.uint8)
    eu, ev, tr = O.trussness_induced(rowptr, col, mask)
    top = tr == tr.max()
    nodes = sorted(set(eu[top].tolist()) | set(ev[top].tolist()))
    seqs = _fasta_map(fasta)
    got = open(out / "truss_unitigs.fasta").read().split("\n")
    recs = {got[i][1:]: got[i + 1] for i in range(0, len(got) - 1, 2)}
    assert sorted(recs) == sorted("Unitig_" + names[v] for v in nodes)
    for v in nodes:
        assert recs["Unitig_" + names[v]] == seqs[names[v]]
    assert f"Found {len(nodes)} unitigs in {int(tr.max()) + 1}-truss" in r.stdout       # the +1 of src/graph.cpp:557
    assert f"Succesfully created a {int(core.max())}-core subgraph, with {len(eu)} edges." in r.stdout

    # CoreA resumed from kcore.tsv (src/CombineCoreA.h:20-21 re-reads the file): same bytes
    os.rename(out / "CoreA_anomaly.txt", out / "CoreA_first.txt")
    r2 = subprocess.run(cmd, shell=True, executable="/bin/bash", capture_output=True, text=True,
                        env=dict(os.environ, KOMB_COREA_ONLY="1"))
    assert r2.returncode == 0, r2.stderr
    assert open(out / "CoreA_anomaly.txt").read() == want


def test_komb2_missing_input_is_fatal(tmp_path):
    r = subprocess.run([KOMB2, "-t", "1", "-o", str(tmp_path), "-i", "/nonexistent.sam", "-j", "x", "-u", "y"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "could not be opened" in r.stderr


@pytest.mark.parametrize("mode", ["1", "fixed"])
def test_komb2_v1_outputs_in_the_full_run(fixture, tmp_path, mode):
    """KOMB_V1_OUTPUTS in the complete pipeline (combineFile after the k-core, splitAnomalousUnitigs after CoreA,
    src/komb2.cpp:126,139): the three files follow from the kcore.tsv / CoreA_anomaly.txt of the same run."""
    d, fasta, s1, s2 = fixture
    out = tmp_path / "out"
    out.mkdir()
    cmd = f"{KOMB2} -t 4 -l -1 -o {out} -i {d}/reads1.fastq.sam -j {d}/reads2.fastq.sam -u {d}/unitigs.l-1.fasta"
    r = subprocess.run(cmd, shell=True, executable="/bin/bash", capture_output=True, text=True,
                       env=dict(os.environ, KOMB_V1_OUTPUTS=mode))
    assert r.returncode == 0, r.stderr
    assert r.stdout.index("Time elapsed for combineFile") < r.stdout.index("Identified anomalous unitigs") < r.stdout.index("Created anomalouss")
    unitigs = samgraph.read_unitigs(fasta)
    kcore = (out / "kcore.tsv").read_text()
    corea = (out / "CoreA_anomaly.txt").read_text()
    assert (out / "combined.fasta").read_text() == samgraph.combined_fasta(kcore, unitigs)
    names = [ln.split("\t")[1] for ln in kcore.splitlines()[1:]]
    top, low, _ = samgraph.split_anomalous(corea, unitigs, names if mode == "fixed" else None)
    assert (out / "top_scoring_anomalous_unitigs.txt").read_text() == top
    assert (out / "low_scoring_anomalous_unitigs.txt").read_text() == low
    assert top and low
