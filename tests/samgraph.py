"""Python restatement of how KOMB turns two SAM files into a graph, used to
check the komb2 host code (test infrastructure, small inputs only).

Follows the reference: readSAM src/graph.cpp:166-257 (which lines are parsed,
read key, '*' skipped), getEdgeInfo :259-285 (mates merged per read key),
generateGraph :310-352 (every clique expands to all pairs).  Vertex numbering
in the reference depends on hash iteration order, so everything
here is keyed by unitig NAME.
"""
import random


def parsed_lines(data: bytes, threads: int, strict: bool):
    """Lines the reference parses.  Non-strict reproduces the OpenMP byte-chunk
    rule (src/graph.cpp:195-238): static chunks, first n%T threads get one more
    byte; a thread parses only lines between two newlines of its own chunk
    (thread 0 has a synthetic newline before byte 0)."""
    n = len(data)
    out = []
    if strict:
        return [ln for ln in data.split(b"\n") if ln]
    q, r = divmod(n, threads)
    lo = 0
    for t in range(threads):
        hi = lo + q + (1 if t < r else 0)
        prev = 0 if t == 0 else None
        for i in range(lo, hi):
            if data[i] in (10, 0):
                if prev is not None:
                    start = prev + 1
                    if start == 1:
                        start = 0
                    out.append(data[start:i])
                prev = i
        lo = hi
    return out


def read_sam(data: bytes, threads: int, strict: bool):
    umap = {}
    for ln in parsed_lines(data, threads, strict):
        if not ln or ln[:1] == b"@":
            continue
        tok = [t for t in ln.split(b"\t") if t]           # strtok skips empty fields
        if len(tok) < 3:
            continue
        read, unitig = tok[0].decode(), tok[2].decode()
        if unitig == "*":
            continue
        slash = read.find("/")
        key = read[1:] if slash < 0 else read[1:1 + slash]
        umap.setdefault(key, set()).add(unitig)
    return umap


def build_graph(sam1: bytes, sam2: bytes, threads: int, strict: bool = False):
    """-> (set of vertex names, set of frozenset({name_u, name_v}) simple edges)."""
    u1 = read_sam(sam1, threads, strict)
    u2 = read_sam(sam2, threads, strict)
    names = set()
    for m in (u1, u2):
        for s in m.values():
            names |= s
    for k, s in u2.items():
        u1.setdefault(k, set()).update(s)
    edges = set()
    for s in u1.values():
        c = sorted(s)
        for i in range(len(c)):
            for j in range(i + 1, len(c)):
                edges.add(frozenset((c[i], c[j])))
    return names, edges


def make_fixture(n_unitigs=2000, n_lines=20000, seed=1):
    """unitig FASTA (ggcat-style headers, some multi-line
    records) and two SAM files with @SQ headers, multi-mapped reads, ~5% '*'."""
    rnd = random.Random(seed)
    fasta = []
    for u in range(n_unitigs):
        seq = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(40, 150)))
        fasta.append(f">{u} LN:i:{len(seq)} KC:i:{rnd.randint(1, 99)}\n")
        if u % 7 == 0:
            half = len(seq) // 2
            fasta.append(seq[:half] + "\n" + seq[half:] + "\n")
        else:
            fasta.append(seq + "\n")
    w = [(i + 1) ** -0.8 for i in range(n_unitigs)]

    def sam(mate):
        out = [f"@SQ\tSN:{u}\tLN:100\n" for u in range(0, n_unitigs, 97)]
        out.append("@PG\tID:bwa-mem2\tPN:bwa-mem2\n")
        nreads = n_lines // 2
        lines = 0
        r = 0
        while lines < n_lines:
            rid = rnd.randrange(nreads)
            qname = f"r{rid}/{mate}" if rid % 5 == 0 else f"r{rid}"
            for _ in range(rnd.choice((1, 1, 1, 2, 3))):
                rname = "*" if rnd.random() < 0.05 else str(rnd.choices(range(n_unitigs), weights=w)[0])
                out.append(f"{qname}\t{rnd.choice((0, 16, 256))}\t{rname}\t{rnd.randint(1, 90)}\t60\t50M\t*\t0\t0\tACGT\tIIII\n")
                lines += 1
            r += 1
        return "".join(out).encode()

    return "".join(fasta).encode(), sam(1), sam(2)


# ---- v1 outputs (combineFile / splitAnomalousUnitigs), restated for the checker -------------------

def read_unitigs(fasta: bytes):
    """readUnitigsFile (src/graph.cpp:565-589): name = text between '>' and the first space;
    each sequence line contributes all but its last character."""
    out, cur = {}, None
    for ln in fasta.decode().splitlines(keepends=True):
        if ln.startswith(">"):
            sp = ln.find(" ")
            cur = ln[1:sp] if sp >= 0 else ln[1:]
            out[cur] = ""
        else:
            out[cur] += ln[:-1]
    return out


def combined_fasta(kcore_tsv: str, unitigs: dict) -> str:
    """combineFile (src/graph.cpp:591-635)."""
    out = []
    for ln in kcore_tsv.splitlines():
        if ln.startswith("#"):
            continue
        f = ln.split("\t")
        out.append(f">Unitig_{f[1]}|{f[2]}\n")
        if f[1] in unitigs:
            out.append(unitigs[f[1]] + "\n")
    return "".join(out)


def _v1_median(v, start, end):
    """getMedian (src/graph.cpp:650-665), window quirk kept: size = end - start - 1."""
    size = end - start - 1
    if size % 2 == 0:
        return (v[start + size // 2 - 1] + v[start + size // 2]) / 2
    return v[start + (size - 1) // 2]


def split_anomalous(corea_txt: str, unitigs: dict, names=None):
    """splitAnomalousUnitigs (src/graph.cpp:667-749): returns (top, low, cutoff). names=None keeps the
    reference's behaviour (row i decided by the i-th smallest score, labelled by the index i);
    a list of names gives the "fixed" variant."""
    score = [float(ln.split("\t")[1]) for ln in corea_txt.splitlines()]
    n = len(score)
    s = sorted(score)
    q1 = _v1_median(s, 0, n // 2 - 1)
    q3 = _v1_median(s, n // 2, n - 1) if n % 2 == 0 else _v1_median(s, n // 2 + 1, n - 1)
    cutoff = q3 + 1.5 * (q3 - q1)
    decide = s if names is None else score
    top, low = [], []
    for i in range(n):
        key = str(i) if names is None else names[i]
        dst = top if decide[i] >= cutoff else low
        dst.append(f"Unitig_{key}\n")
        if key in unitigs:
            dst.append(unitigs[key] + "\n")
    return "".join(top), "".join(low), cutoff
