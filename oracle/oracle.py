"""ctypes binding of oracle/liboracle.so -- the CPU checker.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from komb_amd/ (the product).  See
komb_oracle.h for the parity status of each half ("parity unpinned" for the
igraph half; CoreA half pinned by oracle/_ref/corea_ref).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i64 = ctypes.c_int64


def build():
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_simplify.restype = _i64
        L.orc_simplify.argtypes = [_i64, _i64, _i64p, _i64p, _i32p]
        L.orc_degree.restype = None
        L.orc_degree.argtypes = [_i64, _i64p, _i32p]
        L.orc_coreness.restype = ctypes.c_int32
        L.orc_coreness.argtypes = [_i64, _i64p, _i32p, _i32p]
        L.orc_induced_subgraph.restype = _i64
        L.orc_induced_subgraph.argtypes = [_i64, _i64p, _i32p, _u8p, _i64p, _i32p, _i32p]
        L.orc_edge_list.restype = _i64
        L.orc_edge_list.argtypes = [_i64, _i64p, _i32p, _i32p, _i32p]
        L.orc_support.restype = _i64
        L.orc_support.argtypes = [_i64, _i64p, _i32p, _i32p]
        L.orc_trussness.restype = ctypes.c_int32
        L.orc_trussness.argtypes = [_i64, _i64p, _i32p, _i32p]
        L.orc_fractional_rank_faithful.restype = None
        L.orc_fractional_rank_faithful.argtypes = [_f64p, _i64, _f64p]
        L.orc_fractional_rank_fast.restype = None
        L.orc_fractional_rank_fast.argtypes = [_i64p, _i64, _f64p]
        L.orc_corea_scores.restype = None
        L.orc_corea_scores.argtypes = [_i32p, _i32p, _i64, ctypes.c_int, _f64p]
        L.orc_run_merge.restype = ctypes.c_int32
        L.orc_run_merge.argtypes = [_i64, _i64p, _i32p, ctypes.c_void_p, _i32p, _i32p, ctypes.POINTER(ctypes.c_double)]
        _LIB = L
    return _LIB


_NATIVE = None


def native_lib():
    """liboracle built for THIS machine (-march=native, OpenMP): bench.py's CPU baseline.  None if it cannot be built."""
    global _NATIVE
    if _NATIVE is None:
        path = os.path.join(_HERE, "_native", "liboracle_native.so")
        try:
            subprocess.run(["make", "-s", "-C", _HERE, "native"], check=True, stdout=subprocess.DEVNULL)
            L = ctypes.CDLL(path)
            for name in ("orc_trussness", "orc_trussness_omp"):
                getattr(L, name).restype = ctypes.c_int32
            L.orc_trussness.argtypes = [_i64, _i64p, _i32p, _i32p]
            L.orc_trussness_omp.argtypes = [_i64, _i64p, _i32p, _i32p, ctypes.c_int]
            _NATIVE = L
        except (OSError, subprocess.CalledProcessError):
            _NATIVE = False
    return _NATIVE or None


def trussness_native(rowptr, col, threads=1):
    """orc_trussness (threads == 1) or orc_trussness_omp from the native build."""
    L = native_lib()
    if L is None:
        raise RuntimeError("native oracle build unavailable")
    nv = len(rowptr) - 1
    ne = int(rowptr[nv]) // 2
    t = np.zeros(max(ne, 1), dtype=np.int32)
    if threads == 1:
        L.orc_trussness(nv, rowptr, _colbuf(col), t)
    else:
        L.orc_trussness_omp(nv, rowptr, _colbuf(col), t, int(threads))
    return t[:ne]


def ref_corea_path():
    p = os.path.join(_HERE, "_ref", "corea_ref")
    return p if os.path.exists(p) else None


# ------------------------------------------------------------------ wrappers
def simplify(nv, uv):
    """a1: raw (u,v) pairs [n_raw,2] int64 -> (rowptr int64[nv+1], col int32[2*ne])."""
    uv = np.ascontiguousarray(np.asarray(uv, dtype=np.int64).reshape(-1, 2))
    n_raw = uv.shape[0]
    rowptr = np.zeros(nv + 1, dtype=np.int64)
    col = np.zeros(max(2 * n_raw, 1), dtype=np.int32)
    ne = lib().orc_simplify(nv, n_raw, uv.reshape(-1), rowptr, col)
    if ne < 0:
        raise ValueError("orc_simplify: bad input")
    return rowptr, np.ascontiguousarray(col[: 2 * ne])


def degree(rowptr):
    nv = len(rowptr) - 1
    d = np.zeros(max(nv, 1), dtype=np.int32)
    lib().orc_degree(nv, rowptr, d)
    return d[:nv]


def _colbuf(col):
    return col if len(col) else np.zeros(1, dtype=np.int32)


def coreness(rowptr, col):
    nv = len(rowptr) - 1
    c = np.zeros(max(nv, 1), dtype=np.int32)
    lib().orc_coreness(nv, rowptr, _colbuf(col), c)
    return c[:nv]


def edge_list(rowptr, col):
    nv = len(rowptr) - 1
    ne = int(rowptr[nv]) // 2
    eu = np.zeros(max(ne, 1), dtype=np.int32)
    ev = np.zeros(max(ne, 1), dtype=np.int32)
    lib().orc_edge_list(nv, rowptr, _colbuf(col), eu, ev)
    return eu[:ne], ev[:ne]


def support(rowptr, col):
    nv = len(rowptr) - 1
    ne = int(rowptr[nv]) // 2
    s = np.zeros(max(ne, 1), dtype=np.int32)
    t = lib().orc_support(nv, rowptr, _colbuf(col), s)
    return s[:ne], int(t)


def trussness(rowptr, col):
    nv = len(rowptr) - 1
    ne = int(rowptr[nv]) // 2
    t = np.zeros(max(ne, 1), dtype=np.int32)
    lib().orc_trussness(nv, rowptr, _colbuf(col), t)
    return t[:ne]


def induced_subgraph(rowptr, col, vmask):
    nv = len(rowptr) - 1
    vmask = np.ascontiguousarray(vmask, dtype=np.uint8)
    srp = np.zeros(nv + 1, dtype=np.int64)
    scol = np.zeros(max(len(col), 1), dtype=np.int32)
    inv = np.zeros(max(nv, 1), dtype=np.int32)
    ns = lib().orc_induced_subgraph(nv, rowptr, _colbuf(col), vmask, srp, scol, inv)
    srp = np.ascontiguousarray(srp[: ns + 1])
    return srp, np.ascontiguousarray(scol[: int(srp[ns])]), inv[:ns]


def trussness_induced(rowptr, col, vmask):
    """a5+a6 as KOMB's runTruss composes them (src/graph.cpp:502,508,529-532):
    returns (eu, ev, truss) with ORIGINAL vertex ids, canonical order."""
    srp, scol, inv = induced_subgraph(rowptr, col, vmask)
    t = trussness(srp, scol)
    su, sv = edge_list(srp, scol)
    return inv[su].astype(np.int32), inv[sv].astype(np.int32), t


def fractional_rank_faithful(scores):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    out = np.zeros(max(len(scores), 1), dtype=np.float64)
    lib().orc_fractional_rank_faithful(scores if len(scores) else out, len(scores), out)
    return out[: len(scores)]


def fractional_rank_fast(keys):
    keys = np.ascontiguousarray(keys, dtype=np.int64)
    out = np.zeros(max(len(keys), 1), dtype=np.float64)
    lib().orc_fractional_rank_fast(keys if len(keys) else np.zeros(1, np.int64), len(keys), out)
    return out[: len(keys)]


def corea_scores(deg, core, faithful=False):
    deg = np.ascontiguousarray(deg, dtype=np.int32)
    core = np.ascontiguousarray(core, dtype=np.int32)
    n = len(deg)
    out = np.zeros(max(n, 1), dtype=np.float64)
    if n:
        lib().orc_corea_scores(deg, core, n, 1 if faithful else 0, out)
    return out[:n]


def ref_corea_scores(deg, core):
    """Scores from the REFERENCE's own CoreA.h (oracle/_ref/corea_ref)."""
    exe = ref_corea_path()
    if exe is None:
        raise FileNotFoundError("oracle/_ref/corea_ref not built")
    text = "%d\n" % len(deg) + "".join("%d %d\n" % (int(d), int(c)) for d, c in zip(deg, core))
    out = subprocess.run([exe, "arrays"], input=text, capture_output=True, text=True, check=True).stdout
    return np.array([float(x) for x in out.split()], dtype=np.float64)


def run_merge(rowptr, col, susp=None):
    """a12 + a13: the greedy densest-block peel of CombineCoreA::runMerge over its indexed min-heaps.
    Returns (order int32[2*nv], side int32[2*nv], n_block, max_density)."""
    nv = len(rowptr) - 1
    order = np.zeros(max(2 * nv, 1), dtype=np.int32)
    side = np.zeros(max(2 * nv, 1), dtype=np.int32)
    dens = ctypes.c_double(0.0)
    sp = None
    if susp is not None:
        susp = np.ascontiguousarray(susp, dtype=np.float64)
        sp = susp.ctypes.data_as(ctypes.c_void_p)
    nb = lib().orc_run_merge(nv, rowptr, _colbuf(col), sp, order, side, ctypes.byref(dens))
    return order[: 2 * nv], side[: 2 * nv], int(nb), float(dens.value)


def ref_merge_path():
    p = os.path.join(_HERE, "_ref", "merge_ref")
    return p if os.path.exists(p) else None


def ref_run_merge(rowptr, col, susp=None):
    """The same from the REFERENCE's own HashIndexedMinHeap.h (oracle/_ref/merge_ref: its class under a restatement of the loop)."""
    import tempfile
    exe = ref_merge_path()
    if exe is None:
        raise FileNotFoundError("oracle/_ref/merge_ref not built")
    nv = len(rowptr) - 1
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        np.array([nv, len(col)], dtype=np.int64).tofile(f)
        np.asarray(rowptr, dtype=np.int64).tofile(f)
        np.asarray(col, dtype=np.int32).tofile(f)
        np.array([0 if susp is None else 1], dtype=np.int64).tofile(f)
        if susp is not None:
            np.asarray(susp, dtype=np.float64).tofile(f)
        f.flush()
        out = subprocess.run([exe, f.name], capture_output=True, text=True, check=True).stdout.split()
    nb, dens = int(out[0]), float(out[1])
    rest = np.array(out[2:], dtype=np.int64).reshape(-1, 2)
    return rest[:, 0].astype(np.int32), rest[:, 1].astype(np.int32), nb, dens
