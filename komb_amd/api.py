"""Host-side Python view of the komb_accel C ABI.

Thin plumbing for tests and bench.py: one `KombAccel` object = one komb_ctx on
one GPU.  Method names follow the reference's seam: `run_core` stands where
Kgraph::runCore calls igraph_degree + igraph_coreness (src/graph.cpp:462-463),
`run_truss` where Kgraph::runTruss calls igraph_induced_subgraph_map +
igraph_trussness (src/graph.cpp:502,508), `get_anomaly_score` where
CombineCoreA::run calls CoreA::getAnomalyScore (src/CombineCoreA.h:30).
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import KombOpts, KombStats, as_c, ptr


class KombError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"komb_accel error {code}: {msg}")
        self.code = code


def gen_hug_edges(nv, n_cliques, alpha=2.6, seed=42):
    """Synthetic power-law unitig graph: raw (u,v) pairs int64[n_raw,2] (host code)."""
    lib = _lib.load()
    n_raw = lib.komb_gen_hug_edges(nv, n_cliques, alpha, seed, None)
    if n_raw < 0:
        raise ValueError("komb_gen_hug_edges: bad arguments")
    uv = np.empty((n_raw, 2), dtype=np.int64)
    got = lib.komb_gen_hug_edges(nv, n_cliques, alpha, seed, ptr(uv))
    assert got == n_raw
    return uv


# The library reads no KOMB_* environment variable (ABI 7): its tuning / test switches are per-context options
# (komb_set_option).  The TESTS still say what they want through the environment (monkeypatch.setenv("KOMB_FINISH", "lds")):
# with FORWARD_ENV_OPTIONS on (tests/conftest.py turns it on; bench.py never does) this plumbing hands those variables to
# komb_set_option before every compute call -- the forwarding is test infrastructure, the option mechanism is the ABI's.
FORWARD_ENV_OPTIONS = False
OPTION_NAMES = ("FINISH", "LOCAL_LIMIT", "LOCAL_ITEMS", "LOCAL_DENSITY", "LOCAL_DEFER_CHUNKS", "TAIL", "CORE_TAIL", "INDEX",
                "REC_CAP", "OWN_DENSE_CAP", "NO_OWN_DENSE", "NO_REC_SCRATCH", "NO_FIRST_QUEUE", "FULL_CAPS", "PREP_ROW_STAGE", "RETIRE_EVERY", "SHARD_ENGINE",
                "TRI_DEBUG", "POOL_DEBUG", "BUILD_DEBUG", "LOCAL_DEBUG", "TAIL_DEBUG")


class KombAccel:
    def __init__(self, device=0, verbosity=0, flags=0):
        self._lib = _lib.load()
        opts = KombOpts(device=device, verbosity=verbosity)
        opts.reserved[0] = flags
        self._ctx = self._lib.komb_create(ctypes.byref(opts))
        if not self._ctx:
            raise MemoryError("komb_create failed")
        self.device = device
        self.nv = -1
        self.ne = 0
        self._forwarded = {}

    def set_option(self, name, value):
        """komb_set_option: a tuning / test switch of this context (None unsets).  No option changes a result."""
        v = None if value is None else str(value).encode()
        self._check(self._lib.komb_set_option(self._ctx, name.encode(), v))

    def _sync_env_options(self):
        if not FORWARD_ENV_OPTIONS:
            return
        for name in OPTION_NAMES:
            v = os.environ.get("KOMB_" + name)
            if name == "INDEX" and os.environ.get("KOMB_TWO_PASS"):
                v = "two_pass"
            if name == "SHARD_ENGINE" and v is None:
                v = os.environ.get("KOMB_SHARD_PEEL")
            if self._forwarded.get(name) != v:
                self.set_option(name, v)
                self._forwarded[name] = v

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.komb_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise KombError(rc, self._lib.komb_last_error(self._ctx).decode())

    # ---- graph (a1)
    def from_edges(self, nv, uv):
        uv = as_c(np.asarray(uv).reshape(-1, 2), np.int64)
        self._check(self._lib.komb_graph_from_edges(self._ctx, nv, uv.shape[0], ptr(uv)))
        self._info()
        return self

    def from_csr(self, rowptr, col):
        rowptr = as_c(rowptr, np.int64)
        col = as_c(col, np.int32)
        self._check(self._lib.komb_graph_from_csr(self._ctx, len(rowptr) - 1, ptr(rowptr), ptr(col)))
        self._info()
        return self

    def _info(self):
        nv, ne = ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.komb_graph_info(self._ctx, ctypes.byref(nv), ctypes.byref(ne)))
        self.nv, self.ne = nv.value, ne.value

    def get_csr(self):
        rowptr = np.zeros(self.nv + 1, dtype=np.int64)
        col = np.zeros(2 * self.ne, dtype=np.int32)
        self._check(self._lib.komb_graph_get_csr(self._ctx, ptr(rowptr), ptr(col)))
        return rowptr, col

    # ---- k-core (a2 + a3)
    def core_run(self):
        self._sync_env_options()
        self._check(self._lib.komb_core_run(self._ctx))

    def set_shard_peel(self, on=True):
        """komb_set_shard_peel: sharded k-truss runs split the peel by edge range as well (opt-in; shard_dev.h)."""
        self._check(self._lib.komb_set_shard_peel(self._ctx, 1 if on else 0))

    def core_fetch(self):
        deg = np.zeros(self.nv, dtype=np.int32)
        core = np.zeros(self.nv, dtype=np.int32)
        self._check(self._lib.komb_core_fetch(self._ctx, ptr(deg), ptr(core)))
        return deg, core

    def run_core(self):
        """degree, coreness -- what Kgraph::runCore gets from igraph."""
        self.core_run()
        return self.core_fetch()

    # ---- k-truss (a5 + a6)
    def truss_run(self, vmask=None):
        if vmask is not None:
            vmask = as_c(vmask, np.uint8)
            if len(vmask) != self.nv:
                raise ValueError("vmask must have nv entries")
        self._sync_env_options()
        self._check(self._lib.komb_truss_run(self._ctx, ptr(vmask)))

    def truss_run_slice(self, rank, world, vmask=None):
        """komb_truss_run_slice: the whole k-truss path on this rank's copy of the graph, the results of canonical edges
        [ne*rank/world, ne*(rank+1)/world) only (zeros elsewhere); no exchange between the ranks."""
        if vmask is not None:
            vmask = as_c(vmask, np.uint8)
            if len(vmask) != self.nv:
                raise ValueError("vmask must have nv entries")
        self._sync_env_options()
        self._check(self._lib.komb_truss_run_slice(self._ctx, ptr(vmask), int(rank), int(world)))

    def truss_prepare(self):
        """komb_truss_prepare: the k-truss side of the resident graph, made now (a no-op when it exists)."""
        self._check(self._lib.komb_truss_prepare(self._ctx))

    def truss_unprepare(self):
        """komb_truss_unprepare: drop the preparation and the last k-truss result (the next k-truss call rebuilds it)."""
        self._check(self._lib.komb_truss_unprepare(self._ctx))

    def graph_moments(self):
        """komb_graph_moments (measurement only): fills stats sum_deg_sq / wedge_items / max_degree / oriented_items."""
        self._check(self._lib.komb_graph_moments(self._ctx))

    def truss_fetch_into(self, eu=None, ev=None, tr=None):
        """komb_truss_fetch into caller-owned int32 arrays of komb_truss_count entries; None = not wanted."""
        self._check(self._lib.komb_truss_fetch(self._ctx, ptr(eu) if eu is not None else None, ptr(ev) if ev is not None else None,
                                               ptr(tr) if tr is not None else None))

    def truss_fetch(self, with_support=False):
        n = ctypes.c_int64()
        self._check(self._lib.komb_truss_count(self._ctx, ctypes.byref(n)))
        eu = np.zeros(n.value, dtype=np.int32)
        ev = np.zeros(n.value, dtype=np.int32)
        tr = np.zeros(n.value, dtype=np.int32)
        self._check(self._lib.komb_truss_fetch(self._ctx, ptr(eu), ptr(ev), ptr(tr)))
        if not with_support:
            return eu, ev, tr
        sup = np.zeros(n.value, dtype=np.int32)
        self._check(self._lib.komb_truss_fetch_support(self._ctx, ptr(sup)))
        return eu, ev, tr, sup

    def run_truss(self, vmask=None, with_support=False):
        """(eu, ev, trussness) in canonical edge order, original vertex ids."""
        self.truss_run(vmask)
        return self.truss_fetch(with_support)

    # ---- CoreA (a9 + a10)
    def get_anomaly_score(self, degree, coreness):
        degree = as_c(degree, np.int32)
        coreness = as_c(coreness, np.int32)
        score = np.zeros(len(degree), dtype=np.float64)
        self._check(self._lib.komb_corea_scores(self._ctx, ptr(degree), ptr(coreness), len(degree), ptr(score)))
        return score

    def fractional_ranks(self, degree, coreness):
        degree = as_c(degree, np.int32)
        coreness = as_c(coreness, np.int32)
        rd = np.zeros(len(degree), dtype=np.float64)
        rk = np.zeros(len(degree), dtype=np.float64)
        self._check(self._lib.komb_corea_ranks(self._ctx, ptr(degree), ptr(coreness), len(degree), ptr(rd), ptr(rk)))
        return rd, rk

    def densest_block(self, suspiciousness=None):
        """komb_densest_block: (order int32[2*nv], side int32[2*nv], n_block, max_density) of the resident graph."""
        n = int(self.nv)
        order = np.zeros(max(2 * n, 1), dtype=np.int32)
        side = np.zeros(max(2 * n, 1), dtype=np.int32)
        nb = ctypes.c_int64(0)
        dens = ctypes.c_double(0.0)
        susp = None if suspiciousness is None else as_c(suspiciousness, np.float64)
        self._check(self._lib.komb_densest_block(self._ctx, ptr(susp), ptr(order), ptr(side), ctypes.byref(nb), ctypes.byref(dens)))
        return order[: 2 * n], side[: 2 * n], int(nb.value), float(dens.value)

    def stats(self):
        st = KombStats()
        self._check(self._lib.komb_get_stats(self._ctx, ctypes.byref(st)))
        return {name: getattr(st, name) for name, _ in KombStats._fields_}
