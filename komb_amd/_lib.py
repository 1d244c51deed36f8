"""ctypes loader for libkomb_accel.so (the C ABI in include/komb_accel.h).

The product path has no CPU fallback: if the shared library is missing this
module raises, and every compute entry point of the library itself returns
KOMB_ERR_DEVICE when no gfx950 device is usable.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KOMB_ACCEL_LIB selects another build of the same ABI (a tuning variant, a system-wide install)
LIB_PATH = os.environ.get("KOMB_ACCEL_LIB") or os.path.join(_HERE, "lib", "libkomb_accel.so")

KOMB_OK = 0
KOMB_ERR_ARG, KOMB_ERR_DEVICE, KOMB_ERR_NOMEM, KOMB_ERR_LIMIT, KOMB_ERR_STATE = -1, -2, -3, -4, -5


class KombOpts(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("verbosity", ctypes.c_int32),
                ("reserved", ctypes.c_int32 * 2)]


class KombStats(ctypes.Structure):
    _fields_ = [
        ("nv", ctypes.c_int64), ("ne", ctypes.c_int64), ("triangles", ctypes.c_int64),
        ("sum_deg_sq", ctypes.c_int64), ("wedge_items", ctypes.c_int64), ("oriented_items", ctypes.c_int64),
        ("max_degree", ctypes.c_int32), ("max_coreness", ctypes.c_int32), ("max_trussness", ctypes.c_int32),
        ("core_levels", ctypes.c_int32), ("core_subrounds", ctypes.c_int32),
        ("core_launches", ctypes.c_int32), ("truss_tail_runs", ctypes.c_int32),
        ("truss_levels", ctypes.c_int32), ("truss_subrounds", ctypes.c_int32),
        ("truss_scans", ctypes.c_int32), ("truss_launches", ctypes.c_int32),
        ("ms_build", ctypes.c_double), ("ms_core", ctypes.c_double), ("ms_orient", ctypes.c_double),
        ("ms_tri_count", ctypes.c_double), ("ms_tri_fill", ctypes.c_double), ("ms_compact", ctypes.c_double),
        ("ms_support", ctypes.c_double),
        ("ms_allreduce", ctypes.c_double),
        ("ms_peel", ctypes.c_double), ("ms_gather", ctypes.c_double), ("ms_corea", ctypes.c_double), ("ms_tail", ctypes.c_double),
        ("core_local_items", ctypes.c_int64), ("truss_local_items", ctypes.c_int64),
        ("core_local_units", ctypes.c_int32), ("core_local_sweeps", ctypes.c_int32),
        ("truss_local_units", ctypes.c_int32), ("truss_local_sweeps", ctypes.c_int32),
        ("ms_core_local", ctypes.c_double), ("ms_truss_local", ctypes.c_double),
        ("ms_sort", ctypes.c_double), ("tri_records", ctypes.c_int64),
        ("index_layout", ctypes.c_int32), ("shard_exchanges", ctypes.c_int32),
        ("ms_exchange", ctypes.c_double), ("exchange_words", ctypes.c_int64),
        ("ms_build_h2d", ctypes.c_double), ("ms_build_relabel", ctypes.c_double),
        ("ms_prepare", ctypes.c_double), ("truss_prepared", ctypes.c_int32), ("engine_flags", ctypes.c_int32),
        ("ms_prep_vertex", ctypes.c_double), ("ms_prep_edges", ctypes.c_double), ("ms_prep_rows", ctypes.c_double),
        ("stream_retries", ctypes.c_int32), ("reserved0", ctypes.c_int32),
    ]

KOMB_CREATE_NULL_STREAM, KOMB_CREATE_NO_WARMUP, KOMB_CREATE_WARM_UPLOAD = 1, 2, 4


# every symbol include/komb_accel.h declares: name -> (restype, argtypes)
_vp, _i64, _i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
SIGNATURES = {
    "komb_abi_version": (_i32, []),
    "komb_set_option": (_i32, [_vp, ctypes.c_char_p, ctypes.c_char_p]),
    "komb_truss_prepare": (_i32, [_vp]),
    "komb_truss_unprepare": (_i32, [_vp]),
    "komb_graph_moments": (_i32, [_vp]),
    "komb_create": (_vp, [ctypes.POINTER(KombOpts)]),
    "komb_destroy": (None, [_vp]),
    "komb_last_error": (ctypes.c_char_p, [_vp]),
    "komb_graph_from_edges": (_i32, [_vp, _i64, _i64, _vp]),
    "komb_graph_from_csr": (_i32, [_vp, _i64, _vp, _vp]),
    "komb_graph_info": (_i32, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "komb_graph_get_csr": (_i32, [_vp, _vp, _vp]),
    "komb_core_run": (_i32, [_vp]),
    "komb_core_run_sharded": (_i32, [_vp, ctypes.c_int32, ctypes.c_int32, _vp, _vp]),
    "komb_set_shard_peel": (_i32, [_vp, ctypes.c_int32]),
    "komb_core_fetch": (_i32, [_vp, _vp, _vp]),
    "komb_degree_coreness": (_i32, [_vp, _vp, _vp]),
    "komb_truss_run": (_i32, [_vp, _vp]),
    "komb_truss_run_sharded": (_i32, [_vp, _vp, ctypes.c_int32, ctypes.c_int32, _vp, _vp]),
    "komb_truss_run_slice": (_i32, [_vp, _vp, ctypes.c_int32, ctypes.c_int32]),
    "komb_truss_count": (_i32, [_vp, ctypes.POINTER(_i64)]),
    "komb_truss_fetch": (_i32, [_vp, _vp, _vp, _vp]),
    "komb_truss_fetch_support": (_i32, [_vp, _vp]),
    "komb_trussness": (_i32, [_vp, _vp, ctypes.POINTER(_i64), _vp, _vp, _vp]),
    "komb_corea_scores": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "komb_corea_ranks": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "komb_densest_block": (_i32, [_vp, _vp, _vp, _vp, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double)]),
    "komb_get_stats": (_i32, [_vp, ctypes.POINTER(KombStats)]),
    "komb_gen_hug_edges": (_i64, [_i64, _i64, ctypes.c_double, ctypes.c_uint64, _vp]),
}

ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)

_LIB = None


def load():
    """Load libkomb_accel.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C komb_amd/csrc` -- komb_amd has no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError = symbol missing from the ABI
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def as_c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)
