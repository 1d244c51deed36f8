"""komb_amd -- MI355X-native k-core / k-truss / CoreA path of KOMB.

The product is komb_amd/lib/libkomb_accel.so (hand-written HIP kernels for
gfx950 behind the C ABI of include/komb_accel.h).  This package is only the
ctypes view of that ABI used by tests/ and bench.py.
"""
from .api import KombAccel, KombError, gen_hug_edges  # noqa: F401
from . import _lib  # noqa: F401

__all__ = ["KombAccel", "KombError", "gen_hug_edges"]
