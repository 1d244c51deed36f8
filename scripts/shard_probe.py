"""The sharded peel's engine with ONE rank and no collective (option SHARD_ENGINE) beside the replicated engine: what the
host-driven protocol costs by itself.  usage: shard_probe.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import komb_amd, bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
nv, ncl, alpha, seed = bench.CONFIGS[cfg][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
a = komb_amd.KombAccel(); a.from_edges(nv, uv); del uv
ref = None
for eng in (None, "1", None, "1"):
    a.set_option("SHARD_ENGINE", eng)
    a.truss_run(); st = a.stats(); a.core_run(); sc = a.stats()
    tr = a.truss_fetch()[2]; core = a.core_fetch()[1]
    if ref is None: ref = (tr.copy(), core.copy())
    print(f"{cfg} SHARD_ENGINE={eng}: truss peel {st['ms_peel']:.2f} ms ({st['truss_subrounds']} sub-rounds, {st['shard_exchanges']} exchanges, {st['ms_exchange']:.2f} ms in them), "
          f"k-core {sc['ms_core']:.2f} ms ({sc['core_subrounds']} sub-rounds, {sc['shard_exchanges']} exchanges); same results: {np.array_equal(tr, ref[0]) and np.array_equal(core, ref[1])}", flush=True)
