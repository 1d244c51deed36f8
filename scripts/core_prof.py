import os, sys
sys.path.insert(0, os.getcwd())
import komb_amd, bench
nv, ncl, alpha, seed = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
with komb_amd.KombAccel() as a:
    a.from_edges(nv, uv)
    for _ in range(4):
        a.core_run()
    print(a.stats()["ms_core"])
