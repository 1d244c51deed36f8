"""k-truss / k-core times of a generator shape under the library variants named on the command line (komb_amd/libv/<name>; base = komb_amd/lib):
python scripts/ab_shape.py nv n_cliques alpha name..."""
import os, subprocess, sys
nv, ncl, alpha = sys.argv[1:4]
code = f"""
import os, sys
sys.path.insert(0, os.getcwd())
import komb_amd
uv = komb_amd.gen_hug_edges({nv}, {ncl}, {alpha}, 5)
a = komb_amd.KombAccel(); a.from_edges({nv}, uv)
best = None
for i in range(4):
    a.truss_run(); st = a.stats()
    t = st['ms_support'] + st['ms_peel'] + st['ms_gather']
    best = t if best is None else min(best, t)
a.core_run(); a.core_run()
print(round(best, 2), 'peel', round(st['ms_peel'], 2), 'sub-rounds', st['truss_subrounds'], 'launches', st['truss_launches'], 'core', round(a.stats()['ms_core'], 2))
"""
for name in sys.argv[4:]:
    lib = "komb_amd/lib/libkomb_accel.so" if name == "base" else f"komb_amd/libv/{name}/libkomb_accel.so"
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, KOMB_ACCEL_LIB=lib), capture_output=True, text=True)
    print(name, r.stdout.strip() or r.stderr[-300:], flush=True)
