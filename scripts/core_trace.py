import os, sys
sys.path.insert(0, os.getcwd())
import komb_amd
nv, ncl = (10_000_000, 24_250_000) if len(sys.argv) < 3 else (int(sys.argv[2]), int(sys.argv[3]))
uv = komb_amd.gen_hug_edges(nv, ncl, 2.6, 42)
a = komb_amd.KombAccel(); a.from_edges(nv, uv); del uv
out = sys.argv[1]
for rep in range(2):
    if os.path.exists(out): os.remove(out)
    os.environ["KOMB_PEEL_TRACE"] = out
    try: a.core_run()
    except Exception as e: print("err", str(e)[:80])
import numpy as np
rows=[[float(x) for x in l.split()] for l in open(out) if not l.startswith('#')]
r=np.array(rows); p=r[r[:,0]==1]; sc=r[r[:,0]==0]
for x in r[:60]: print("  mode %d L=%d light=%d heavy=%d live_mode=%d live=%d rem=%d  %.0f us" % (x[0],x[1],x[3],x[4],x[5],x[6],x[7],x[8]))
print("steps", len(r), "sum ms", r[:,8].sum()/1e3, "scan", sc[:,8].sum()/1e3, "proc", p[:,8].sum()/1e3)
w=p[:,3]+64*p[:,4]
for lo,hi in ((0,16384),(16384,262144),(262144,1e9)):
    m=(w>=lo)&(w<hi)
    if m.sum(): print(f"  frontier [{lo:.0f},{hi:.0f}): n={m.sum()} sum={p[m][:,8].sum()/1e3:.2f} ms")
