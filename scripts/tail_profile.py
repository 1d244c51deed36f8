"""Debug: per truss level k, the size of the sub-graph still alive when level k starts
(edges with trussness >= k, their endpoints), next to the traced cost of that level's steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import komb_amd
cfg = {"c3": (10_000_000, 24_250_000), "c2": (1_000_000, 2_450_000)}[sys.argv[1] if len(sys.argv) > 1 else "c3"]
out = "gpurun_out/peel_trace.txt"
os.makedirs("gpurun_out", exist_ok=True)
if os.path.exists(out): os.remove(out)
uv = komb_amd.gen_hug_edges(cfg[0], cfg[1], 2.6, 42)
a = komb_amd.KombAccel(); a.from_edges(cfg[0], uv); del uv
eu, ev, tr = a.run_truss()
os.environ["KOMB_PEEL_TRACE"] = out
a.truss_run()
del os.environ["KOMB_PEEL_TRACE"]
rows = np.array([[float(x) for x in l.split()] for l in open(out) if not l.startswith("#")])
order = np.argsort(-tr, kind="stable")
trs = tr[order]; eus = eu[order]; evs = ev[order]
print("k  edges>=k  vertices  maxdeg  steps  us")
for k in range(3, int(tr.max()) + 1):
    n = int(np.searchsorted(-trs, -k, side="right"))
    if n > 3_000_000:
        nvx = mdeg = -1
    else:
        vs, cnt = np.unique(np.concatenate([eus[:n], evs[:n]]), return_counts=True)
        nvx, mdeg = len(vs), int(cnt.max())
    lv = rows[rows[:, 1] == k - 2]
    print(k, n, nvx, mdeg, len(lv), int(lv[:, 8].sum()))
