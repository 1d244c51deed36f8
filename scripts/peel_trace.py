"""Debug: run one k-truss (and k-core) pass with KOMB_PEEL_TRACE and summarise cost per step."""
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
a.truss_run()                      # warm the pool
os.environ["KOMB_PEEL_TRACE"] = out
a.truss_run()
a.core_run()
