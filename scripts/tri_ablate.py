"""Debug: time k_triangles with parts switched off (KOMB_TRI_ABLATE; results are wrong on purpose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import komb_amd
nv, ncl = 10_000_000, 24_250_000
uv = komb_amd.gen_hug_edges(nv, ncl, 2.6, 42)
a = komb_amd.KombAccel(); a.from_edges(nv, uv); del uv
a.truss_run()
for mode in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("single", "twopass")):
    if mode == "twopass": os.environ["KOMB_TWO_PASS"] = "1"
    for ab in (0, 4, 16, 32, 48, 1, 2, 3, 7, 8):   # 16: no own-role stores, 32: no third-role atomic + store
        os.environ["KOMB_TRI_ABLATE"] = str(ab)
        try:
            a.truss_run()
        except Exception as e:
            print(mode, ab, "error", str(e)[:60])
        s = a.stats()
        print(f"{mode:8s} ablate={ab}: count {s['ms_tri_count']:.2f} fill {s['ms_tri_fill']:.2f} compact {s['ms_compact']:.2f}", flush=True)
