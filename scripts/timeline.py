"""Timeline of ONE cold k-truss step out of a rocprofv3 --kernel-trace CSV: every kernel from the last k_prep_vertex to the
k_truss_results after it, with its duration and the idle gap before it; sums of busy and idle time.
usage: timeline.py <dir with *_kernel_trace.csv> [min_gap_us] [truss|core]   (core: the last k-core pass, k_core_init .. its k_local_finish)"""
import csv, glob, os, sys
d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(({"name": r["Kernel_Name"], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])} for r in csv.DictReader(open(f))), key=lambda r: r["s"])
what = sys.argv[3] if len(sys.argv) > 3 else "truss"
first, last = ("k_prep_vertex", "k_truss_results") if what == "truss" else ("k_core_init", "k_local_finish")
starts = [i for i, r in enumerate(rows) if first in r["name"]]
i0 = starts[-1]
i1 = next(i for i in range(i0, len(rows)) if last in rows[i]["name"])
step = rows[i0:i1 + 1]
busy = sum(r["e"] - r["s"] for r in step)
wall = step[-1]["e"] - step[0]["s"]
print(f"{len(step)} kernels, wall {wall/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(wall-busy)/1e6:.3f} ms")
def short(n):
    n = n.replace("komb::(anonymous namespace)::", "").replace("komb::", "").replace("void ", "")
    if "rocprim" in n:
        import re
        m = re.search(r"detail::(\w+)<", n.split("trampoline_kernel<")[-1]) if "trampoline" in n else None
        return "rocprim:" + (m.group(1) if m else n[:40])
    return n.split("(")[0][:60]
prev = step[0]["s"]
agg = {}
for r in step:
    gap = (r["s"] - prev) / 1e3
    dur = (r["e"] - r["s"]) / 1e3
    k = short(r["name"])
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += dur; a[2] += max(gap, 0.0)
    if gap >= min_gap or dur >= min(100.0, 10 * min_gap):
        print(f"  +{(r['s']-step[0]['s'])/1e3:9.1f} us  gap {gap:7.1f}  dur {dur:8.1f}  {k}")
    prev = max(prev, r["e"])
print("by kernel: calls, busy us, gap-before us")
for k, a in sorted(agg.items(), key=lambda x: -(x[1][1] + x[1][2])):
    print(f"  {k:60s} {a[0]:4d} {a[1]:9.1f} {a[2]:9.1f}")
