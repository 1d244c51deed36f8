#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 results database (kernel trace): ktdb.py <results.db> [top]"""
import collections
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^_ZN4komb\d*", "", n)
    m = re.search(r"(k_[a-z_0-9]+)", n)
    if m:
        tag = m.group(1)
        for key in ("TrussProblem", "CoreProblem", "TrussCollect", "CoreCollect", "TrussLocal", "CoreLocal", "PredOrient", "PredMask"):
            if key in n:
                tag += f"<{key}>"
        if tag.startswith("k_triangles"):
            tag += "<" + re.search(r"k_trianglesI(\w+?)EEv", n).group(1) + ">" if re.search(r"k_trianglesI(\w+?)EEv", n) else ""
        if "k_slot_filter" in tag:
            tag += "<fill>" if "ELb1EEEv" in n else "<count>"
        return tag
    if "rocprim" in n:
        for key in ("onesweep", "radix_sort", "scan", "partition", "histogram"):
            if key in n:
                return "rocprim::" + key
        return "rocprim::other"
    return n[:60]


def main():
    c = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    disp = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    sym = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_info_kernel_symbol%'")][0]
    rows = c.execute(f"select k.kernel_name, d.end-d.start from {disp} d join {sym} k on d.kernel_id=k.id").fetchall()
    agg = collections.defaultdict(lambda: [0, 0])
    for n, t in rows:
        a = agg[short(n)]
        a[0] += 1
        a[1] += t
    print(f"{'total ms':>10} {'calls':>7} {'avg us':>10}  kernel")
    for n, (cnt, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
        print(f"{t / 1e6:10.2f} {cnt:7d} {t / cnt / 1e3:10.1f}  {n}")


if __name__ == "__main__":
    main()
