#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small summaries kept under profiles/.

    python scripts/prof_summary.py <tag> <kernel_trace_dir> [<pmc_fetch_dir> <pmc_write_dir> [<pmc_sq_dir>]]

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3 --stats), profiles/<tag>_kernels.json
(per-kernel launches / total / average from the kernel trace) and, when the two PMC directories
are given, profiles/<tag>_traffic.json: HBM bytes per launch from FETCH_SIZE and WRITE_SIZE
(separate --pmc passes, kilobyte units).  FETCH_SIZE was calibrated on this box
(scripts/calib/fetch_calib.hip, profiles/r01_fetch_size_calibration.txt): coalesced streams of 4, 8 and
16 B/lane read exactly 1/2, random 4-byte gathers exactly one 64-byte line each; the kernels here mix both,
so the raw sum (lower bound) and 2*FETCH+WRITE (upper bound) are both recorded.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.replace("komb::(anonymous namespace)::", "").replace("komb::", "").replace("void ", "")
    if n.startswith("k_peel_step<"):
        for key, nice in (("TrussProblem", "Truss"), ("CoreProblem", "Core"), ("TrussCollect", "TrussCollect"), ("CoreCollect", "CoreCollect")):
            if key in n:
                return f"k_peel_step<{nice}>"
    if n.startswith("k_local_step<"):
        return "k_local_step<Truss>" if "TrussLocal" in n else "k_local_step<Core>"
    if n.startswith("k_triangles<2") and n.split("(")[0].rstrip().endswith("true, true>"):
        return "k_triangles<stream>"
    for key, nice in (("k_wedges<2", "k_wedges<stream>"), ("k_wedges<0", "k_wedges<count>"), ("k_triangles<0", "k_triangles<count>"), ("k_triangles<1", "k_triangles<fill>"),
                      ("k_triangles<2", "k_triangles<single>"), ("k_compact_inc<", "k_compact_inc")):
        if n.startswith(key):
            return nice
    if n.startswith("k_slot_filter<"):
        return n.split("(")[0].replace(" ", "")
    if "rocprim" in n:
        return "rocprim::" + ("radix_sort" if ("radix_sort" in n or "onesweep" in n) else "scan" if "scan" in n else "partition" if "partition" in n else "other")
    return n.split("(")[0]


def main():
    tag, trace_dir = sys.argv[1], sys.argv[2]
    out = os.environ.get("KOMB_PROF_OUT", os.path.join(ROOT, "profiles"))    # on the GPU box: a directory under gpurun_out/
    os.makedirs(out, exist_ok=True)
    stats = glob.glob(os.path.join(trace_dir, "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
    trace = glob.glob(os.path.join(trace_dir, "**", "*_kernel_trace.csv"), recursive=True)[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    kernels = {k: {"launches": len(v), "total_ms": sum(v) / 1e3, "avg_us": sum(v) / len(v), "max_us": max(v)}
               for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))}
    json.dump(kernels, open(os.path.join(out, f"{tag}_kernels.json"), "w"), indent=1)
    print(json.dumps(kernels, indent=1)[:1500])
    if len(sys.argv) >= 6:
        # SQ counters (one --pmc pass): per kernel sums, and the share of wave cycles spent waiting / issuing
        f = glob.glob(os.path.join(sys.argv[5], "**", "*_counter_collection.csv"), recursive=True)[0]
        sq = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            sq[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        res = {}
        for k, v in sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
            wc = v.get("SQ_WAVE_CYCLES", 0.0)
            if wc <= 0:
                continue
            res[k] = dict(v)
            res[k]["wait_any_frac"] = v.get("SQ_WAIT_ANY", 0.0) / wc
            res[k]["wait_inst_frac"] = v.get("SQ_WAIT_INST_ANY", 0.0) / wc
            res[k]["active_inst_frac"] = v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        json.dump(res, open(os.path.join(out, f"{tag}_sq.json"), "w"), indent=1)
        print({k: {"wait": round(v["wait_any_frac"], 3), "issue": round(v["active_inst_frac"], 3)} for k, v in list(res.items())[:8]})
    if len(sys.argv) >= 5:
        traffic = collections.defaultdict(lambda: {"launches": 0})
        for ctr_dir in sys.argv[3:5]:
            f = glob.glob(os.path.join(ctr_dir, "**", "*_counter_collection.csv"), recursive=True)[0]
            seen = collections.Counter()
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                    continue
                k = short(r["Kernel_Name"])
                key = r["Counter_Name"].lower() + "_bytes"
                traffic[k][key] = traffic[k].get(key, 0.0) + float(r["Counter_Value"]) * 1024.0
                seen[k] += 1
            for k, n in seen.items():
                traffic[k]["launches"] = max(traffic[k]["launches"], n)
        res = {}
        for k, v in traffic.items():
            tot = v.get("fetch_size_bytes", 0.0) + v.get("write_size_bytes", 0.0)
            hi = 2.0 * v.get("fetch_size_bytes", 0.0) + v.get("write_size_bytes", 0.0)
            res[k] = {"launches_profiled": v["launches"], "fetch_bytes": v.get("fetch_size_bytes", 0.0),
                      "write_bytes": v.get("write_size_bytes", 0.0), "hbm_bytes": tot,
                      "hbm_bytes_per_launch": tot / max(v["launches"], 1),
                      # FETCH_SIZE is exact for 64-byte-line gathers and 1/2 for coalesced streams
                      # (profiles/r01_fetch_size_calibration.txt): true bytes lie between the two
                      "hbm_bytes_per_launch_max": hi / max(v["launches"], 1)}
        json.dump(res, open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
        print({k: round(v["hbm_bytes"] / 1e9, 2) for k, v in res.items() if v["hbm_bytes"] > 1e8})


if __name__ == "__main__":
    main()
