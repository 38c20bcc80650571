#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 counter passes under gpurun_out/pmc_<tag>/ (tools/pmc_overlap.sh).
usage: tools/pmc_summary.py <tag> [kernel filter]   -> prints a table and returns a dict when imported"""
import collections, csv, glob, json, sys


def collect(tag):
    out = collections.defaultdict(dict)
    for f in glob.glob(f"gpurun_out/pmc_{tag}/pmc_*/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            for c, v in d.items():
                # the walk / scan kernels are launched once small (warm-up) and then for real: report the largest dispatch
                out[k][c] = max(v)
                out[k]["dispatches"] = len(v)
    for f in glob.glob(f"gpurun_out/pmc_{tag}/trace/*/*_kernel_trace.csv"):
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in dur.items():
            out[k]["max_ms"] = max(v) / 1e6
            out[k]["total_ms"] = sum(v) / 1e6
    return out


if __name__ == "__main__":
    d = collect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("total_ms", 0)):
        if flt in k:
            print(k)
            print("   ", {c: (round(x, 3) if isinstance(x, float) and x < 1e6 else f"{x:.4g}") for c, x in sorted(v.items())})
