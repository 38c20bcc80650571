#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/profile_trace.sh) -> profiles/<round>_trace_{kernel_stats.csv,bench.json}.
usage: tools/summarise_trace_profile.py <tag> <round-prefix>"""
import collections, csv, glob, json, shutil, sys
tag, rp = sys.argv[1], sys.argv[2]
src = f"gpurun_out/prof_{tag}"
shutil.copy(glob.glob(f"{src}/trace/*/*_kernel_stats.csv")[0], f"profiles/{rp}_trace_kernel_stats.csv")
plain = json.load(open(f"{src}/bench_plain.json")); under = json.load(open(f"{src}/bench_trace.json"))
out = {"command": f"tools/profile_trace.sh {tag} --reads 32768 --reps 2  (plain run, rocprofv3 --kernel-trace --stats run, one --pmc pass per counter)",
       "plain": plain, "under_rocprof_kernel_trace": under, "pmc": {}}
for sub in ("pmc_fetch", "pmc_write"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f"{src}/{sub}/*/*_counter_collection.csv")[0])):
        if "k_trace_pairs" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        out["pmc"][c + "_KB_per_launch"] = sum(v) / len(v)
st = [r for r in csv.DictReader(open(f"profiles/{rp}_trace_kernel_stats.csv")) if "k_trace_pairs" in r["Name"]][0]
avg_ms = float(st["AverageNs"]) / 1e6
wb = out["pmc"]["WRITE_SIZE_KB_per_launch"] * 1024; fb = out["pmc"]["FETCH_SIZE_KB_per_launch"] * 1024
alg = plain["pairs"] * plain["parent_bits_bytes_per_pair"]
out["roofline"] = {"bound": "hbm", "kernel": st["Name"].split("(")[0].replace("void ", ""), "launch_ms_rocprof": round(avg_ms, 3),
                   "algorithmic_bytes_per_launch": alg, "achieved": round(alg / avg_ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                   "frac": round(alg / avg_ms / 1e6 / 8000, 4), "traffic": int(wb + 2 * fb), "pmc_write_bytes": int(wb),
                   "pmc_fetch_bytes_raw": int(fb),
                   "note": "algorithmic = 2 parent bits per cell the sweep processes (window columns x NB x 8 B per pair); the kernel stores whole "
                           "128-byte lines of the 16-lane groups with an open window (pmc_write_bytes); reads are the walk's 64-step "
                           "tiles (raw FETCH_SIZE, x2 in traffic per the gfx950 correction)"}
json.dump(out, open(f"profiles/{rp}_trace_bench.json", "w"), indent=1)
print(json.dumps(out["roofline"]))
