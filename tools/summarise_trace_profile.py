#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/profile_trace.sh) -> profiles/<round>_trace_{kernel_stats.csv,bench.json}.
usage: tools/summarise_trace_profile.py <tag> <round-prefix>"""
import collections, csv, glob, json, shutil, sys, os


def newest(pattern):
    """a re-run into the same tag leaves the older run's files beside the new ones: take the latest"""
    return max(glob.glob(pattern), key=os.path.getmtime)

tag, rp = sys.argv[1], sys.argv[2]
src = f"gpurun_out/prof_{tag}"
shutil.copy(newest(f"{src}/trace/*/*_kernel_stats.csv"), f"profiles/{rp}_trace_kernel_stats.csv")
plain = json.load(open(f"{src}/bench_plain.json")); under = json.load(open(f"{src}/bench_trace.json"))
out = {"command": f"tools/profile_trace.sh {tag} --reads 32768 --reps 2  (plain run, rocprofv3 --kernel-trace --stats run, one --pmc pass per counter)",
       "plain": plain, "under_rocprof_kernel_trace": under, "pmc": {}}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    agg = collections.defaultdict(list)
    if not glob.glob(f"{src}/{sub}/*/*_counter_collection.csv"):
        continue
    for r in csv.DictReader(open(newest(f"{src}/{sub}/*/*_counter_collection.csv"))):
        if "k_trace_pairs" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        out["pmc"][c + "_KB_per_launch"] = sum(v) / len(v)
st = [r for r in csv.DictReader(open(f"profiles/{rp}_trace_kernel_stats.csv")) if "k_trace_pairs" in r["Name"]][0]
avg_ms = float(st["AverageNs"]) / 1e6
wb = out["pmc"]["WRITE_SIZE_KB_per_launch"] * 1024; fb = out["pmc"]["FETCH_SIZE_KB_per_launch"] * 1024
alg = plain["pairs"] * plain["parent_bits_bytes_per_pair"]
stored = plain["pairs"] * plain.get("stored_bytes_per_pair", plain["parent_bits_bytes_per_pair"])
ck = plain["form"].startswith("checkpoint")
sq = out["pmc"]
valu = sq.get("SQ_INSTS_VALU_KB_per_launch")            # (the helper above names every counter "..._KB_per_launch": a plain count here)
VALU_PEAK = 256 * 4 * 2.4e9 / 2
traffic = int(wb + 2 * fb)
out["roofline"] = {"bound": "valu-issue (recomputation)" if ck else "hbm", "kernel": st["Name"].split("(")[0].replace("void ", ""),
                   "launch_ms_rocprof": round(avg_ms, 3),
                   "hbm": {"stored_bytes_per_launch": stored, "traffic": traffic, "pmc_write_bytes": int(wb), "pmc_fetch_bytes_raw": int(fb),
                           "achieved": round(traffic / avg_ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                           "frac": round(traffic / avg_ms / 1e6 / 8000, 4),
                           "parent_bits_bytes_per_launch": alg, "parent_bits_equivalent_GBps": round(alg / avg_ms / 1e6, 1)},
                   "valu": {"insts_per_launch": valu, "achieved": round(valu / avg_ms / 1e6, 1) if valu else None,
                            "peak": round(VALU_PEAK / 1e9, 1), "unit": "G wave-instr/s",
                            "frac": round(valu / (avg_ms * 1e-3) / VALU_PEAK, 4) if valu else None},
                   "note": "checkpoint form: the sweep stores one checkpoint per 32 steps (stored_bytes), the walk re-runs a chunk at a time "
                           "into LDS; HBM traffic (WRITE_SIZE + 2 x FETCH_SIZE, gfx950 correction) is a fraction of the peak and the kernel "
                           "is bound by the recomputation's integer instructions.  parent_bits_*: the 2 parent bits per processed cell "
                           "that the streamed form (PBA_TRACE_STREAM=1) writes and reads back -- the rate it would need for the same "
                           "scripts/s is above the HBM peak" if ck else
                           "stream form: 2 parent bits per cell the sweep processes, whole 128-byte lines of the 16-lane groups with an open window"}
json.dump(out, open(f"profiles/{rp}_trace_bench.json", "w"), indent=1)
print(json.dumps(out["roofline"]))
