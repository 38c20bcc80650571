#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/profile_bench.sh) into the committed summaries under profiles/.
usage: tools/summarise_profiles.py <tag> <round-prefix, e.g. r01>"""
import collections, csv, glob, json, os, shutil, sys


def newest(pattern):
    """a re-run into the same tag leaves the older run's files beside the new ones: take the latest"""
    return max(glob.glob(pattern), key=os.path.getmtime)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pacbioassembly_amd import build as pba_build
tag, rp = sys.argv[1], sys.argv[2]
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
shutil.copy(newest(f"{src}/trace/*/*_kernel_stats.csv"), f"profiles/{rp}_bench_kernel_stats.csv")
shutil.copy(f"{src}/bench_trace.json", f"profiles/{rp}_bench_under_rocprof.json")

def counters(sub):
    f = sorted(glob.glob(f"{src}/{sub}/*/*_counter_collection.csv"), key=os.path.getmtime, reverse=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f[0])):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} | {"dispatches": len(next(iter(d.values())))} for k, d in agg.items()}

hbm = {"command": "rocprofv3 --pmc <CTR> --kernel-trace --output-format csv -- python3 bench.py --cpu-sample 0 --steps 3 --warmup 1 (one pass per counter group; tools/profile_bench.sh)",
       "units": "FETCH_SIZE / WRITE_SIZE in KB per dispatch; averages over the dispatches of each kernel",
       "gfx950_correction": "MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of the bytes of a wide coalesced streaming read; k_locate's loads are 8/16-byte per-lane reads (uncalibrated width), so both the raw and the x2 figure are given",
       "kernels": {}}
for sub in ("pmc_fetch", "pmc_write"):
    for k, d in counters(sub).items():
        hbm["kernels"].setdefault(k, {}).update({c + "_KB_avg" if c != "dispatches" else c: round(v, 3) for c, v in d.items()})
json.dump(hbm, open(f"profiles/{rp}_bench_pmc_hbm.json", "w"), indent=1)
sq = {"note": "SQ_* per dispatch, averaged; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY are in quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs",
      "kernels": {}}
for sub in ("pmc_sq", "pmc_grbm"):
    for k, d in counters(sub).items():
        sq["kernels"].setdefault(k, {}).update({c: v for c, v in d.items()})
json.dump(sq, open(f"profiles/{rp}_bench_pmc_sq.json", "w"), indent=1)
loc = next(k for k in hbm["kernels"] if k.startswith("k_locate"))
h, s = hbm["kernels"][loc], sq["kernels"].get(loc, {})
traffic = {"kernel": loc,
           "k_locate_hbm_bytes_per_launch": int((2 * h["FETCH_SIZE_KB_avg"] + h["WRITE_SIZE_KB_avg"]) * 1024),
           "raw_fetch_bytes": int(h["FETCH_SIZE_KB_avg"] * 1024), "raw_write_bytes": int(h["WRITE_SIZE_KB_avg"] * 1024),
           "method": f"profiles/{rp}_bench_pmc_hbm.json: 2*FETCH_SIZE + WRITE_SIZE (gfx950 half-count correction applied to the read side)",
           "valu_insts_per_launch": s.get("SQ_INSTS_VALU"), "salu_insts_per_launch": s.get("SQ_INSTS_SALU"),
           "gpu_cycles_per_launch": (s.get("GRBM_GUI_ACTIVE") or 0) / 8 or None, "round": int(rp[1:]),
           # the kernels these counters were collected on: bench.py drops the PMC-derived fields when the sources have changed
           "source_digest": pba_build.source_digest()}
json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(traffic))
