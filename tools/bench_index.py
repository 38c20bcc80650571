#!/usr/bin/env python3
"""Seed-index build rate (0.25 B read + 8 B written per indexed position = 8.25 B): builds the locator index
(PBA_INDEX_ALL) of synthetic genomes of growing size and prints one JSON line per size with a roofline block."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pacbioassembly_amd import Context, engine as eng

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", type=str, default="5000000,50000000,500000000")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
ctx = Context(0)
mask = eng.mask_from_pattern("111*11*11*1*1111")
for L in (int(x) for x in a.sizes.split(",")):
    g = eng.synth_genome(2, L)
    T = ctx.seqs_from_text(g, np.array([0, L], np.uint64), strict_acgt=True)
    del g
    ix = ctx.index_build(T, 0, mask, eng.PBA_INDEX_ALL); ix.close()            # warm-up
    ms, wall = [], []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        ix = ctx.index_build(T, 0, mask, eng.PBA_INDEX_ALL)
        wall.append(time.perf_counter() - t0)
        ms.append(ctx.last_profile()["index_ms"])
        n = ix.entries
        ix.close()
    t = float(np.median(ms)) * 1e-3
    algo = 8.25 * L
    print(json.dumps({"workload": f"locator seed index of a {L}-base genome (mask 111*11*11*1*1111)", "positions": L, "entries": int(n),
                      "index_ms": round(t * 1e3, 3), "wall_ms": round(float(np.median(wall)) * 1e3, 3),
                      "roofline": {"bound": "hbm", "kernel": "k_seed_count + k_seed_scatter + k_lvl_* + k_seg_sort", "unit": "GB/s", "peak": 8000.0,
                                   "achieved": round(algo / t / 1e9, 1), "frac": round(algo / t / 8e12, 5),
                                   "algorithmic_bytes": int(algo)}}))
    T.close()
