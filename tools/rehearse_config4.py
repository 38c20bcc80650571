#!/usr/bin/env python3
"""BASELINE configs[4] (10 M x 15 kb reads all-vs-all on 8 GPUs) rehearsed as ONE rank of the eight on one MI355X: the whole
read set resident in HBM like on every rank after the packed-read all-gather, the probe table of all reads, and the rank's
share of the targets (reads [rank * n / ranks, (rank + 1) * n / ranks)) through pba_overlap_all_table range by range.

The read set is put together shard by shard -- synth_reads_range -> pba_seqs_from_text (H2D + 2-bit pack on the GPU) ->
pba_seqs_export into one device buffer -> pba_seqs_from_device_packed -- so the 150 GB of ASCII never exist at once (what a
rank of the real run does with its own shard before the all-gather).  Prints what is resident per GPU and refuses before
anything is launched if it cannot fit; then per-stage seconds for the ranges it ran (--max-ranges bounds the run: the rest of
the share is extrapolated from them, stated as such) and the implied time of the 8-GPU run.

    python tools/rehearse_config4.py --reads 10000000 --ranks 8 --rank 0 --max-ranges 6
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=15_000)
    ap.add_argument("--coverage", type=float, default=20.0)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--shard-reads", type=int, default=250_000, help="reads generated, uploaded and packed at a time")
    ap.add_argument("--targets-per-call", type=int, default=25_000)
    ap.add_argument("--max-ranges", type=int, default=6, help="target ranges of the rank's share to run (0 = all of it)")
    ap.add_argument("--R", type=float, default=0.30)
    ap.add_argument("--trials", type=int, default=32)
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()

    import torch
    from pacbioassembly_amd import Context, ProbeTable, engine as eng

    n, rl, t2 = a.reads, a.read_len, 2 * a.trials
    L = int(n * rl / a.coverage)
    pk = ((rl + 3) // 4 + 15) & ~15                                   # packed bytes of a read (16-byte aligned in a set)
    pw = (rl + 31) // 32                                              # plane word pairs of a read
    # ---- what will be resident, before anything is allocated (bytes)
    plan = {
        "packed_reads": n * pk + 2048, "bit_planes": (n * pw + 128) * 8, "offsets_lengths": n * (8 + 4 + 8),
        "probe_entries_exchanged": n * t2 * 8,                        # the all-gathered object: freed once the table stands
        "table_offsets_presence": (1 << 24) * 4 + (1 << 24) // 8, "table_probe_ids": n * t2 * 4, "table_records": n * t2 * 16,
    }
    share = (n + a.ranks - 1) // a.ranks
    t_lo, t_hi = a.rank * share, min(n, (a.rank + 1) * share)
    # per target range: survivors of the scan (~1.6 % of n * 0.057 candidates per read of the set) with 1.5 x room, the
    # overlap rows of the device (32 B per listed candidate at most), items, redo lists
    listed = int(a.targets_per_call * n * 0.057 * 0.02 * 1.5) + a.targets_per_call * 64
    plan["range_work_buffers"] = listed * (8 + 32 + 8 + 8)
    need = sum(plan.values())
    free_b, total_b = torch.cuda.mem_get_info(0)
    out = {"workload": f"BASELINE configs[4] as rank {a.rank} of {a.ranks}: {n} x {rl} reads @15%, {a.coverage}x coverage, R={a.R}, "
                       f"{a.trials} probe offsets per end; targets [{t_lo}, {t_hi}) in ranges of {a.targets_per_call}",
           "resident_bytes_planned": plan, "resident_gb_planned": round(need / 1e9, 1), "hbm_free_gb": round(free_b / 1e9, 1),
           "hbm_total_gb": round(total_b / 1e9, 1)}
    if need > free_b * 0.92:
        out["error"] = "PBA_E_NOMEM: the read set, its table and one range's work buffers do not fit this GPU -- nothing was launched"
        print(json.dumps(out))
        sys.exit(2)

    ctx = Context(0)
    mask = eng.mask_from_pattern("111*11*11*1*1111")
    t0 = time.perf_counter()
    g = eng.synth_genome(2, L)
    t_genome = time.perf_counter() - t0
    # ---- the read set, shard by shard (never the whole ASCII on the host)
    big = torch.empty(n * pk, dtype=torch.uint8, device="cuda")
    offs_all = np.arange(n, dtype=np.uint64) * np.uint64(pk)
    lens_all = np.full(n, rl, np.uint32)
    t_gen = t_up = 0.0
    for lo in range(0, n, a.shard_reads):
        hi = min(n, lo + a.shard_reads)
        t0 = time.perf_counter()
        text, offs = eng.synth_reads_range(3, g, lo, hi, rl, nthreads=a.threads)
        t_gen += time.perf_counter() - t0
        t0 = time.perf_counter()
        sh = ctx.seqs_from_text(text, offs, strict_acgt=True)
        o = sh.export(big[lo * pk:].data_ptr(), (hi - lo) * pk)
        assert int(o[0]) == 0 and (hi - lo == 1 or int(o[1]) == pk)
        sh.close()
        del text
        t_up += time.perf_counter() - t0
        print(f"reads {hi}/{n}: generate {t_gen:.1f} s, upload+pack+export {t_up:.1f} s", file=sys.stderr, flush=True)
    del g
    t0 = time.perf_counter()
    S = ctx.seqs_from_device_packed(big.data_ptr(), big.numel(), offs_all, lens_all)
    del big
    torch.cuda.empty_cache()
    t_set = time.perf_counter() - t0
    # ---- probes of every read (what the ranks all-gather: 8 B per probe), the table
    t0 = time.perf_counter()
    slots = n * t2
    probes = torch.full((slots,), -1, dtype=torch.int64, device="cuda")
    n_pe = ctx.overlap_probes(S, 0, n, mask, a.trials, probes.data_ptr(), slots)
    torch.cuda.synchronize()
    t_probe = time.perf_counter() - t0
    t0 = time.perf_counter()
    table = ProbeTable(ctx, probes.data_ptr(), probes.numel(), mask, a.trials)
    del probes
    torch.cuda.empty_cache()
    t_table = time.perf_counter() - t0
    free_after, _ = torch.cuda.mem_get_info(0)
    out["setup_s"] = {"genome": round(t_genome, 1), "generate_reads": round(t_gen, 1), "upload_pack_export": round(t_up, 1),
                      "set_from_packed_with_planes": round(t_set, 1), "probe_entries": round(t_probe, 2), "probe_table": round(t_table, 2)}
    out["probe_entries"] = int(n_pe)
    out["hbm_used_gb_reads_and_table"] = round((free_b - free_after) / 1e9, 1)

    # ---- the rank's share of the targets, range by range
    ranges = [(lo, min(t_hi, lo + a.targets_per_call)) for lo in range(t_lo, t_hi, a.targets_per_call)]
    run = ranges if a.max_ranges <= 0 else ranges[:a.max_ranges]
    per = []
    tot = {k: 0 for k in ("n_candidates", "n_pairs", "n_overlaps", "n_listed", "n_prefiltered", "cap_overflow")}
    for i, (lo, hi) in enumerate(run):
        t0 = time.perf_counter()
        ov, st = ctx.overlap_all_table(S, table, a.R, 64, lo, hi, cap=(hi - lo) * 400)
        dt = time.perf_counter() - t0
        per.append({"targets": hi - lo, "seconds": round(dt, 3), "scan_s": round(st["scan_ms"] * 1e-3, 3), "sort_s": round(st["sort_ms"] * 1e-3, 3),
                    "walk_s": round(st["walk_ms"] * 1e-3, 3), "candidates": int(st["n_candidates"]), "overlaps": int(st["n_overlaps"])})
        for k in tot:
            tot[k] += int(st[k])
        print(f"range {i + 1}/{len(run)} (of {len(ranges)}): {per[-1]}", file=sys.stderr, flush=True)
    free_end, _ = torch.cuda.mem_get_info(0)
    out["hbm_used_gb_peak_with_range_buffers"] = round((free_b - free_end) / 1e9, 1)
    steady = per[1:] if len(per) > 1 else per                          # (the first range also sizes the ctx's work buffers and takes the census)
    tgt = sum(p["targets"] for p in steady)
    sec_per_target = sum(p["seconds"] for p in steady) / max(tgt, 1)
    share_s = per[0]["seconds"] + sec_per_target * ((t_hi - t_lo) - per[0]["targets"]) if per else 0.0
    out["ranges_run"] = len(run)
    out["ranges_in_share"] = len(ranges)
    out["per_range"] = per
    out["totals_of_ranges_run"] = tot
    out["share_seconds"] = round(share_s, 2)
    out["share_seconds_is"] = "measured" if len(run) == len(ranges) else f"extrapolated from {len(run)} of {len(ranges)} ranges (targets are uniform: reads in random genome order)"
    stage = {k: sum(p[k] for p in steady) / max(tgt, 1) * (t_hi - t_lo) for k in ("scan_s", "sort_s", "walk_s")}
    out["share_stage_seconds"] = {k: round(v, 2) for k, v in stage.items()}
    # the 8-GPU run: every rank does one share (same size, same density) after exchanging packed reads and probe entries once
    xgmi = 7 * 50e9                                                  # B/s into one GPU over its 7 links at ~50 GB/s achieved each (all-gather: every link carries one peer's shard)
    exch = (n * pk * (a.ranks - 1) / a.ranks) / xgmi + (n * t2 * 8 * (a.ranks - 1) / a.ranks) / xgmi
    out["implied_8gpu_seconds"] = {"per_rank_share": round(share_s, 2), "table_build": round(t_table, 2),
                                   "all_gather_estimate": round(exch, 2), "total": round(share_s + t_table + exch, 2),
                                   "note": "all_gather_estimate = (packed reads + probe entries) x 7/8 over 7 xGMI links at an assumed 50 GB/s each: not measured (one GPU here)"}
    out["pairs_per_s_implied_8gpu"] = round(tot["n_pairs"] / max(sum(p["seconds"] for p in per), 1e-9) * a.ranks, 1) if per else None
    table.close()
    S.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
