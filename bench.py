#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on its named configuration.

  metric   : aligned candidate pairs/sec (whole node), 15 kb PacBio reads @15% error
  workload : BASELINE.json configs[1] -- 100 k synthetic 15 kb reads @15 % error against a 5 Mb synthetic
             genome, seed-hash + banded align on 1 MI355X (the locator.cpp path with R = 0.30, the
             reference's MAXR, because 15 %-error reads do not align at locator's hard-coded 0.15;
             mask 111*11*11*1*1111, 50 probe offsets, reads >= 500 bases) -- SURVEY.md 8d.
  step     : one pass of the hot path over the batch: seed-index build of the genome + the ordered
             first-success locate of every read (probe -> candidate pairs -> banded DP), inputs (packed
             genome and packed reads) already resident in HBM, result rows returned to the host.
  value    : candidate pairs the reference's loop hands to seq_aligner::align (all ranks) / wall time.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One process per GPU.  Reads shard across ranks (every rank owns its own 100 k reads: weak scaling); with
N > 1 each rank scans 1/N of the genome's positions and the seed-index entries are all-gathered over
RCCL/xGMI before every rank builds its lookup structure; the align step has no cross-GPU dependency.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "aligned candidate pairs/sec (whole node), 15 kb PacBio reads @15% error"
HBM_PEAK = 8.0e12            # B/s, MI355X spec (MI355X_MICROARCH.md)
INT_LANE_OPS_PEAK = 256 * 4 * 32 * 2.4e9   # 32-bit integer lane-ops/s: 256 CU x 4 SIMD-32 x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=15_000)
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--R", type=float, default=0.30)
    ap.add_argument("--trials", type=int, default=50)
    ap.add_argument("--kernel", choices=["auto", "rowsweep", "bitvec"], default="auto")
    ap.add_argument("--cpu-sample", type=int, default=512, help="reads timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores available to this process")
    ap.add_argument("--exchange", action="store_true",
                    help="take the multi-GPU seed-index exchange path (scan slice -> RCCL all-gather -> build) even at N=1")
    return ap.parse_args()


def host_cores() -> int:
    """CPU cores this process may really use: the affinity mask capped by the cgroup CPU quota (a 1-GPU box
    exposes every host core in the mask but grants a 16-core share)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, int(os.environ.get("PBA_MAX_HOST_THREADS", "64")))


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import torch
    import torch.distributed as dist
    from pacbioassembly_amd import Context, engine as eng
    from pacbioassembly_amd.engine import PBA_INDEX_ALL

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path to time")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or a.exchange
    if use_dist:
        os.environ["NCCL_DEBUG"] = os.environ.get("PBA_NCCL_DEBUG", "WARN")   # no RCCL banner on stdout: ONE JSON line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    ctx = Context(local_rank)
    kernel = {"auto": eng.PBA_KERNEL_AUTO, "rowsweep": eng.PBA_KERNEL_ROWSWEEP, "bitvec": eng.PBA_KERNEL_BITVEC}[a.kernel]
    mask = eng.mask_from_pattern("111*11*11*1*1111")
    nthreads = a.cpu_threads or host_cores()

    # ---- synthetic inputs: genome seed 2, reads seed 3 (+ rank), 5/5/5 % ins/del/sub
    t0 = time.time()
    genome = eng.synth_genome(2, a.genome)
    reads, offs, _ = eng.synth_reads(3 + 1000 * rank, genome, a.reads, a.read_len, 0.05, 0.05, 0.05, nthreads=nthreads)
    t_gen = time.time() - t0
    cpu_reads = reads[: a.cpu_sample * a.read_len].copy() if rank == 0 else None
    t0 = time.time()
    T = ctx.seqs_from_text(genome, np.array([0, genome.size], np.uint64), strict_acgt=True)
    Rd = ctx.seqs_from_text(reads, offs, strict_acgt=True)        # H2D + 2-bit pack on the GPU: now resident
    t_up = time.time() - t0
    del reads

    def exchange_index():
        """N > 1: scan 1/N of the genome's positions, all-gather the entries over RCCL, build the lookup."""
        from pacbioassembly_amd import distributed as pd
        cap = pd.slice_capacity(a.genome, world)
        mine = torch.empty(cap, dtype=torch.int64, device="cuda")
        n_mine = ctx.index_scan(T, 0, mask, PBA_INDEX_ALL, rank, world, mine.data_ptr(), cap)
        allent, _ = pd.all_gather_entries(mine, n_mine)
        torch.cuda.synchronize()
        return ctx.index_from_entries(allent.data_ptr(), allent.numel(), mask, PBA_INDEX_ALL, a.genome)

    def step():
        ix = exchange_index() if use_dist else ctx.index_build(T, 0, mask, PBA_INDEX_ALL)
        prof_ix = ctx.last_profile()["index_ms"]
        rows, st = ctx.locate(ix, T, 0, Rd, a.R, a.trials, 500, kernel=kernel)
        prof = ctx.last_profile()
        prof["index_ms"] = prof_ix
        ix.close()
        return rows, st, prof

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    profs = []
    for _ in range(a.steps):
        rows, st, prof = step()
        profs.append(prof)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        agg = torch.tensor([st["n_pairs"], st["n_located"], st["n_cells"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(agg)
        pairs, located, cells = (int(x) for x in agg.tolist())
    else:
        pairs, located, cells = st["n_pairs"], st["n_located"], st["n_cells"]

    if rank != 0:
        dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / a.steps
    value = pairs * a.steps / elapsed

    # ---- roofline of the dominant kernel (k_locate, first launch), per launch, this rank
    # algorithmic bytes (score-only align): per candidate pair the packed bases of both windows
    # plus a 24 B descriptor and a 28 B result; a = read from j (<= read_len), b = contig clipped to
    # len_a + max_dst (seq_aligner.h:94-102)
    md = 1 + int(a.read_len * a.R)
    bytes_per_pair = (a.read_len + 3) // 4 + (a.read_len + md + 3) // 4 + 24 + 28
    align_ms = float(np.mean([p["align_ms"] for p in profs]))
    redo_ms = float(np.mean([p["align_redo_ms"] for p in profs]))
    index_ms = float(np.mean([p["index_ms"] for p in profs]))
    algo_bytes = st["n_pairs"] * bytes_per_pair
    achieved = algo_bytes / (align_ms * 1e-3) / 1e9 if align_ms > 0 else 0.0
    traffic, pmc = None, {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")      # per-launch PMC figures from separate --pmc runs
    if os.path.exists(tpath):
        try:
            pmc = json.load(open(tpath))
            traffic = pmc.get("k_locate_hbm_bytes_per_launch")
        except Exception:
            traffic, pmc = None, {}
    roofline = {"bound": "hbm", "kernel": f"k_locate<{profs[-1]['nb_first']}>", "achieved": round(achieved, 3),
                "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved * 1e9 / HBM_PEAK, 6),
                "traffic": traffic, "algorithmic_bytes_per_launch": algo_bytes, "launch_ms": round(align_ms, 3),
                "note": "score-only banded DP keeps its state in registers: HBM is not what bounds it (SURVEY 8d "
                        "expects <<1 %); the binding resource is integer VALU issue, see roofline_valu"}
    gcups = st["n_cells"] / (align_ms * 1e-3) / 1e9 if align_ms > 0 else 0.0
    # VALU issue roofline: wave64 instructions per second against the measured full-rate issue (one every 2.5 cycles
    # per SIMD, tools/ubench_ops.hip); the instruction count comes from the committed PMC run of this same command
    valu_peak = 256 * 4 * 2.4e9 / 2.5
    valu_insts = pmc.get("valu_insts_per_launch")
    valu_rate = valu_insts / (align_ms * 1e-3) if valu_insts and align_ms > 0 else None
    roofline_valu = {"bound": "valu-issue", "achieved": round(valu_rate / 1e9, 1) if valu_rate else None,
                     "peak": round(valu_peak / 1e9, 1), "unit": "G wave-instr/s",
                     "frac": round(valu_rate / valu_peak, 4) if valu_rate else None,
                     "achieved_gcups": round(gcups, 1),
                     "note": "every instruction priced at the full rate (a lower bound on pipe occupancy: ~22 % of the "
                             "step's instructions are half-rate v_addc_co / v_bfe / v_alignbit); gcups = reference-band "
                             "cells per second of the first k_locate launch"}

    # ---- CPU baseline: the faithful oracle on this box's host cores, bounded sample of the same workload
    cpu = None
    if a.cpu_sample > 0 and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oraclelib import Oracle
        orc = Oracle()
        ns = min(a.cpu_sample, a.reads)
        coffs = np.arange(ns + 1, dtype=np.uint64) * np.uint64(a.read_len)
        orc.prefault(nthreads, a.read_len, a.R)      # DP matrices mapped and touched before the clock starts
        t0 = time.perf_counter()
        crow, cst = orc.locator(genome, mask, a.R, cpu_reads[: ns * a.read_len], coffs, a.trials, 500,
                                nthreads=nthreads)
        ct = time.perf_counter() - t0
        same = all((crow[c] == rows[c][:ns]).all() for c in ("found", "j", "pos", "cost", "seglen", "matlen_a",
                                                              "matlen_b", "n_pairs"))
        cpu = {"value": round(cst["n_pairs"] / ct, 3), "unit": "pairs/s", "cores": nthreads, "kind": "port",
               "sample": f"first {ns} reads of the same workload (index build + locate), {ct:.1f} s wall, "
                         f"{cst['n_pairs']} pairs, {cst['n_located']} located, {cst['n_cells'] / ct / 1e9:.2f} GCUPS",
               "gpu_rows_identical_on_sample": bool(same)}
        orc.release()

    out = {
        "metric": METRIC, "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 100k synthetic 15 kb reads @15% error vs 5 Mb genome, "
                               "seed-hash + banded align (locator.cpp path, R=0.30, 50 probe offsets)",
                   "reads_per_gpu": a.reads, "read_len": a.read_len, "genome": a.genome, "R": a.R,
                   "trials": a.trials, "mask": "111*11*11*1*1111", "kernel": a.kernel,
                   "parallelism": f"reads sharded over {world} GPU(s)" + (", seed index all-gathered over RCCL" if use_dist else "")},
        "pairs_per_step": pairs, "located_per_step": located, "successful_pairs_per_s": round(located * a.steps / elapsed, 2),
        "band_gcups": round(cells * a.steps / elapsed / 1e9, 1),
        "kernel_ms": {"index_build": round(index_ms, 3), "locate_first": round(align_ms, 3),
                      "locate_redo": round(redo_ms, 3), "n_redo_reads": profs[-1]["n_redo"]},
        "setup_s": {"generate": round(t_gen, 2), "upload_and_pack": round(t_up, 2)},
        "roofline": roofline, "roofline_valu": roofline_valu, "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
