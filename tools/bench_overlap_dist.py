#!/usr/bin/env python3
"""All-vs-all overlap across the GPUs of one node (BASELINE configs 4-5), one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_overlap_dist.py --gpus N

Every rank holds the whole (packed) read set; rank r emits the probe entries of ITS shard of the queries, the padded
entry buffers are all-gathered over RCCL/xGMI (the one collective), and rank r walks ITS shard of the targets against
the full probe table in target ranges.  Strong scaling: the read set is fixed, the targets split.  Rank 0 prints one
JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--coverage", type=float, default=20.0)
    ap.add_argument("--R", type=float, default=0.30)
    ap.add_argument("--trials", type=int, default=32)
    ap.add_argument("--targets-per-call", type=int, default=25000)
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); lr = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    import torch
    import torch.distributed as dist
    from pacbioassembly_amd import Context, distributed as pd, engine as eng
    torch.cuda.set_device(lr)
    os.environ["NCCL_DEBUG"] = os.environ.get("PBA_NCCL_DEBUG", "WARN")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lr))
    ctx = Context(lr)
    L = int(a.reads * a.read_len / a.coverage)
    g = eng.synth_genome(2, L)
    reads, offs, _ = eng.synth_reads(3, g, a.reads, a.read_len, nthreads=16)      # the same set on every rank
    S = ctx.seqs_from_text(reads, offs, strict_acgt=True)
    del reads
    mask = eng.mask_from_pattern("111*11*11*1*1111")
    q_lo, q_hi = pd.shard_range(a.reads, rank, world)
    t_lo, t_hi = pd.shard_range(a.reads, rank, world)
    cap = (a.reads + world - 1) // world * 2 * a.trials + 64                        # probe slots of the largest shard
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    mine = torch.empty(cap, dtype=torch.int64, device="cuda")
    n_mine = ctx.overlap_probes(S, q_lo, q_hi, mask, a.trials, mine.data_ptr(), cap)
    probes, n_probes = pd.all_gather_entries(mine, n_mine)                          # RCCL all-gather: the one exchange
    torch.cuda.synchronize()
    t_x = time.perf_counter() - t0
    n_ov = n_pairs = n_cand = 0
    for lo in range(t_lo, t_hi, a.targets_per_call):
        hi = min(t_hi, lo + a.targets_per_call)
        ov, st = ctx.overlap_all_probes(S, probes.data_ptr(), probes.numel(), mask, a.R, a.trials, 64, lo, hi, cap=(hi - lo) * 400)
        n_ov += st["n_overlaps"]; n_pairs += st["n_pairs"]; n_cand += st["n_candidates"]
    torch.cuda.synchronize(); dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, t_x], dtype=torch.float64, device="cuda"); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    agg = torch.tensor([n_ov, n_pairs, n_cand], dtype=torch.int64, device="cuda"); dist.all_reduce(agg)
    if rank == 0:
        dt, t_x = (float(x) for x in tt.tolist()); n_ov, n_pairs, n_cand = (int(x) for x in agg.tolist())
        print(json.dumps({"workload": f"all-vs-all, {a.reads} x {a.read_len} reads @15%, {a.coverage}x, R={a.R}, targets sharded over {world} GPU(s)",
                          "n_gpus": world, "scaling": "strong", "seconds": round(dt, 3), "exchange_s": round(t_x, 3),
                          "probe_entries": n_probes, "candidates": n_cand, "pairs": n_pairs, "overlaps": n_ov,
                          "pairs_per_s": round(n_pairs / dt, 1), "overlaps_per_s": round(n_ov / dt, 1)}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
