"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in
the CPU tests).  Only the exchange protocol lives here; every kernel is behind the C ABI.

The hot path shards by reads, so the only collective is the seed-index exchange: rank r scans
slice r of the reference's visiting order (pba_index_scan), the flat (key<<32|ordinal) entry lists are
all-gathered, and every rank builds the identical index from the union (pba_index_from_entries).  The
result cannot depend on arrival order because a partition is sorted by the 64-bit entry.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

PAD = -1   # an all-ones u64 entry: never a real entry (key 0xFFFFFFFF with ordinal 0xFFFFFFFF), dropped by the builder


def slice_bounds(n_visited: int, part: int, nparts: int):
    """Ordinal range [lo, hi) of the visiting order that rank `part` scans; mirrors pba_index_scan."""
    return n_visited * part // nparts, n_visited * (part + 1) // nparts


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard of reads owned by `rank`."""
    return n_items * rank // world, n_items * (rank + 1) // world


def slice_capacity(n_visited: int, nparts: int) -> int:
    """Entries a rank's buffer must hold: its slice never yields more entries than positions."""
    return (n_visited + nparts - 1) // nparts + 64


def all_gather_entries(mine: torch.Tensor, n_mine: int):
    """mine: int64[cap] holding n_mine entries.  Pads the tail, all-gathers, returns (int64[world*cap], total)."""
    world = dist.get_world_size()
    cap = mine.numel()
    mine[n_mine:] = PAD
    counts = torch.zeros(world, dtype=torch.int64, device=mine.device)
    counts[dist.get_rank()] = n_mine
    dist.all_reduce(counts)
    allent = torch.empty(cap * world, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(allent, mine)
    return allent, int(counts.sum().item())
