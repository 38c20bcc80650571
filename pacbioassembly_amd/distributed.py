"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in
the CPU tests).  Only the exchange protocol lives here; every kernel is behind the C ABI.

The hot path shards by reads, so the only collective is the seed-index exchange: rank r scans
slice r of the reference's visiting order (pba_index_scan), the flat (key<<32|ordinal) entry lists are
all-gathered, and every rank builds the identical index from the union (pba_index_from_entries).  The
result cannot depend on arrival order because a partition is sorted by the 64-bit entry.

All-vs-all (BASELINE configs 3-4) adds one exchange per read set: every rank packs ITS shard of the reads
and the packed shards are all-gathered (all_gather_packed -> pba_seqs_from_device_packed); per step the probe entries
of the rank's queries are all-gathered the same way as index entries (all_gather_entries) and every rank walks its own
shard of the targets -- no cross-GPU dependency in the align step.
"""
from __future__ import annotations

import numpy as np
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


def all_gather_packed(packed: torch.Tensor, offsets: np.ndarray, lengths: np.ndarray):
    """All-gather of packed read shards.  packed: uint8[nbytes] -- this rank's packed arena (pba_seqs_export) on the device
    the backend moves (cuda for nccl = RCCL, cpu for gloo); offsets[i] / lengths[i]: byte offset inside it and length in
    bases of the rank's i-th sequence.  Shards are padded to the largest one (16-byte granules) for one
    all_gather_into_tensor; returns (uint8[world * stride], offsets u64[n_total] into that buffer, lengths u32[n_total]) with
    the sequences in rank order -- global read id = reads before the rank's shard + local id (shard_range)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = packed.device
    n_mine, nb = int(len(lengths)), int(packed.numel())
    meta = torch.tensor([n_mine, nb], dtype=torch.int64, device=dev)
    metas = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta)
    metas = metas.view(world, 2).cpu().numpy()
    counts, nbytes = metas[:, 0], metas[:, 1]
    stride = (int(nbytes.max()) + 15) // 16 * 16
    mine = torch.zeros(max(stride, 16), dtype=torch.uint8, device=dev)
    mine[:nb] = packed[:nb]
    stride = mine.numel()
    allp = torch.empty(world * stride, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(allp, mine)
    cmax = max(int(counts.max()), 1)
    ol = np.zeros(2 * cmax, np.int64)
    ol[:n_mine] = np.asarray(offsets, np.uint64).astype(np.int64)
    ol[cmax:cmax + n_mine] = np.asarray(lengths, np.uint32).astype(np.int64)
    ol_t = torch.from_numpy(ol).to(dev)
    all_ol = torch.empty(world * 2 * cmax, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_ol, ol_t)
    all_ol = all_ol.view(world, 2, cmax).cpu().numpy()
    offs = np.concatenate([all_ol[r, 0, :counts[r]].astype(np.uint64) + np.uint64(r * stride) for r in range(world)])
    lens = np.concatenate([all_ol[r, 1, :counts[r]].astype(np.uint32) for r in range(world)])
    return allp, offs, lens
