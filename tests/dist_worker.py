"""Worker of tests/test_distributed_cpu.py: one rank of the seed-index exchange over gloo.  The per-slice
scan is done by the CPU oracle here (there is no GPU in this test); the protocol code is the product's."""
import hashlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from oraclelib import Oracle                                  # noqa: E402
from pacbioassembly_amd import distributed as pd, engine as eng   # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    mask = eng.mask_from_pattern("111*11*11*1*1111")
    L = 30000
    g = eng.synth_genome(61, L).tobytes()
    keys, pos, _, _ = orc.index(g, mask, "all")              # locator order: ordinal == position
    lo, hi = pd.slice_bounds(L, rank, world)
    sel = (pos >= lo) & (pos < hi)
    ent = (keys[sel].astype(np.uint64) << np.uint64(32)) | pos[sel].astype(np.uint64)
    rng = np.random.RandomState(rank)
    rng.shuffle(ent)                                          # a scan emits in no particular order
    cap = pd.slice_capacity(L, world)
    mine = torch.full((cap,), 0, dtype=torch.int64)
    mine[:ent.size] = torch.from_numpy(ent.view(np.int64).copy())
    allent, total = pd.all_gather_entries(mine, int(ent.size))
    u = allent.numpy().view(np.uint64)
    u = np.sort(u[u != np.uint64(0xFFFFFFFFFFFFFFFF)])        # what the partition sort does, globally
    assert total == keys.size == u.size, (total, keys.size, u.size)
    gk, gp = (u >> np.uint64(32)).astype(np.uint32), (u & np.uint64(0xFFFFFFFF)).astype(np.int32)
    assert (gk == keys).all() and (gp == pos).all()           # same keys, same per-key hit order as one rank alone
    # reads shard contiguously and cover everything exactly once
    spans = [pd.shard_range(1001, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == 1001 and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    # every rank ends with the identical index
    dig = hashlib.sha256(u.tobytes()).digest()
    t = torch.frombuffer(bytearray(dig), dtype=torch.uint8).clone()
    ts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(ts, t)
    assert all(bool((x == t).all()) for x in ts)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
