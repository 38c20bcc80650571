"""Worker of tests/test_distributed_cpu.py::test_overlap_exchange_gloo: one rank of the all-vs-all exchange over gloo.
There is no GPU in this test: what the HIP kernels do on the GPU box (packing, probe emission, the target walk) is done
here by the host codec and the CPU oracle; the protocol code (shard_range, all_gather_packed, all_gather_entries and the
merge of the per-rank results) is the product's."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from oraclelib import Oracle                                       # noqa: E402
from pacbioassembly_amd import distributed as pd, engine as eng    # noqa: E402

N_READS, READ_LEN, TRIALS = 26, 1300, 32


def pack_shard(texts):
    """What pba_seqs_from_text + pba_seqs_export hand over: 2-bit payloads, every sequence 16-byte aligned."""
    blob, offs = bytearray(), []
    for t in texts:
        offs.append(len(blob))
        blob += eng.text2bin(t)[4:]
        blob += b"\0" * (-len(blob) % 16)
    return np.frombuffer(bytes(blob), np.uint8).copy(), np.array(offs, np.uint64), np.array([len(t) for t in texts], np.uint32)


def probe_entries(texts, q_lo, mask):
    """k_probe_emit on the host: key << 32 | (query * 2*trials + 2j + backward), zero keys dropped (overlap.h)."""
    out, t2 = [], 2 * TRIALS
    for q, t in enumerate(texts, start=q_lo):
        for j in range(TRIALS):
            for back in (0, 1):
                pos = len(t) - j - 16 if back else j
                if pos < 0 or pos + 16 > len(t):
                    continue
                key = eng.encode(t[pos:pos + 16]) & mask
                if key:
                    out.append((key << 32) | (q * t2 + 2 * j + back))
    return np.array(out, np.uint64)


def composition(orc, texts, mask, t_lo, t_hi):
    """The all-vs-all answer for targets [t_lo, t_hi): the oracle's locked round once per target (intended seed_at)."""
    file = b"".join(eng.text2bin(t) for t in texts)
    rec_offs = np.cumsum([0] + [4 + (len(t) + 3) // 4 for t in texts[:-1]]).astype(np.uint64)
    out = []
    for t in range(t_lo, t_hi):
        rows = orc.spaced_round(texts[t], mask, 0.30, file, rec_offs, TRIALS, 64, buggy=False, nthreads=2)
        out += [(t, q, int(rows["j"][q]), int(rows["dir"][q]), int(rows["ref_pos"][q]), int(rows["cost"][q]))
                for q in range(len(texts)) if q != t and rows["found"][q]]
    return out


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    mask = eng.mask_from_pattern("111*11*11*1*1111")
    g = eng.synth_genome(71, 6000)
    lo, hi = pd.shard_range(N_READS, rank, world)
    # every rank generates only its shard (the range form of the generator) ...
    text, offs = eng.synth_reads_range(72, g, lo, hi, READ_LEN)
    mine = [text[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(hi - lo)]
    if rank == 0:
        mine[3] = mine[3][:700]                                    # ragged shards: different byte counts per rank
    # ... and the whole set exists here only to check the exchange against
    full_text, full_offs, _ = eng.synth_reads(72, g, N_READS, READ_LEN)
    full = [full_text[int(full_offs[i]):int(full_offs[i + 1])].tobytes() for i in range(N_READS)]
    full[3] = full[3][:700]
    assert mine == full[lo:hi]

    # 1. packed shards -> all-gather -> every rank holds every read, in global order
    packed, poffs, plens = pack_shard(mine)
    allp, all_offs, all_lens = pd.all_gather_packed(torch.from_numpy(packed), poffs, plens)
    buf = allp.numpy()
    assert all_lens.tolist() == [len(t) for t in full] and all_offs.size == N_READS
    for i, t in enumerate(full):
        rec = np.frombuffer(np.uint32(len(t)).tobytes() + buf[int(all_offs[i]):int(all_offs[i]) + (len(t) + 3) // 4].tobytes(), np.uint8)
        assert eng.bin2text(rec.tobytes()) == t, i

    # 2. probe entries of the rank's queries -> all-gather -> the probe table of the whole set
    ent = probe_entries(mine, lo, mask)
    cap = ((N_READS + world - 1) // world) * 2 * TRIALS + 64
    slot = torch.zeros(cap, dtype=torch.int64)
    slot[:ent.size] = torch.from_numpy(ent.view(np.int64).copy())
    allent, total = pd.all_gather_entries(slot, int(ent.size))
    u = allent.numpy().view(np.uint64)
    u = np.sort(u[u != np.uint64(0xFFFFFFFFFFFFFFFF)])
    want = np.sort(probe_entries(full, 0, mask))
    assert total == want.size and (u == want).all()

    # 3. every rank walks its shard of the targets; the shards' answers concatenate to the single-process answer
    part = composition(orc, full, mask, lo, hi)
    parts = [None] * world
    dist.all_gather_object(parts, part)
    if rank == 0:
        merged = [x for p in parts for x in p]
        assert merged == composition(orc, full, mask, 0, N_READS) and len(merged) > 15
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
