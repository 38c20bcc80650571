// overlap_dist.cpp -- all-vs-all overlap of a synthetic read set across the GPUs of one node from a C++ host: one process
// per GPU, the exchange through include/pba_dist.h (RCCL over xGMI), everything else through include/pba.h.  Plain g++; no
// HIP at the call site.  What pacbioassembly_amd/distributed.py + bench.py --mode overlap do, for a host that is not Python
// (SURVEY 8e): every rank packs ITS shard of the reads, the packed shards are all-gathered once, the probe entries of the
// rank's queries are all-gathered once per pass, and the rank walks its own shard of the targets -- no cross-GPU dependency
// in the align step.
//
//   overlap_dist N_READS READ_LEN [ID_FILE]        with RANK / WORLD_SIZE / LOCAL_RANK in the environment (default: one rank)
// Rank 0 writes the communicator id to ID_FILE, the others wait for it (any side channel does: a launcher's environment, MPI).
// Prints one line per rank: its target shard, candidate pairs, overlaps.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "pba.h"
#include "pba_dist.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int st__ = (call);                                                                       \
        if (st__ != PBA_OK) { fprintf(stderr, "%s: %s (%s)\n", #call, pba_strerror(st__), ctx ? pba_ctx_error(ctx) : ""); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: overlap_dist n_reads read_len [id_file]\n"); return 2; }
    const uint32_t n = (uint32_t)atoi(argv[1]), rl = (uint32_t)atoi(argv[2]);
    const char *id_file = argc > 3 ? argv[3] : nullptr;
    const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0, world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
    const int dev = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank;
    pba_ctx *ctx = nullptr;
    CHECK(pba_ctx_create(dev, &ctx));

    // the communicator
    uint8_t id[PBA_DIST_ID_BYTES];
    if (rank == 0) {
        CHECK(pba_dist_unique_id(id));
        if (world > 1) {
            if (!id_file) { fprintf(stderr, "more than one rank needs an id file\n"); return 2; }
            const std::string tmp = std::string(id_file) + ".tmp";
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 2;
            fclose(f);
            rename(tmp.c_str(), id_file);
        }
    } else {
        FILE *f = nullptr;
        for (int k = 0; k < 6000 && !(f = fopen(id_file, "rb")); ++k) usleep(10000);
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "rank %d: no communicator id in %s\n", rank, id_file); return 2; }
        fclose(f);
    }
    pba_comm *comm = nullptr;
    CHECK(pba_dist_comm_create(ctx, rank, world, id, &comm));

    // this rank's shard of the reads (synthetic: 15 % error, 20 x coverage), packed on its GPU
    uint64_t lo = 0, hi = 0;
    pba_dist_shard(n, rank, world, &lo, &hi);
    const size_t L = (size_t)n * rl / 20;
    std::vector<char> genome(L), text((size_t)(hi - lo) * rl);
    pba_synth_genome(2, genome.data(), L);
    if (pba_synth_reads_range(3, genome.data(), L, (uint32_t)lo, (uint32_t)hi, rl, 0.05, 0.05, 0.05, text.data(), nullptr, 8) != 0) return 1;
    std::vector<uint64_t> offs(hi - lo + 1);
    for (uint64_t i = 0; i <= hi - lo; ++i) offs[i] = i * rl;
    pba_seqs *mine = nullptr, *all = nullptr;
    CHECK(pba_seqs_from_text(ctx, text.data(), offs.data(), (uint32_t)(hi - lo), 1, &mine));

    // exchange 1 (once per read set): the packed shards -> every rank holds every read
    CHECK(pba_dist_gather_reads(comm, mine, &all));
    pba_seqs_destroy(mine);
    // exchange 2 (once per pass): the probe entries of the rank's queries -> every rank builds the same probe table
    const uint32_t mask = pba_mask_from_pattern("111*11*11*1*1111");
    pba_probe_table *table = nullptr;
    CHECK(pba_dist_probe_table(comm, all, (uint32_t)lo, (uint32_t)hi, mask, 32, &table));
    // the rank's shard of the targets: scan, sort, walk -- no other rank involved
    const uint64_t cap = (hi - lo) * 400 + 1;
    std::vector<pba_overlap> out(cap);
    uint64_t n_out = 0;
    pba_overlap_stats st;
    CHECK(pba_overlap_all_table(ctx, all, (uint32_t)lo, (uint32_t)hi, table, 0.30, 64, PBA_KERNEL_AUTO, out.data(), cap, &n_out, &st));
    printf("rank %d of %d: targets [%llu, %llu) of %u reads: %llu candidate pairs, %llu overlaps\n", rank, world, (unsigned long long)lo,
           (unsigned long long)hi, pba_seqs_count(all), (unsigned long long)st.n_pairs, (unsigned long long)st.n_overlaps);
    // the node's totals
    uint64_t tot[2] = {st.n_pairs, st.n_overlaps};
    CHECK(pba_dist_all_reduce_u64(comm, tot, 2, 0));
    if (rank == 0) printf("all ranks: %llu candidate pairs, %llu overlaps\n", (unsigned long long)tot[0], (unsigned long long)tot[1]);
    pba_probe_table_destroy(table);
    pba_seqs_destroy(all);
    pba_dist_comm_destroy(comm);
    pba_ctx_destroy(ctx);
    return 0;
}
