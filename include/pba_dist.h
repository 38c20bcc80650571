/* pba_dist.h -- the multi-GPU exchange of the overlap engine as C entry points over RCCL.
 *
 * One process per GPU.  The hot path shards by reads; the only cross-GPU steps are exchanges of SMALL objects before the
 * align step, and they are all all-gathers over xGMI:
 *   - locator mode (locator.cpp:62-66 built by N ranks): the seed-index entries of each rank's slice of the target
 *   - all-vs-all   (configs 3-4): the packed read shards once per read set, the probe entries once per pass
 * The reference is a single-process program and has no counterpart of this header; it exists so that a C / C++ host can do
 * what pacbioassembly_amd/distributed.py does with torch.distributed (same protocol, same padding, same results).
 *
 * Library: pacbioassembly_amd/lib/libpba_dist.so (links libpba.so and librccl.so).  Status codes are pba_status (pba.h);
 * PBA_E_HIP also stands for a failed RCCL call, with the text in pba_ctx_error().
 *
 * Bring-up: rank 0 calls pba_dist_unique_id and hands the 128 bytes to the other ranks by whatever side channel the host
 * program has (a file, MPI, its launcher's environment); then every rank calls pba_dist_comm_create with the same bytes.
 */
#ifndef PBA_DIST_H
#define PBA_DIST_H

#include "pba.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PBA_DIST_ID_BYTES 128

typedef struct pba_comm pba_comm;

int pba_dist_unique_id(uint8_t id[PBA_DIST_ID_BYTES]);
/* the communicator of `world` ranks on the ctx's device and stream; collective */
int pba_dist_comm_create(pba_ctx *ctx, int rank, int world, const uint8_t id[PBA_DIST_ID_BYTES], pba_comm **out);
void pba_dist_comm_destroy(pba_comm *c);
int pba_dist_rank(const pba_comm *c);
int pba_dist_world(const pba_comm *c);

/* contiguous shard [lo, hi) of n items owned by `rank` of `world` (reads of a rank; the same split everywhere) */
void pba_dist_shard(uint64_t n, int rank, int world, uint64_t *lo, uint64_t *hi);

/* all-gather of equally sized device buffers: d_all receives world * n_bytes, rank order */
int pba_dist_all_gather(pba_comm *c, const void *d_mine, uint64_t n_bytes, void *d_all);
/* element-wise sum / max of n u64 values across the ranks, in place, on the host (small metadata) */
int pba_dist_all_reduce_u64(pba_comm *c, uint64_t *values, uint32_t n, int take_max);

/* The seed index of target sequence `seq` built by all ranks together (pba_index_build's multi-GPU form): this rank scans its
 * slice of the reference's visiting order, the entry lists are all-gathered, every rank builds the identical index. */
int pba_dist_index_build(pba_comm *c, const pba_seqs *target, uint32_t seq, uint32_t mask, int mode, pba_index **out);

/* The read set of an all-vs-all run from the ranks' shards: every rank hands in ITS packed shard (reads
 * [lo, hi) of the set, pba_dist_shard), gets back the set of all reads (global id = shard offset + local id), resident on
 * its GPU.  Once per read set. */
int pba_dist_gather_reads(pba_comm *c, const pba_seqs *mine, pba_seqs **all);

/* The probe table of all reads from the ranks' probe entries: this rank emits the probes of its own queries [q_lo, q_hi) of
 * `reads` (the gathered set), the entry buffers are all-gathered, every rank builds the same table.  Once per pass; the
 * rank then walks its own shard of the targets with pba_overlap_all_table -- no cross-GPU dependency in the align step. */
int pba_dist_probe_table(pba_comm *c, const pba_seqs *reads, uint32_t q_lo, uint32_t q_hi, uint32_t mask, int max_trial,
                         pba_probe_table **out);

#ifdef __cplusplus
}
#endif
#endif
