// pba_dist.hip -- the multi-GPU exchange as C entry points over RCCL (include/pba_dist.h).  Host code only: every
// kernel is behind libpba.so's ABI; this library adds the collectives between its calls -- the same protocol, padding and
// results as pacbioassembly_amd/distributed.py, for hosts that are not Python.  One process per GPU, one communicator per
// ctx, everything on the ctx's stream.
#include <rccl/rccl.h>

#include "pba_host.h"
#include "pba_dist.h"

struct pba_comm {
    pba_ctx *ctx;
    ncclComm_t comm;
    int rank, world;
};

#define NCCLCHK(call)                                                                         \
    do {                                                                                      \
        ncclResult_t r__ = (call);                                                            \
        if (r__ != ncclSuccess) {                                                             \
            snprintf(ctx->err, sizeof ctx->err, "%s: %s", #call, ncclGetErrorString(r__));    \
            return PBA_E_HIP;                                                                 \
        }                                                                                     \
    } while (0)

extern "C" {

int pba_dist_unique_id(uint8_t id[PBA_DIST_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) <= PBA_DIST_ID_BYTES, "ncclUniqueId outgrew PBA_DIST_ID_BYTES");
    if (!id) return PBA_E_INVALID;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return PBA_E_HIP;
    memset(id, 0, PBA_DIST_ID_BYTES);
    memcpy(id, &u, sizeof u);
    return PBA_OK;
}

int pba_dist_comm_create(pba_ctx *ctx, int rank, int world, const uint8_t id[PBA_DIST_ID_BYTES], pba_comm **out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    pba_comm *c = new (std::nothrow) pba_comm();
    if (!c) PBA_FAIL(PBA_E_NOMEM, "pba_comm");
    c->ctx = ctx; c->rank = rank; c->world = world; c->comm = nullptr;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    const ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        snprintf(ctx->err, sizeof ctx->err, "ncclCommInitRank: %s", ncclGetErrorString(r));
        delete c;
        return PBA_E_HIP;
    }
    *out = c;
    return PBA_OK;
}

void pba_dist_comm_destroy(pba_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
}

int pba_dist_rank(const pba_comm *c) { return c ? c->rank : -1; }
int pba_dist_world(const pba_comm *c) { return c ? c->world : 0; }

void pba_dist_shard(uint64_t n, int rank, int world, uint64_t *lo, uint64_t *hi) {
    if (lo) *lo = world > 0 ? n * (uint64_t)rank / (uint64_t)world : 0;
    if (hi) *hi = world > 0 ? n * ((uint64_t)rank + 1) / (uint64_t)world : n;
}

int pba_dist_all_gather(pba_comm *c, const void *d_mine, uint64_t n_bytes, void *d_all) {
    if (!c || (n_bytes && (!d_mine || !d_all))) return PBA_E_INVALID;
    pba_ctx *ctx = c->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    if (n_bytes) NCCLCHK(ncclAllGather(d_mine, d_all, (size_t)n_bytes, ncclUint8, c->comm, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

int pba_dist_all_reduce_u64(pba_comm *c, uint64_t *values, uint32_t n, int take_max) {
    if (!c || (n && !values)) return PBA_E_INVALID;
    if (!n) return PBA_OK;
    pba_ctx *ctx = c->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf d;
    HIPCHK(hipMalloc(&d.p, sizeof(uint64_t) * n));
    HIPCHK(hipMemcpyAsync(d.p, values, sizeof(uint64_t) * n, hipMemcpyHostToDevice, ctx->stream));
    NCCLCHK(ncclAllReduce(d.p, d.p, n, ncclUint64, take_max ? ncclMax : ncclSum, c->comm, ctx->stream));
    HIPCHK(hipMemcpyAsync(values, d.p, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

// entries of equal capacity from every rank (unused slots all-ones: never a real entry, dropped by the builders)
static int gather_entries(pba_comm *c, DevBuf &mine, uint64_t cap, DevBuf &all) {
    pba_ctx *ctx = c->ctx;
    HIPCHK(hipMalloc(&all.p, sizeof(uint64_t) * cap * (uint64_t)c->world));
    return pba_dist_all_gather(c, mine.p, sizeof(uint64_t) * cap, all.p);
}

int pba_dist_index_build(pba_comm *c, const pba_seqs *target, uint32_t seq, uint32_t mask, int mode, pba_index **out) {
    if (!c || !target || !out || seq >= target->n) return PBA_E_INVALID;
    pba_ctx *ctx = c->ctx;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t len = target->h_len[seq];
    // a rank's slice of the visiting order never yields more entries than positions (distributed.py: slice_capacity)
    const uint64_t cap = ((uint64_t)len + c->world - 1) / c->world + 64;
    DevBuf mine, all;
    HIPCHK(hipMalloc(&mine.p, sizeof(uint64_t) * cap));
    HIPCHK(memset_big(mine.p, 0xFF, sizeof(uint64_t) * cap, ctx->stream));
    uint64_t n_mine = 0;
    int st = pba_index_scan(ctx, target, seq, mask, mode, (uint32_t)c->rank, (uint32_t)c->world, mine.p, cap, &n_mine);
    if (st != PBA_OK) return st;
    st = gather_entries(c, mine, cap, all);
    if (st != PBA_OK) return st;
    return pba_index_from_entries(ctx, all.p, cap * (uint64_t)c->world, mask, mode, len, out);
}

int pba_dist_gather_reads(pba_comm *c, const pba_seqs *mine, pba_seqs **out) {
    if (!c || !mine || !out) return PBA_E_INVALID;
    pba_ctx *ctx = c->ctx;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    const int W = c->world;
    // how much every rank brings: reads, packed bytes, whether it holds bytes outside ACGT
    std::vector<uint64_t> meta(3 * (size_t)W, 0);
    meta[3 * c->rank] = mine->n; meta[3 * c->rank + 1] = mine->packed_bytes; meta[3 * c->rank + 2] = mine->non_acgt ? 1 : 0;
    int st = pba_dist_all_reduce_u64(c, meta.data(), (uint32_t)meta.size(), 0);
    if (st != PBA_OK) return st;
    uint64_t n_total = 0, nb_max = 0, c_max = 1, any_non = 0;
    for (int r = 0; r < W; ++r) {
        n_total += meta[3 * r]; nb_max = std::max(nb_max, meta[3 * r + 1]); c_max = std::max(c_max, meta[3 * r]);
        any_non |= meta[3 * r + 2];
    }
    if (n_total >= 0xFFFFFFFFull) PBA_FAIL(PBA_E_TOOLONG, "pba_dist_gather_reads: 2^32 reads or more");
    const uint64_t stride = std::max<uint64_t>(16, (nb_max + 15) / 16 * 16);        // shards padded to the largest, 16-byte granules
    DevBuf d_mine, d_all, d_ol, d_all_ol;
    HIPCHK(hipMalloc(&d_mine.p, stride));
    HIPCHK(memset_big(d_mine.p, 0, stride, ctx->stream));
    std::vector<uint64_t> offs(mine->n + 1, 0);
    st = pba_seqs_export(ctx, mine, d_mine.p, stride, offs.data());
    if (st != PBA_OK) return st;
    HIPCHK(hipMalloc(&d_all.p, stride * (uint64_t)W));
    st = pba_dist_all_gather(c, d_mine.p, stride, d_all.p);
    if (st != PBA_OK) return st;
    d_mine.reset();
    // offsets and lengths of every rank's reads, padded to the largest count
    std::vector<uint64_t> ol(2 * c_max, 0), all_ol(2 * c_max * (uint64_t)W, 0);
    for (uint32_t i = 0; i < mine->n; ++i) { ol[i] = offs[i]; ol[c_max + i] = mine->h_len[i]; }
    HIPCHK(hipMalloc(&d_ol.p, sizeof(uint64_t) * 2 * c_max));
    HIPCHK(hipMalloc(&d_all_ol.p, sizeof(uint64_t) * 2 * c_max * (uint64_t)W));
    HIPCHK(hipMemcpyAsync(d_ol.p, ol.data(), sizeof(uint64_t) * 2 * c_max, hipMemcpyHostToDevice, ctx->stream));
    st = pba_dist_all_gather(c, d_ol.p, sizeof(uint64_t) * 2 * c_max, d_all_ol.p);
    if (st != PBA_OK) return st;
    HIPCHK(hipMemcpyAsync(all_ol.data(), d_all_ol.p, sizeof(uint64_t) * 2 * c_max * (uint64_t)W, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> g_off(n_total + 1, 0);
    std::vector<uint32_t> g_len(n_total + 1, 0);
    uint64_t k = 0;
    for (int r = 0; r < W; ++r)                                                       // rank order: global id = reads before the shard + local id
        for (uint64_t i = 0; i < meta[3 * r]; ++i, ++k) {
            g_off[k] = all_ol[2 * c_max * r + i] + (uint64_t)r * stride;
            g_len[k] = (uint32_t)all_ol[2 * c_max * r + c_max + i];
        }
    return pba_seqs_from_device_packed(ctx, d_all.p, stride * (uint64_t)W, g_off.data(), g_len.data(), (uint32_t)n_total, any_non ? 1 : 0, out);
}

int pba_dist_probe_table(pba_comm *c, const pba_seqs *reads, uint32_t q_lo, uint32_t q_hi, uint32_t mask, int max_trial,
                         pba_probe_table **out) {
    if (!c || !reads || !out || q_lo > q_hi || q_hi > reads->n || max_trial < 1) return PBA_E_INVALID;
    pba_ctx *ctx = c->ctx;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t cap = (uint64_t)(q_hi - q_lo) * 2u * (uint64_t)max_trial + 64;          // probe slots of the largest shard
    int st = pba_dist_all_reduce_u64(c, &cap, 1, 1);
    if (st != PBA_OK) return st;
    DevBuf mine, all;
    HIPCHK(hipMalloc(&mine.p, sizeof(uint64_t) * cap));
    HIPCHK(memset_big(mine.p, 0xFF, sizeof(uint64_t) * cap, ctx->stream));
    uint64_t n_mine = 0;
    st = pba_overlap_probes(ctx, reads, q_lo, q_hi, mask, max_trial, mine.p, cap, &n_mine);
    if (st != PBA_OK) return st;
    st = gather_entries(c, mine, cap, all);
    if (st != PBA_OK) return st;
    return pba_probe_table_create(ctx, all.p, cap * (uint64_t)c->world, mask, max_trial, out);
}

}  // extern "C"
