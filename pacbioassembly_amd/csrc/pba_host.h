// pba_host.h -- host-side objects and helpers shared by the translation units of libpba.so (pba_core.hip: context,
// sequence sets, seed index; pba_align.hip: explicit pairs and edit scripts; pba_drivers.hip: the reference's ordered
// first-success loops; pba_overlap.hip: all-vs-all; pba_cons.hip: consensus voting).  Internal: nothing here is part of
// the C ABI (include/pba.h).
#ifndef PBA_HOST_H
#define PBA_HOST_H

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <new>
#include <vector>

#include "align_common.h"
#include "consensus.h"
#include "dev_common.h"
#include "pba.h"
#include "seed_index.h"

#define PBA_INTERNAL __attribute__((visibility("hidden")))
// a kernel that takes more than the default 64 KB of dynamic LDS; every translation unit does this once for the kernels
// it launches, once per ctx (tu_attrs(ctx) in each .hip: the attribute is per device, and a ctx is bound to one)
// (128 KB of dynamic LDS: the largest LDS sort and the largest band row, with room for a kernel's static arrays; a refused
// attribute must not linger as the thread's last error)
#define PBA_BIG_LDS(kernel)                                                                                            \
    do {                                                                                                               \
        if (hipFuncSetAttribute((const void *)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) \
            (void)hipGetLastError();                                                                                   \
    } while (0)

// ---------------------------------------------------------------------------------------------
// host-side objects
// ---------------------------------------------------------------------------------------------
struct pba_ctx {
    int device;
    hipStream_t own_stream, stream;
    hipDeviceProp_t prop;
    hipEvent_t ev[6];        // index begin/end, align begin/end, redo begin/end
    hipEvent_t ev_aux;       // "the small D2H copy queued before the last kernel has landed" (index build: sizes before the last scatter)
    uint32_t *d_queue;       // work-queue counters of the persistent aligning kernels (one per launch in flight)
    void *d_scratch;         // parent-bit scratch of the trace / vote kernels, kept between calls (tens of GB: mapping
    size_t scratch_bytes;    // it anew on every call cost seconds); grown on demand, freed with the ctx
    // Work buffers of the drivers, kept between calls for the same reason (hipMalloc / hipFree of the 11 GB candidate
    // array of a million-read target range cost more than the kernels that fill it; the 4 MB of per-read rows of a
    // locate step cost 1 ms of a 50 ms step): grown on demand (pool_reserve), released by pba_ctx_trim or with the ctx.
    struct { void *p; size_t cap; } pool[24];
    // pinned host staging of the one-pair text entry points (pba_align_text*: one H2D and one D2H copy per call)
    void *h_stage;
    size_t h_stage_cap;
    uint32_t attr_done;      // translation units whose big-LDS kernel attributes are set on this ctx's device (tu_attrs)
    // the entry / offset arrays of the index destroyed last, for the next build (a step of the locate loop builds and drops
    // one index: the hipFree / hipMalloc pair of its 40 MB cost 0.2 ms of a 48 ms step)
    struct { void *ent; size_t ent_cap; void *off; size_t off_cap; void *ent2; size_t ent2_cap; } ix_cache;
    pba_profile prof;
    char err[512];
};

static const size_t kPlaneSlack = 64;           // zero words before the first and after the last sequence of a bit plane
struct pba_seqs {
    pba_ctx *ctx;
    uint32_t n, max_len;
    uint64_t packed_bytes;   // packed payload resident in HBM (incl. alignment padding)
    bool non_acgt;           // some byte outside ACGT was packed as code 3 (C2I): the packed DP would match it against T
    uint8_t *d_alloc;        // allocation; d_packed = d_alloc + kSlack
    uint8_t *d_packed;
    uint64_t *d_off;
    uint32_t *d_len;
    uint32_t *d_planes;      // allocation of the two bit planes, interleaved word by word, kPlaneSlack zero word pairs around them
    uint64_t *d_poff;        // word offset of every sequence inside a plane
    uint64_t plane_words;    // words of one plane incl. its slack
    std::vector<uint64_t> h_off;
    std::vector<uint32_t> h_len;
    SeqSetDev dev() const { return SeqSetDev{d_packed, d_off, d_len, d_planes + 2 * kPlaneSlack, d_poff}; }
};

struct pba_index {
    pba_ctx *ctx;
    size_t ent_cap, off_cap;     // bytes allocated behind d_ent / d_part_off
    uint32_t mask, seq_len, visited, nhead;
    int32_t tail_top;
    int mode, logP;
    uint64_t n_entries;
    uint64_t *d_ent;
    uint32_t *d_part_off;
    IndexDev dev() const { return IndexDev{d_ent, d_part_off, logP, mask, nhead, tail_top}; }
};

static inline int ctx_fail(pba_ctx *ctx, int st, const char *what, hipError_t e) {
    if (ctx)
        snprintf(ctx->err, sizeof ctx->err, "%s: %s", what, e == hipSuccess ? pba_strerror(st) : hipGetErrorString(e));
    return st;
}
#define HIPCHK(call)                                                        \
    do {                                                                    \
        hipError_t e__ = (call);                                            \
        if (e__ != hipSuccess) return ctx_fail(ctx, PBA_E_HIP, #call, e__); \
    } while (0)
#define PBA_FAIL(st, what) return ctx_fail(ctx, (st), (what), hipSuccess)

// engine limits
static const int kMaxSeqLen = 65000;            // u16 DP costs: D(i,j) <= max(i,j) < 65535
static const int kRowSweepLdsCap = 96 * 1024;   // LDS bytes one wavefront may take for its band row
static const size_t kSlack = 1024;              // readable bytes before the first and after the last packed byte
                                                // (the bit-vector kernel streams a few hundred bases past an accessor)

// pool slots
enum { POOL_OVL_CAND = 0, POOL_OVL_TMP, POOL_OVL_ITEMS, POOL_OVL_REDO, POOL_OVL_REDO_IN, POOL_OVL_OUT, POOL_OVL_SMALL,
       POOL_LOC_ROWS, POOL_LOC_AUX, POOL_LOC_IDS, POOL_IX_OFFS, POOL_IX_WORK, POOL_OVL_BLOOM, POOL_OVL_ENDS,
       POOL_TXT_IN, POOL_TXT_OUT, POOL_TXT_PAR, POOL_TXT_CST };
// a buffer of at least `bytes` in pool slot `slot` (contents undefined); grows by reallocation with 1/8 headroom
static inline int pool_reserve(pba_ctx *ctx, int slot, size_t bytes, void **out) {
    if (ctx->pool[slot].cap < bytes) {
        if (ctx->pool[slot].p) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(ctx->pool[slot].p);
            ctx->pool[slot].p = nullptr; ctx->pool[slot].cap = 0;
        }
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&ctx->pool[slot].p, want);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&ctx->pool[slot].p, bytes); if (e == hipSuccess) ctx->pool[slot].cap = bytes; }
        else ctx->pool[slot].cap = want;
        if (e != hipSuccess) { ctx->pool[slot].p = nullptr; return ctx_fail(ctx, PBA_E_NOMEM, "work buffer", e); }
    }
    *out = ctx->pool[slot].p;
    return PBA_OK;
}
#define POOL(slot, bytes, ptr)                                                     \
    do {                                                                           \
        void *p__ = nullptr;                                                       \
        int st__ = pool_reserve(ctx, (slot), (bytes), &p__);                       \
        if (st__ != PBA_OK) return st__;                                           \
        (ptr) = (decltype(ptr))p__;                                                \
    } while (0)

// a pooled buffer seen through the same face as a DevBuf (not owned: the ctx keeps it)
struct BufRef {
    void *p = nullptr;
    template <class T> T *as() const { return (T *)p; }
};

// A device-to-device copy of any size: in pieces of 1 GiB.  (One hipMemcpyAsync of 37.6 GB -- the packed reads of BASELINE
// configs[4] -- copied only the size modulo 2^32 on ROCm 7.2: tools/rehearse_config4.py found the reads beyond the first
// 3.2 GB missing; tests/test_gpu_parity.py: test_packed_set_beyond_4_gib.)
static inline hipError_t copy_d2d(void *dst, const void *src, size_t bytes, hipStream_t stream) {
    const size_t piece = (size_t)1 << 30;
    for (size_t at = 0; at < bytes; at += piece) {
        const hipError_t e = hipMemcpyAsync((uint8_t *)dst + at, (const uint8_t *)src + at, std::min(piece, bytes - at), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
// ... and a memset likewise
static inline hipError_t memset_big(void *dst, int v, size_t bytes, hipStream_t stream) {
    const size_t piece = (size_t)1 << 30;
    for (size_t at = 0; at < bytes; at += piece) {
        const hipError_t e = hipMemsetAsync((uint8_t *)dst + at, v, std::min(piece, bytes - at), stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Workgroups for a kernel that walks n elements with a grid-stride loop: one element per thread up to 2^30 threads.  (A
// launch's global size -- workgroups x threads -- is a 32-bit number; a launch asking for more does not fail, it runs a part.)
static inline uint32_t elem_grid(uint64_t n, uint32_t threads) {
    const uint64_t blocks = (n + threads - 1) / threads, cap = ((uint64_t)1 << 30) / threads;
    return (uint32_t)std::max<uint64_t>(1, std::min(blocks, cap));
}

// the ctx's pinned host staging buffer, at least `bytes` (one H2D / D2H copy per call of the one-pair text entry points; small
// results a host decision waits for)
static inline int stage_reserve(pba_ctx *ctx, size_t bytes) {
    if (ctx->h_stage_cap >= bytes) return PBA_OK;
    if (ctx->h_stage) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->h_stage); ctx->h_stage = nullptr; ctx->h_stage_cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&ctx->h_stage, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError(); ctx->h_stage = nullptr;
        return ctx_fail(ctx, PBA_E_NOMEM, "pinned staging buffer", hipSuccess);
    }
    ctx->h_stage_cap = want;
    return PBA_OK;
}

// RAII for temporaries so early returns do not leak device memory
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <class T> T *as() const { return (T *)p; }
    void reset() { if (p) (void)hipFree(p); p = nullptr; }
};

// ---------------------------------------------------------------------------------------------
// host API: alignment
// ---------------------------------------------------------------------------------------------
static inline int max_dst_of(int la, int lb, double R) {      // seq_aligner.h:94-102
    return 1 + (int)((lb >= la ? la : lb) * R);
}

// Launch plan for a batch whose widest band is max_dst_max.
struct Plan {
    AlignCfg cfg;
    size_t lds;
    int nb1;     // first launch: 0 = row sweep, else bit-vector array with nb1 blocks per lane (narrow band)
    int nb2;     // second launch (uncertified pairs only): bit-vector array at the reference band
};

static inline int make_plan(pba_ctx *ctx, double R, int maxn, int maxm, int kernel, int max_dst_max, Plan *pl) {
    if (!(R > 0.0) || !(R < 1.0)) PBA_FAIL(PBA_E_INVALID, "R must be in (0,1)");
    if (kernel != PBA_KERNEL_AUTO && kernel != PBA_KERNEL_ROWSWEEP && kernel != PBA_KERNEL_BITVEC)
        PBA_FAIL(PBA_E_INVALID, "unknown kernel");
    const bool bv = kernel != PBA_KERNEL_ROWSWEEP && bitvec_supports(max_dst_max);
    if (kernel == PBA_KERNEL_BITVEC && !bv) PBA_FAIL(PBA_E_TOOLONG, "band too wide for the bit-vector kernel");
    // the bit-vector kernel needs LDS only for its m <= 10 corner (a 23-cell row at most); the row sweep
    // needs the whole band row
    const int nb_hi = bv ? bv_nb_for_span(bv_full_wl(max_dst_max) + max_dst_max) : 0;     // the most blocks per lane a launch of this plan uses
    const long long W = bv ? (long long)PBA_BV_FIN_WORDS(nb_hi) * 2 : 2ll * max_dst_max + 1;   // (u16 cells: the fin words of bitvec_pass)
    const long long bytes = ((W * 2 + 15) / 16) * 16;
    if (bytes > kRowSweepLdsCap) PBA_FAIL(PBA_E_TOOLONG, "band row does not fit the per-wavefront LDS budget");
    pl->cfg.R = R; pl->cfg.maxn = maxn; pl->cfg.maxm = maxm; pl->cfg.full_band = 0;
    pl->cfg.row_cap = (int)(bytes / 2);
    pl->lds = (size_t)bytes;
    pl->nb1 = bv ? bv_nb_for_span(bv_first_wl(max_dst_max) + bv_first_w(max_dst_max)) : 0;
    pl->nb2 = bv ? bv_nb_for_span(bv_full_wl(max_dst_max) + max_dst_max) : 0;
    return PBA_OK;
}

// Workgroups for a persistent launch: enough to fill every CU (up to 8 waves per SIMD, as many as the LDS
// allows), never more than there are work items.  Any residency works: the queue needs no co-residency.
static inline uint32_t persistent_grid(const pba_ctx *ctx, uint32_t n_items, int waves_per_wg, size_t lds_per_wave) {
    const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
    uint32_t wg_per_cu = 32u / (uint32_t)waves_per_wg;
    const size_t lds_wg = lds_per_wave * (size_t)waves_per_wg;
    if (lds_wg) wg_per_cu = std::min<uint32_t>(wg_per_cu, (uint32_t)std::max<size_t>(1, (160 * 1024) / lds_wg));
    const uint32_t need = (n_items + (uint32_t)waves_per_wg - 1) / (uint32_t)waves_per_wg;
    return std::max(1u, std::min(need, cus * wg_per_cu));
}

static inline void prof_finish(pba_ctx *ctx) {      // all launches of the call have completed (stream synchronised)
    (void)hipEventElapsedTime(&ctx->prof.align_ms, ctx->ev[2], ctx->ev[3]);
    if (ctx->prof.n_redo) (void)hipEventElapsedTime(&ctx->prof.align_redo_ms, ctx->ev[4], ctx->ev[5]);
}

static inline bool pair_ok(const pba_seqs *S, uint32_t seq, int pos, int len, bool backward) {
    if (seq >= S->n || len < 0 || len > kMaxSeqLen || pos < 0) return false;
    const long long L = S->h_len[seq];
    if (len == 0) return pos <= L;
    return backward ? (pos < L && pos - (len - 1) >= 0) : ((long long)pos + len <= L);
}

// the five move masks of compress(x, mask) (Hacker's Delight 7-4): gathers the bits of x under `mask` into a number
static inline void compress_masks(uint32_t mask, uint32_t mv[5]) {
    uint32_t m = mask, mk = ~m << 1;
    for (int i = 0; i < 5; ++i) {
        uint32_t mp = mk ^ (mk << 1);
        mp ^= mp << 2; mp ^= mp << 4; mp ^= mp << 8; mp ^= mp << 16;
        const uint32_t mvi = mp & m;
        mv[i] = mvi;
        m = (m ^ mvi) | (mvi >> (1 << i));
        mk &= ~mp;
    }
}
// how segments are spread over buckets by k_seg_sort (seed_index.h)
static inline SegBkt seg_bkt_range() { SegBkt b; memset(&b, 0, sizeof b); b.mode = 0; return b; }
static inline SegBkt seg_bkt_key(uint32_t mask) {
    SegBkt b; memset(&b, 0, sizeof b);
    b.mode = 1; b.mask = mask; b.care = __builtin_popcount(mask);
    compress_masks(mask, b.mv);
    return b;
}
// n_seg segments of at most max_n entries each (larger ones, and those with a bucket beyond 256 entries, are listed in
// oversize[1 ..] for the caller's global pass; oversize[0] must be zero on entry)
static inline void launch_seg_sort(pba_ctx *ctx, const uint64_t *src, uint64_t *dst, const uint32_t *seg_off, const SegRef *segs,
                                   uint64_t n_seg, uint32_t max_n, const SegBkt &bk, uint32_t *oversize, uint32_t oversize_cap) {
    // (a launch's global size -- workgroups x threads -- is a 32-bit number: at most 2^21 segments of 1 024 threads, 2^23 of 256,
    // per launch; seg_off / segs are addressed from the launch's first segment, reports carry the global index)
    const bool small = max_n <= 256 * PBA_SS_EPT;
    const uint64_t per = small ? (1ull << 23) : (1ull << 21);
    for (uint64_t s0 = 0; s0 < n_seg; s0 += per) {
        const uint32_t g = (uint32_t)std::min<uint64_t>(n_seg - s0, per);
        if (small)
            hipLaunchKernelGGL(k_seg_sort<256>, dim3(g), dim3(256), 0, ctx->stream, src, dst, seg_off ? seg_off + s0 : nullptr,
                               segs ? segs + s0 : nullptr, bk, oversize, oversize_cap, (uint32_t)s0);
        else
            hipLaunchKernelGGL(k_seg_sort<1024>, dim3(g), dim3(1024), 0, ctx->stream, src, dst, seg_off ? seg_off + s0 : nullptr,
                               segs ? segs + s0 : nullptr, bk, oversize, oversize_cap, (uint32_t)s0);
    }
}

// ---- internal entry points that cross translation units
extern "C" {
// sort one oversize partition / candidate piece in global memory (pba_core.hip)
PBA_INTERNAL int sort_partition_global(pba_ctx *ctx, uint64_t *d_part, uint32_t n);
// edit scripts of a batch, or their votes (pba_align.hip)
struct pba_cons;
PBA_INTERNAL int trace_batch(pba_ctx *ctx, const pba_seqs *A, const pba_seqs *B, const pba_pair *pairs, size_t n, double R,
                             int maxn, int maxm, int kernel, pba_result *out, uint8_t *ops, const uint64_t *ops_off,
                             int32_t *nedit, const pba_cons *vote, int overlap_min);
// the vote boxes a batch of walks votes into (pba_cons.hip)
PBA_INTERNAL int cons_vote_view(const pba_cons *c, ConsDev *dev, int *beg, int *pre, int *post);
// one locked round over a subset of the reads (pba_drivers.hip)
PBA_INTERNAL int spaced_round_subset(pba_ctx *ctx, const pba_index *ix, const pba_seqs *ref, uint32_t ref_seq, const pba_seqs *reads,
                                     double R, int max_trial, int overlap_min, int buggy_seed_at, int kernel,
                                     const uint32_t *subset, uint32_t n_subset, pba_ss_row *rows, int ref_org = 0, int maxn = 0,
                                     int maxm = 0, uint8_t *touch = nullptr);
}  // extern "C"

#endif
