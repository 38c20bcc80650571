// pba_align.hip -- alignment of explicit pairs (seq_aligner<>::align) and edit scripts: kernels and C ABI.
// One process per GPU, one pba_ctx per process, one HIP stream per ctx.  Everything here fails loudly
// (PBA_E_NODEVICE / PBA_E_HIP): there is no CPU path behind these entry points.
#include "pba_host.h"
#include "align_bvtrace.h"

// ids (nullable): the subset of pairs / reads to process (second, full-band launch)
template <int NB>
__global__ void __launch_bounds__(PBA_WAVE * Wpb<NB>::v, Wpb<NB>::occ)
k_align_pairs(SeqSetDev A, SeqSetDev B, const pba_pair *pairs, const uint32_t *ids, uint32_t n, AlignCfg cfg,
              pba_result *out, uint32_t *queue) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / PBA_WAVE));   // wave-uniform on purpose: keeps the walk in SGPRs
    uint8_t *lds = lds_all + (size_t)wave * cfg.row_cap * 2;
    for (;;) {                                // persistent wavefront: pull the next pair until the queue is dry
        const uint32_t slot = next_slot(queue);
        if (slot >= n) break;
        const uint32_t q = ids ? ids[slot] : slot;
        const pba_pair pr = pairs[q];
        const PackedFetch fa = fetch_of(A, pr.a_seq, pr.a_pos, (pr.flags & PBA_A_BACKWARD) ? -1 : 1);
        const PackedFetch fb = fetch_of(B, pr.b_seq, pr.b_pos, (pr.flags & PBA_B_BACKWARD) ? -1 : 1);
        AlnOut o;
        align_dispatch<NB>(fa, pr.a_len, fb, pr.b_len, cfg, lds, o, true);
        store_result(out + q, o);
    }
}

__global__ void __launch_bounds__(PBA_WAVE)
k_align_bytes(const uint8_t *a, int a_dir, int la, const uint8_t *b, int b_dir, int lb, AlignCfg cfg,
              pba_result *out) {
    extern __shared__ __align__(16) uint8_t lds[];
    ByteFetch fa{a, a_dir}, fb{b, b_dir};
    AlnOut o;
    align_rowsweep(fa, la, fb, lb, cfg.R, cfg.maxn, cfg.maxm, (uint16_t *)lds, cfg.row_cap, o);
    store_result(out, o);
}

// ---- traceback: full-band row sweep that also stores one parent code per band cell, then a backward walk
__global__ void __launch_bounds__(PBA_WAVE)
k_align_pairs_trace(SeqSetDev A, SeqSetDev B, const pba_pair *pairs, uint32_t n, AlignCfg cfg, pba_result *out,
                    uint8_t *par, const uint64_t *par_off) {
    extern __shared__ __align__(16) uint8_t lds[];
    const uint32_t q = blockIdx.x;
    if (q >= n) return;
    const pba_pair pr = pairs[q];
    const PackedFetch fa = fetch_of(A, pr.a_seq, pr.a_pos, (pr.flags & PBA_A_BACKWARD) ? -1 : 1);
    const PackedFetch fb = fetch_of(B, pr.b_seq, pr.b_pos, (pr.flags & PBA_B_BACKWARD) ? -1 : 1);
    AlnOut o;
    align_rowsweep(fa, pr.a_len, fb, pr.b_len, cfg.R, cfg.maxn, cfg.maxm, (uint16_t *)lds, cfg.row_cap, o, par + par_off[q]);
    store_result(out + q, o);
}

__global__ void __launch_bounds__(PBA_WAVE)
k_align_bytes_trace(const uint8_t *a, int a_dir, int la, const uint8_t *b, int b_dir, int lb, AlignCfg cfg,
                    pba_result *out, uint8_t *par, uint16_t *cst) {
    extern __shared__ __align__(16) uint8_t lds[];
    ByteFetch fa{a, a_dir}, fb{b, b_dir};
    AlnOut o;
    align_rowsweep(fa, la, fb, lb, cfg.R, cfg.maxn, cfg.maxm, (uint16_t *)lds, cfg.row_cap, o, par, cst);
    store_result(out, o);
}

// ---- traceback on the bit-vector array (align_bvtrace.h): persistent wavefronts, each with its own scratch area
// of wave_words u32 (cap_words of parent bits, then the goal-first ops of the pair in flight).
// ids (nullable): the subset of pairs to process (second, full-band launch)
// CK: the checkpoint form of the traced pass (align_bvtrace.h): scratch holds one checkpoint per 32 steps, the walk re-runs
// chunks into this wavefront's LDS tile; else every step's parent words are streamed to scratch.
#ifndef PBA_TR_OCC12
#define PBA_TR_OCC12 6        // waves per SIMD of the checkpoint-form trace kernels at one or two blocks per lane (tuning hook;
                              // 32 768 reads of BASELINE configs[1], same box: 4 -> 63.0 ms, 5 -> 60.0, 6 -> 50.7, 8 -> 52.3)
#endif
template <int NB, bool CK>
__global__ void __launch_bounds__(PBA_WAVE * 4, NB <= 4 ? (CK && NB <= 2 ? PBA_TR_OCC12 : 4) : 2)
k_trace_pairs(SeqSetDev A, SeqSetDev B, const pba_pair *pairs, const uint32_t *ids, uint32_t n, AlignCfg cfg,
              pba_result *out, uint32_t *scratch, uint64_t wave_words, uint64_t cap_words, uint8_t *ops,
              const uint64_t *ops_off, int32_t *nedit, uint32_t *queue) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    __shared__ uint2 s_tile[4][CK ? PBA_BV_TILE_WORDS(NB) : 1];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / PBA_WAVE));
    uint8_t *lds = lds_all + (size_t)wave * cfg.row_cap * 2;
    uint32_t *mine = scratch + ((uint64_t)blockIdx.x * 4 + wave) * wave_words;
    for (;;) {
        const uint32_t slot = next_slot(queue);
        if (slot >= n) break;
        const uint32_t q = ids ? ids[slot] : slot;
        const pba_pair pr = pairs[q];
        const PackedFetch fa = fetch_of(A, pr.a_seq, pr.a_pos, (pr.flags & PBA_A_BACKWARD) ? -1 : 1);
        const PackedFetch fb = fetch_of(B, pr.b_seq, pr.b_pos, (pr.flags & PBA_B_BACKWARD) ? -1 : 1);
        AlnOut o;
        int ne = 0;
        const uint64_t o0 = ops_off[q], o1 = ops_off[q + 1];
        OpSink sink{(uint8_t *)(mine + cap_words), 0, 0u};
        if (align_bitvec_trace<NB, CK>(fa, pr.a_len, fb, pr.b_len, cfg.R, cfg.maxn, cfg.maxm, cfg.full_band != 0, (uint16_t *)lds,
                                       cfg.row_cap, mine, cap_words, 0, sink, o, s_tile[wave]))
            ne = sink.finish(ops + o0, o1 - o0);
        store_result(out + q, o);
        if ((threadIdx.x & (PBA_WAVE - 1)) == 0) nedit[q] = ne;
    }
}

// The same sweep and walk, but the path goes straight into the vote boxes of an unlocked reference (consensus.h:
// VoteSink) -- ref_seq::try_align's align + OVERLAP_MIN gate + elect (ref_seq.h:264-267) for a batch, no script in
// memory.  a is the reference: pair.a_pos is the position the votes start at.
template <int NB, bool CK>
__global__ void __launch_bounds__(PBA_WAVE * 4, NB <= 4 ? (CK && NB <= 2 ? PBA_TR_OCC12 : 4) : 2)
k_vote_pairs(SeqSetDev A, SeqSetDev B, const pba_pair *pairs, const uint32_t *ids, uint32_t n, AlignCfg cfg, int overlap_min,
             pba_result *out, uint32_t *scratch, uint64_t wave_words, uint64_t cap_words, ConsDev C, int beg, int pre, int post,
             uint32_t *queue) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    __shared__ uint2 s_tile[4][CK ? PBA_BV_TILE_WORDS(NB) : 1];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / PBA_WAVE));
    uint8_t *lds = lds_all + (size_t)wave * cfg.row_cap * 2;
    uint32_t *mine = scratch + ((uint64_t)blockIdx.x * 4 + wave) * wave_words;
    for (;;) {
        const uint32_t slot = next_slot(queue);
        if (slot >= n) break;
        const uint32_t q = ids ? ids[slot] : slot;
        const pba_pair pr = pairs[q];
        const bool fwd = !(pr.flags & PBA_A_BACKWARD);
        const PackedFetch fa = fetch_of(A, pr.a_seq, pr.a_pos, fwd ? 1 : -1);
        const PackedFetch fb = fetch_of(B, pr.b_seq, pr.b_pos, (pr.flags & PBA_B_BACKWARD) ? -1 : 1);
        AlnOut o;
        VoteSink sink{C, beg + pr.a_pos, pre, post, fwd, fb, 0, 0, 0, 0u};
        if (align_bitvec_trace<NB, CK>(fa, pr.a_len, fb, pr.b_len, cfg.R, cfg.maxn, cfg.maxm, cfg.full_band != 0, (uint16_t *)lds,
                                       cfg.row_cap, mine, cap_words, overlap_min, sink, o, s_tile[wave]))
            sink.finish();
        store_result(out + q, o);
    }
}

// find_path (seq_aligner.h:214-233) walked iteratively from the goal cell; one thread per pair
__global__ void k_trace_walk(const pba_result *res, const uint8_t *par, const uint64_t *par_off, uint8_t *ops,
                             const uint64_t *ops_off, int32_t *nedit, uint32_t n) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const pba_result r = res[q];
    if (r.rc < 0) { nedit[q] = 0; return; }
    const uint8_t *p = par + par_off[q];
    uint8_t *o = ops + ops_off[q];
    const uint64_t capq = ops_off[q + 1] - ops_off[q];
    const int md = r.max_dst, W = 2 * md + 1;
    int i = r.matlen_a, j = r.matlen_b;
    uint64_t k = 0;
    while (i > 0 || j > 0) {
        int src;
        if (j == 0) src = 3;                   // init_cell: (i,0) has parent DELETE, (0,j) INSERT (seq_aligner.h:140-147)
        else if (i == 0) src = 2;
        else src = p[(size_t)i * W + (j - i + md)];
        if (k < capq) o[k] = (uint8_t)src;
        ++k;
        if (src == 1) { --i; --j; } else if (src == 2) --j; else --i;
    }
    const uint64_t m = k < capq ? k : capq;
    for (uint64_t x = 0, y = m; x + 1 < y; ++x) { --y; const uint8_t t = o[x]; o[x] = o[y]; o[y] = t; }   // goal-first -> origin-first
    nedit[q] = (int32_t)k;
}


// (the attribute belongs to the current device: remembered per ctx, so a second ctx on another GPU sets it there too)
static void tu_attrs(pba_ctx *ctx) {
    if (ctx->attr_done & 1u) return;
    ctx->attr_done |= 1u;
    PBA_BIG_LDS(k_align_pairs<0>);
    PBA_BIG_LDS(k_align_bytes);
    PBA_BIG_LDS(k_align_bytes_trace);
    PBA_BIG_LDS(k_align_pairs_trace);
}

extern "C" {

int pba_align_batch(pba_ctx *ctx, const pba_seqs *A, const pba_seqs *B, const pba_pair *pairs, size_t n, double R,
                    int maxn, int maxm, int kernel, pba_result *out) {
    if (!ctx || !A || !B || (!pairs && n) || (!out && n)) return PBA_E_INVALID;
    if (n == 0) return PBA_OK;
    if (n > 0x7FFFFFFFull) PBA_FAIL(PBA_E_INVALID, "too many pairs in one batch");
    if (A->non_acgt || B->non_acgt)
        PBA_FAIL(PBA_E_ALPHABET, "a sequence set holds bytes outside ACGT: the reference compares raw bytes, use pba_align_text");
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs(ctx);
    int mdmax = 1;
    for (size_t q = 0; q < n; ++q) {
        const pba_pair &p = pairs[q];
        if (!pair_ok(A, p.a_seq, p.a_pos, p.a_len, p.flags & PBA_A_BACKWARD) ||
            !pair_ok(B, p.b_seq, p.b_pos, p.b_len, p.flags & PBA_B_BACKWARD))
            PBA_FAIL(PBA_E_INVALID, "pair outside its sequence (or longer than the engine limit)");
        if (R > 0.0 && R < 1.0) mdmax = std::max(mdmax, max_dst_of(p.a_len, p.b_len, R));
    }
    Plan pl;
    int st = make_plan(ctx, R, maxn, maxm, kernel, mdmax, &pl);
    if (st != PBA_OK) return st;
    DevBuf d_pairs, d_out, d_ids;
    HIPCHK(hipMalloc(&d_pairs.p, sizeof(pba_pair) * n));
    HIPCHK(hipMalloc(&d_out.p, sizeof(pba_result) * n));
    HIPCHK(hipMemcpyAsync(d_pairs.p, pairs, sizeof(pba_pair) * n, hipMemcpyHostToDevice, ctx->stream));
#define K_PAIRS(NBV)                                                                                               \
    (void)hipMemsetAsync(ctx->d_queue, 0, 4, ctx->stream);                                                         \
    hipLaunchKernelGGL(k_align_pairs<NBV>, dim3(persistent_grid(ctx, cnt, Wpb<NBV>::v, pl.lds)),                    \
                       dim3(PBA_WAVE * Wpb<NBV>::v), pl.lds * Wpb<NBV>::v, ctx->stream, A->dev(), B->dev(),         \
                       d_pairs.as<pba_pair>(), ids, cnt, pl.cfg, d_out.as<pba_result>(), ctx->d_queue)
    {
        const uint32_t cnt = (uint32_t)n;
        const uint32_t *ids = nullptr;
        (void)hipEventRecord(ctx->ev[2], ctx->stream);
        PBA_DISPATCH_NB(pl.nb1, K_PAIRS);
        (void)hipEventRecord(ctx->ev[3], ctx->stream);
        ctx->prof.nb_first = (uint32_t)pl.nb1; ctx->prof.n_first = cnt; ctx->prof.nb_redo = 0; ctx->prof.n_redo = 0;
        ctx->prof.align_redo_ms = 0.f;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // pairs whose narrow pass could not certify the goal row go round again at the reference band
    std::vector<uint32_t> redo;
    for (size_t q = 0; q < n; ++q)
        if (out[q].rc == PBA_RC_UNCERTIFIED) redo.push_back((uint32_t)q);
    if (!redo.empty()) {
        HIPCHK(hipMalloc(&d_ids.p, sizeof(uint32_t) * redo.size()));
        HIPCHK(hipMemcpyAsync(d_ids.p, redo.data(), sizeof(uint32_t) * redo.size(), hipMemcpyHostToDevice, ctx->stream));
        pl.cfg.full_band = 1;
        const uint32_t cnt = (uint32_t)redo.size();
        const uint32_t *ids = d_ids.as<uint32_t>();
        (void)hipEventRecord(ctx->ev[4], ctx->stream);
        PBA_DISPATCH_NB(pl.nb2, K_PAIRS);
        (void)hipEventRecord(ctx->ev[5], ctx->stream);
        ctx->prof.nb_redo = (uint32_t)pl.nb2; ctx->prof.n_redo = cnt;
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
#undef K_PAIRS
    prof_finish(ctx);
    return PBA_OK;
}

// scratch the traced bit-vector pass of one pair needs (u32 words): narrow first pass or reference band
static uint64_t trace_words_of(int la, int lb, double R, int nb, bool full_band, bool ck) {
    const int md = max_dst_of(la, lb, R);
    const int len_a = lb >= la ? la : std::min(la, lb + md), len_b = lb >= la ? std::min(lb, la + md) : lb;
    const int m = std::min(len_a, len_b), n = std::max(len_a, len_b);
    if (m <= 10) return (((uint64_t)len_a + 1) * (2ull * md + 1) + 3) / 4;       // the row sweep's corner: byte codes
    const int w = full_band ? md : bv_pass1_w(md, nb);
    return ck ? bv_ck_words(nb, m, n, w) : bv_trace_words(nb, m, n, w);
}


// ---------------------------------------------------------------------------------------------
// One pair handed over as host text: what the compat seq_aligner<>::align (include/compat/seq_aligner.h) calls.
// ---------------------------------------------------------------------------------------------
static const uint64_t kTraceBudget = 96ull << 30;      // parent codes / bits resident for one call (and at most 80 % of free HBM)

// What align() looks at of its two accessors.  seq_aligner.h:94-102 clips the longer one to the shorter + max_dst BEFORE
// anything is sized, checked or read -- locator.cpp:80-81 hands it the whole rest of an 800 kb contig, ref_seq.h:282-286 the
// whole rest of the reference -- so the engine's own limit, the size guard and the H2D copy all see the clipped lengths.
// Same FP64 product and truncation as aln_params (dev_common.h); given (len_a, len_b) the kernels derive the same block again.
struct TextClip { int len_a, len_b, md; };
static inline TextClip text_clip(int la, int lb, double R) {
    TextClip c;
    if (lb >= la) { c.len_a = la; c.md = 1 + (int)((double)la * R); c.len_b = (int)std::min<long long>(lb, (long long)la + c.md); }
    else          { c.len_b = lb; c.md = 1 + (int)((double)lb * R); c.len_a = (int)std::min<long long>(la, (long long)lb + c.md); }
    return c;
}
// the reference's size guard (seq_aligner.h:104-107: LOG, return -1) answered on the host; 1 = *out is final, nothing to launch
static inline int text_guard(pba_ctx *ctx, const TextClip &c, int maxn, int maxm, pba_result *out, const char *who) {
    if (maxn > 0 && ((long long)c.len_a >= (long long)maxn + maxm || c.md >= maxm)) {
        out->rc = -1; out->cost = 0; out->matlen_a = 0; out->matlen_b = 0;
        out->len_a = c.len_a; out->len_b = c.len_b; out->max_dst = c.md; out->diag_cost = -1;
        return 1;
    }
    if (c.len_a > kMaxSeqLen || c.len_b > kMaxSeqLen) PBA_FAIL(PBA_E_TOOLONG, who);
    return 0;
}
// elements 0 .. len-1 of an accessor in text order: element k of a backward accessor is p[-k] (dna_seq.h:211,221), so its
// elements are the bytes [p-(len-1), p] and the accessor's origin is the last of them
static inline const uint8_t *acc_low(const char *p, int fwd, int len) { return (const uint8_t *)((fwd || len == 0) ? p : p - (len - 1)); }

static inline int scratch_reserve(pba_ctx *ctx, size_t need) {
    if (need <= ctx->scratch_bytes) return PBA_OK;
    if (ctx->d_scratch) { HIPCHK(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_scratch); }
    ctx->d_scratch = nullptr; ctx->scratch_bytes = 0;
    HIPCHK(hipMalloc(&ctx->d_scratch, need));
    ctx->scratch_bytes = need;
    return PBA_OK;
}

// 2-bit codes (reference byte layout, dna_seq.h:147-159) and the two bit planes (dev_common.h: SeqSetDev::plane) of `len`
// text bytes; false at the first byte outside "ACGT" (the raw-byte row sweep takes the pair then: the reference compares bytes)
static bool pack_acgt(const uint8_t *src, int len, uint8_t *packed, uint32_t *plane) {
    static const struct Lut { uint8_t v[256]; Lut() { memset(v, 0xFF, sizeof v); v['A'] = 0; v['C'] = 1; v['G'] = 2; v['T'] = 3; } } lut;
    for (int w = 0; w * 32 < len; ++w) {
        const int nb = std::min(32, len - w * 32);
        uint32_t plo = 0, phi = 0, bad = 0;
        uint8_t *pk = packed + (size_t)w * 8;
        for (int k = 0; k < nb; ++k) {
            const uint32_t code = lut.v[src[w * 32 + k]];
            bad |= code;
            plo |= (code & 1u) << k;
            phi |= ((code >> 1) & 1u) << k;
            pk[k >> 2] |= (uint8_t)((code & 3u) << (6 - 2 * (k & 3)));
        }
        if (bad & 0x80u) return false;
        plane[2 * w] = plo; plane[2 * w + 1] = phi;
    }
    return true;
}

// A staged pair as a two-sequence set in one pooled device buffer (sequence 0 = a, sequence 1 = b, both stored in text
// order; a backward accessor starts at its last base with the BACKWARD flag): header, packed bases with kSlack zero bytes
// around them, bit planes with kPlaneSlack zero word pairs around them -- the layout pba_seqs gives a set (pba_core.hip).
struct TextStage {
    size_t o_pair, o_off, o_len, o_poff, o_ooff, o_packed, o_plane, bytes;
    uint64_t pkA, pkB, wA, wB;
};
static inline TextStage text_stage_layout(const TextClip &c) {
    TextStage t;
    t.pkA = (((uint64_t)c.len_a + 3) / 4 + 15) & ~15ull; t.pkB = (((uint64_t)c.len_b + 3) / 4 + 15) & ~15ull;
    t.wA = ((uint64_t)c.len_a + 31) / 32; t.wB = ((uint64_t)c.len_b + 31) / 32;
    t.o_pair = 0; t.o_off = 32; t.o_len = 64; t.o_poff = 80; t.o_ooff = 128;
    t.o_packed = 256 + kSlack;
    t.o_plane = ((t.o_packed + t.pkA + t.pkB + kSlack + 63) & ~(size_t)63) + kPlaneSlack * 8;
    t.bytes = t.o_plane + (t.wA + t.wB) * 8 + kPlaneSlack * 8;
    return t;
}

// where the results of a staged pair land (POOL_TXT_OUT): the result, nedit, then the ops
static const size_t kTxtOutOps = 64;

static void launch_trace_pairs(pba_ctx *ctx, int nb, bool ck, uint32_t grid, size_t lds, const SeqSetDev &A, const SeqSetDev &B,
                               const pba_pair *pairs, const uint32_t *ids, uint32_t cnt, const AlignCfg &cfg, pba_result *out,
                               uint32_t *scr, uint64_t wave_words, uint64_t cap_words, uint8_t *ops, const uint64_t *ooff, int32_t *ne) {
#define K_TP(NBV, CKV)                                                                                                \
    hipLaunchKernelGGL((k_trace_pairs<NBV, CKV>), dim3(grid), dim3(PBA_WAVE * 4), lds * 4, ctx->stream, A, B, pairs, ids, cnt, \
                       cfg, out, scr, wave_words, cap_words, ops, ooff, ne, ctx->d_queue)
#define K_TPN(NBV) if (ck) { K_TP(NBV, true); } else { K_TP(NBV, false); }
    switch (nb) {
        case 1: K_TPN(1); break;
        case 2: K_TPN(2); break;
        case 3: K_TPN(3); break;
        case 4: K_TPN(4); break;
        case 6: K_TPN(6); break;
        default: K_TPN(8); break;
    }
#undef K_TPN
#undef K_TP
}

// The fast form of the three text entry points: both accessors hold ACGT only, so comparing 2-bit codes is comparing bytes
// and the pair runs on the bit-vector array like a pair of a batch (narrow window first, the reference band if that cannot
// certify; with `ops` the checkpoint-and-recompute traced pass, align_bvtrace.h).  One H2D copy, one or two launches on one
// wavefront, one D2H copy; every buffer is the ctx's.  Returns 1 when the pair is not ACGT-only (nothing was launched).
static int text_pair_bitvec(pba_ctx *ctx, const char *a, int a_fwd, const char *b, int b_fwd, const TextClip &c, double R,
                            int maxn, int maxm, pba_result *out, uint8_t *ops, int32_t ops_cap, int32_t *nedit, bool want_trace) {
    if (!bitvec_supports(c.md)) return 1;
    const TextStage t = text_stage_layout(c);
    const uint64_t ops_room = (uint64_t)c.len_a + c.len_b + 64;
    // the D2H copy reuses the staging buffer (the input image has been consumed by then: same stream)
    const size_t res_bytes = kTxtOutOps + (want_trace ? ops_room : 0);
    int st = stage_reserve(ctx, std::max(t.bytes, res_bytes));
    if (st != PBA_OK) return st;
    uint8_t *h = (uint8_t *)ctx->h_stage;
    memset(h, 0, t.bytes);
    if (!pack_acgt(acc_low(a, a_fwd, c.len_a), c.len_a, h + t.o_packed, (uint32_t *)(h + t.o_plane)) ||
        !pack_acgt(acc_low(b, b_fwd, c.len_b), c.len_b, h + t.o_packed + t.pkA, (uint32_t *)(h + t.o_plane) + 2 * t.wA))
        return 1;
    Plan pl;
    st = make_plan(ctx, R, maxn, maxm, PBA_KERNEL_AUTO, c.md, &pl);
    if (st != PBA_OK) return st;
    pba_pair pr;
    pr.a_seq = 0; pr.a_pos = a_fwd || !c.len_a ? 0 : c.len_a - 1; pr.a_len = c.len_a;
    pr.b_seq = 1; pr.b_pos = b_fwd || !c.len_b ? 0 : c.len_b - 1; pr.b_len = c.len_b;
    pr.flags = (a_fwd ? 0u : PBA_A_BACKWARD) | (b_fwd ? 0u : PBA_B_BACKWARD);
    memcpy(h + t.o_pair, &pr, sizeof pr);
    const uint64_t off[3] = {0, t.pkA, t.pkA + t.pkB}, poff[3] = {0, t.wA, t.wA + t.wB}, ooff[2] = {0, ops_room};
    const uint32_t len[3] = {(uint32_t)c.len_a, (uint32_t)c.len_b, 0};
    memcpy(h + t.o_off, off, sizeof off); memcpy(h + t.o_len, len, sizeof len);
    memcpy(h + t.o_poff, poff, sizeof poff); memcpy(h + t.o_ooff, ooff, sizeof ooff);
    uint8_t *d_in = nullptr, *d_res = nullptr;
    POOL(POOL_TXT_IN, t.bytes, d_in);
    POOL(POOL_TXT_OUT, kTxtOutOps + ops_room + 64, d_res);
    HIPCHK(hipMemcpyAsync(d_in, h, t.bytes, hipMemcpyHostToDevice, ctx->stream));
    const SeqSetDev S{d_in + t.o_packed, (const uint64_t *)(d_in + t.o_off), (const uint32_t *)(d_in + t.o_len),
                      (const uint32_t *)(d_in + t.o_plane), (const uint64_t *)(d_in + t.o_poff)};
    const char *e_stream = getenv("PBA_TRACE_STREAM");
    const bool ck = !(e_stream && atoi(e_stream) != 0);
    ctx->prof.nb_first = (uint32_t)pl.nb1; ctx->prof.n_first = 1; ctx->prof.nb_redo = 0; ctx->prof.n_redo = 0; ctx->prof.align_redo_ms = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        const int nb = pass ? pl.nb2 : pl.nb1;
        pl.cfg.full_band = pass;
        HIPCHK(hipMemsetAsync(ctx->d_queue, 0, sizeof(uint32_t), ctx->stream));
        (void)hipEventRecord(ctx->ev[pass ? 4 : 2], ctx->stream);
        if (want_trace) {
            uint64_t cap_words = std::max<uint64_t>(128, trace_words_of(c.len_a, c.len_b, R, nb, pass != 0, ck));
            cap_words = (cap_words + 63) & ~63ull;
            const uint64_t wave_words = cap_words + ((ops_room + 255) & ~255ull) / 4;
            st = scratch_reserve(ctx, (size_t)wave_words * 4 * 4);       // whichever of the workgroup's four wavefronts takes the pair
            if (st != PBA_OK) return st;
            launch_trace_pairs(ctx, nb, ck, 1, pl.lds, S, S, (const pba_pair *)(d_in + t.o_pair), nullptr, 1, pl.cfg,
                               (pba_result *)d_res, (uint32_t *)ctx->d_scratch, wave_words, cap_words, d_res + kTxtOutOps,
                               (const uint64_t *)(d_in + t.o_ooff), (int32_t *)(d_res + 32));
        } else {
#define K_ONE(NBV)                                                                                                   \
    hipLaunchKernelGGL(k_align_pairs<NBV>, dim3(1), dim3(PBA_WAVE * Wpb<NBV>::v), pl.lds * Wpb<NBV>::v, ctx->stream, S, S, \
                       (const pba_pair *)(d_in + t.o_pair), (const uint32_t *)nullptr, 1u, pl.cfg, (pba_result *)d_res, ctx->d_queue)
            PBA_DISPATCH_NB(nb, K_ONE);
#undef K_ONE
        }
        (void)hipEventRecord(ctx->ev[pass ? 5 : 3], ctx->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ctx->h_stage, d_res, res_bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        memcpy(out, ctx->h_stage, sizeof(pba_result));
        if (pass) { ctx->prof.nb_redo = (uint32_t)nb; ctx->prof.n_redo = 1; }
        if (out->rc != PBA_RC_UNCERTIFIED) break;
        if (pass) PBA_FAIL(PBA_E_HIP, "reference-band pass left a pair uncertified");     // cannot happen (align_bitvec.h)
    }
    if (out->rc == -2) PBA_FAIL(PBA_E_TOOLONG, "pair outside what the launch was sized for");   // host sizes both: cannot happen
    if (want_trace) {
        int32_t ne = 0;
        memcpy(&ne, (const uint8_t *)ctx->h_stage + 32, sizeof ne);
        *nedit = ne;
        const int32_t ncopy = std::min(ne, ops_cap);
        if (ncopy > 0) memcpy(ops, (const uint8_t *)ctx->h_stage + kTxtOutOps, (size_t)ncopy);
    }
    prof_finish(ctx);
    return PBA_OK;
}

// test hook: PBA_TEXT_ROWSWEEP=1 sends every pair through the general form below (the two forms are cross-checked)
static inline bool text_force_rowsweep() {
    const char *e = getenv("PBA_TEXT_ROWSWEEP");
    return e && atoi(e) != 0;
}

// The general form: raw bytes (any alphabet, case-sensitive, seq_aligner.h:136) through the reference-shaped row sweep on one
// wavefront; par / cst as align_rowsweep takes them.  Only the clipped elements are shipped.
static int text_pair_stage_bytes(pba_ctx *ctx, const char *a, int a_fwd, const char *b, int b_fwd, const TextClip &c,
                                 const uint8_t **da, const uint8_t **db) {
    const size_t ob = ((size_t)c.len_a + 31) & ~(size_t)15, bytes = ob + c.len_b + 32;
    int st = stage_reserve(ctx, bytes);
    if (st != PBA_OK) return st;
    uint8_t *h = (uint8_t *)ctx->h_stage, *d_in = nullptr;
    if (c.len_a) memcpy(h, acc_low(a, a_fwd, c.len_a), c.len_a);
    if (c.len_b) memcpy(h + ob, acc_low(b, b_fwd, c.len_b), c.len_b);
    POOL(POOL_TXT_IN, bytes, d_in);
    HIPCHK(hipMemcpyAsync(d_in, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    *da = d_in + (a_fwd || !c.len_a ? 0 : c.len_a - 1);
    *db = d_in + ob + (b_fwd || !c.len_b ? 0 : c.len_b - 1);
    return PBA_OK;
}

int pba_align_text(pba_ctx *ctx, const char *a, int a_fwd, int la, const char *b, int b_fwd, int lb, double R,
                   int maxn, int maxm, pba_result *out) {
    if (!ctx || !out || la < 0 || lb < 0 || (!a && la) || (!b && lb)) return PBA_E_INVALID;
    if (!(R > 0.0) || !(R < 1.0)) PBA_FAIL(PBA_E_INVALID, "R must be in (0,1)");
    const TextClip c = text_clip(la, lb, R);
    int st = text_guard(ctx, c, maxn, maxm, out, "pba_align_text");
    if (st) return st < 0 ? st : PBA_OK;
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs(ctx);
    if (!text_force_rowsweep()) {
        st = text_pair_bitvec(ctx, a, a_fwd, b, b_fwd, c, R, maxn, maxm, out, nullptr, 0, nullptr, false);
        if (st != 1) return st;
    }
    Plan pl;
    st = make_plan(ctx, R, maxn, maxm, PBA_KERNEL_ROWSWEEP, c.md, &pl);
    if (st != PBA_OK) return st;
    const uint8_t *da = nullptr, *db = nullptr;
    st = text_pair_stage_bytes(ctx, a, a_fwd, b, b_fwd, c, &da, &db);
    if (st != PBA_OK) return st;
    uint8_t *d_res = nullptr;
    POOL(POOL_TXT_OUT, kTxtOutOps, d_res);
    hipLaunchKernelGGL(k_align_bytes, dim3(1), dim3(PBA_WAVE), pl.lds, ctx->stream, da, a_fwd ? 1 : -1, c.len_a, db,
                       b_fwd ? 1 : -1, c.len_b, pl.cfg, (pba_result *)d_res);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_res, sizeof(pba_result), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

// ---- traceback
static uint64_t par_bytes_of(int la, int lb, double R) {           // (len_a + 1) * (2*max_dst + 1), seq_aligner.h:94-102
    const int md = max_dst_of(la, lb, R);
    const int len_a = lb >= la ? la : std::min(la, lb + md);
    return ((uint64_t)len_a + 1) * (2ull * md + 1);
}

int pba_align_text_trace(pba_ctx *ctx, const char *a, int a_fwd, int la, const char *b, int b_fwd, int lb, double R,
                         int maxn, int maxm, pba_result *out, uint8_t *ops, int32_t ops_cap, int32_t *nedit) {
    if (!ctx || !out || !nedit || la < 0 || lb < 0 || (!a && la) || (!b && lb) || (!ops && ops_cap) || ops_cap < 0)
        return PBA_E_INVALID;
    if (!(R > 0.0) || !(R < 1.0)) PBA_FAIL(PBA_E_INVALID, "R must be in (0,1)");
    const TextClip c = text_clip(la, lb, R);
    *nedit = 0;
    int st = text_guard(ctx, c, maxn, maxm, out, "pba_align_text_trace");
    if (st) return st < 0 ? st : PBA_OK;
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs(ctx);
    if (!text_force_rowsweep()) {
        st = text_pair_bitvec(ctx, a, a_fwd, b, b_fwd, c, R, maxn, maxm, out, ops, ops_cap, nedit, true);
        if (st != 1) return st;
    }
    Plan pl;
    st = make_plan(ctx, R, maxn, maxm, PBA_KERNEL_ROWSWEEP, c.md, &pl);
    if (st != PBA_OK) return st;
    const uint64_t pb = ((uint64_t)c.len_a + 1) * (2ull * c.md + 1);
    if (pb > kTraceBudget) PBA_FAIL(PBA_E_NOMEM, "parent codes exceed the traceback budget");
    const uint8_t *da = nullptr, *db = nullptr;
    st = text_pair_stage_bytes(ctx, a, a_fwd, b, b_fwd, c, &da, &db);
    if (st != PBA_OK) return st;
    // result, nedit and the offsets the walk reads (ops_off[0..1], par_off[0]), then the ops
    const size_t o_off = 64, o_ops = 128;
    uint8_t *d_res = nullptr, *d_par = nullptr;
    POOL(POOL_TXT_OUT, o_ops + (size_t)ops_cap + 16, d_res);
    POOL(POOL_TXT_PAR, pb + 16, d_par);
    const uint64_t offs[4] = {0, (uint64_t)ops_cap, 0, 0};
    HIPCHK(hipMemcpyAsync(d_res + o_off, offs, sizeof offs, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_align_bytes_trace, dim3(1), dim3(PBA_WAVE), pl.lds, ctx->stream, da, a_fwd ? 1 : -1, c.len_a, db,
                       b_fwd ? 1 : -1, c.len_b, pl.cfg, (pba_result *)d_res, d_par, (uint16_t *)nullptr);
    hipLaunchKernelGGL(k_trace_walk, dim3(1), dim3(64), 0, ctx->stream, (const pba_result *)d_res, (const uint8_t *)d_par,
                       (const uint64_t *)(d_res + o_off) + 2, d_res + o_ops, (const uint64_t *)(d_res + o_off),
                       (int32_t *)(d_res + 32), 1u);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_res, sizeof(pba_result), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(nedit, d_res + 32, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));                   // (offs is a host array: the copy above has completed too)
    const int32_t ncopy = std::min(*nedit, ops_cap);
    if (ncopy > 0) {
        HIPCHK(hipMemcpyAsync(ops, d_res + o_ops, (size_t)ncopy, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return PBA_OK;
}

// The reference's DP matrix of one pair (seq_aligner.h:81 `mat`, read through get_cost / get_parent :131-134 by
// locator.cpp:86 and by whoever inspects an alignment): cost[i * W + c] / parent[i * W + c] for cell (i, j), W = 2*max_dst+1,
// c = j - i + max_dst -- the reference's own diagonal-stripe layout.  Cells the call writes hold their values (init_cell's
// borders, the band of every row swept: all of them, or up to the row of the early failure, out->diag_cost / rc tell which);
// the others hold cost 0xFFFF, parent 0 (the reference leaves whatever an earlier call wrote there).
int pba_align_text_matrix(pba_ctx *ctx, const char *a, int a_fwd, int la, const char *b, int b_fwd, int lb, double R, int maxn,
                          int maxm, pba_result *out, uint16_t *cost, uint8_t *parent, uint64_t cap_cells, int32_t *rows_swept) {
    if (!ctx || !out || la < 0 || lb < 0 || (!a && la) || (!b && lb) || ((!cost || !parent) && cap_cells)) return PBA_E_INVALID;
    if (!(R > 0.0) || !(R < 1.0)) PBA_FAIL(PBA_E_INVALID, "R must be in (0,1)");
    const TextClip c = text_clip(la, lb, R);
    const int md = c.md;
    const uint64_t pb = ((uint64_t)c.len_a + 1) * (2ull * md + 1);      // cells: (len_a + 1) * (2*max_dst + 1)
    if (rows_swept) *rows_swept = 0;
    int st = text_guard(ctx, c, maxn, maxm, out, "pba_align_text_matrix");
    if (st < 0) return st;
    if (pb * 3 > kTraceBudget) PBA_FAIL(PBA_E_NOMEM, "the matrix exceeds the traceback budget");
    if (cap_cells < pb) PBA_FAIL(PBA_E_INVALID, "pba_align_text_matrix: cost / parent hold fewer than (len_a + 1) * (2*max_dst + 1) cells");
    if (st == 1) {                                                  // the size guard: nothing is written (seq_aligner.h:104-107)
        memset(cost, 0xFF, (size_t)pb * 2); memset(parent, 0, (size_t)pb);
        return PBA_OK;
    }
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs(ctx);
    Plan pl;
    st = make_plan(ctx, R, maxn, maxm, PBA_KERNEL_ROWSWEEP, md, &pl);
    if (st != PBA_OK) return st;
    const uint8_t *da = nullptr, *db = nullptr;
    st = text_pair_stage_bytes(ctx, a, a_fwd, b, b_fwd, c, &da, &db);
    if (st != PBA_OK) return st;
    uint8_t *d_res = nullptr, *d_par = nullptr, *d_cst = nullptr;
    POOL(POOL_TXT_OUT, kTxtOutOps, d_res);
    POOL(POOL_TXT_PAR, pb + 16, d_par);
    POOL(POOL_TXT_CST, 2 * pb + 16, d_cst);
    HIPCHK(hipMemsetAsync(d_par, 0, pb, ctx->stream));
    HIPCHK(hipMemsetAsync(d_cst, 0xFF, 2 * pb, ctx->stream));
    hipLaunchKernelGGL(k_align_bytes_trace, dim3(1), dim3(PBA_WAVE), pl.lds, ctx->stream, da, a_fwd ? 1 : -1, c.len_a, db,
                       b_fwd ? 1 : -1, c.len_b, pl.cfg, (pba_result *)d_res, d_par, (uint16_t *)d_cst);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_res, sizeof(pba_result), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(cost, d_cst, 2 * pb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(parent, d_par, pb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // init_cell (seq_aligner.h:139-150): row 0 is D(0,j) = j, INSERT, for j <= max_dst
    const uint64_t W = 2ull * md + 1;
    int swept = 0;
    for (int j = 0; j <= md; ++j) { cost[(uint64_t)md + j] = (uint16_t)j; parent[(uint64_t)md + j] = j ? 2 : 0; }
    // rows swept: every row up to len_a, or up to the early failure -- the last row whose diagonal-side cell was written
    for (swept = out->len_a; swept > 0; --swept) {
        const int jlo = swept - md > 0 ? swept - md : 0;
        if (cost[(uint64_t)swept * W + (uint64_t)(jlo - swept + md)] != 0xFFFF) break;
    }
    if (rows_swept) *rows_swept = swept;
    return PBA_OK;
}

// Edit scripts of a batch (vote == nullptr: ops / ops_off / nedit receive them) or their votes (vote != nullptr: the
// paths go straight into its boxes, gated by overlap_min; ops / ops_off / nedit unused).
int trace_batch(pba_ctx *ctx, const pba_seqs *A, const pba_seqs *B, const pba_pair *pairs, size_t n, double R,
                       int maxn, int maxm, int kernel, pba_result *out, uint8_t *ops, const uint64_t *ops_off,
                       int32_t *nedit, const pba_cons *vote, int overlap_min) {
    if (!ctx || !A || !B || (!pairs && n) || (!out && n) || (!vote && ((!ops_off && n) || (!nedit && n)))) return PBA_E_INVALID;
    if (n == 0) return PBA_OK;
    if (n > 0x7FFFFFFFull) PBA_FAIL(PBA_E_INVALID, "too many pairs in one batch");
    if (A->non_acgt || B->non_acgt) PBA_FAIL(PBA_E_ALPHABET, "a sequence set holds bytes outside ACGT: use pba_align_text_trace");
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs(ctx);
    int mdmax = 1;
    uint64_t ops_max = 0;
    for (size_t q = 0; q < n; ++q) {
        const pba_pair &p = pairs[q];
        if (!pair_ok(A, p.a_seq, p.a_pos, p.a_len, p.flags & PBA_A_BACKWARD) ||
            !pair_ok(B, p.b_seq, p.b_pos, p.b_len, p.flags & PBA_B_BACKWARD))
            PBA_FAIL(PBA_E_INVALID, "pair outside its sequence (or longer than the engine limit)");
        if (!vote && (ops_off[q + 1] < ops_off[q] || ops_off[q + 1] - ops_off[q] < (uint64_t)p.a_len + p.b_len))
            PBA_FAIL(PBA_E_INVALID, "ops_off must leave a_len + b_len slots per pair");
        if (R > 0.0 && R < 1.0) mdmax = std::max(mdmax, max_dst_of(p.a_len, p.b_len, R));
        ops_max = std::max(ops_max, (uint64_t)p.a_len + p.b_len);
    }
    Plan pl;
    int st = make_plan(ctx, R, maxn, maxm, kernel, mdmax, &pl);
    if (st != PBA_OK) return st;
    ConsDev vdev = {nullptr, nullptr, nullptr, nullptr};
    int vbeg = 0, vpre = 0, vpost = 0;
    if (vote) {
        if (pl.nb1 == 0) PBA_FAIL(PBA_E_TOOLONG, "votes from the walk need the bit-vector kernel (band too wide)");
        st = cons_vote_view(vote, &vdev, &vbeg, &vpre, &vpost);
        if (st != PBA_OK) return st;
        ops_max = 0;                                             // no goal-first temporary
    }
    const uint64_t ops_total = vote ? 0 : ops_off[n] - ops_off[0];
    DevBuf d_pairs, d_out, d_par, d_poff, d_ops, d_ooff, d_ne;
    HIPCHK(hipMalloc(&d_pairs.p, sizeof(pba_pair) * n));
    HIPCHK(hipMalloc(&d_out.p, sizeof(pba_result) * n));
    HIPCHK(hipMalloc(&d_ops.p, ops_total + 16));
    HIPCHK(hipMalloc(&d_ooff.p, sizeof(uint64_t) * (n + 1)));
    HIPCHK(hipMalloc(&d_ne.p, sizeof(int32_t) * n));
    std::vector<uint64_t> rel(n + 1, 0);
    if (!vote) for (size_t q = 0; q <= n; ++q) rel[q] = ops_off[q] - ops_off[0];
    HIPCHK(hipMemcpyAsync(d_pairs.p, pairs, sizeof(pba_pair) * n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_ooff.p, rel.data(), sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    free_b += ctx->scratch_bytes;                            // the scratch kept from earlier calls is ours to reuse
    uint64_t budget = std::min<uint64_t>(kTraceBudget, (uint64_t)(free_b / 10) * 8);
    if (const char *e = getenv("PBA_TRACE_BUDGET_GB"))       // tuning aid: HBM the parent bits / codes of one call may take
        budget = std::min<uint64_t>((uint64_t)atoll(e) << 30, (uint64_t)(free_b / 10) * 9);
    (void)hipEventRecord(ctx->ev[2], ctx->stream);
    if (pl.nb1 == 0) {
        // row sweep: one parent byte per band cell, every pair's codes resident at once
        std::vector<uint64_t> par_off(n + 1, 0);
        for (size_t q = 0; q < n; ++q)
            par_off[q + 1] = par_off[q] + ((par_bytes_of(pairs[q].a_len, pairs[q].b_len, R) + 15) & ~15ull);
        if (par_off[n] > budget) PBA_FAIL(PBA_E_NOMEM, "parent codes of this batch exceed the traceback budget: split it");
        HIPCHK(hipMalloc(&d_par.p, par_off[n] + 16));
        HIPCHK(hipMalloc(&d_poff.p, sizeof(uint64_t) * (n + 1)));
        HIPCHK(hipMemcpyAsync(d_poff.p, par_off.data(), sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_align_pairs_trace, dim3((uint32_t)n), dim3(PBA_WAVE), pl.lds, ctx->stream, A->dev(), B->dev(),
                           d_pairs.as<pba_pair>(), (uint32_t)n, pl.cfg, d_out.as<pba_result>(), d_par.as<uint8_t>(),
                           d_poff.as<uint64_t>());
        hipLaunchKernelGGL(k_trace_walk, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, ctx->stream, d_out.as<pba_result>(),
                           d_par.as<uint8_t>(), d_poff.as<uint64_t>(), d_ops.as<uint8_t>(), d_ooff.as<uint64_t>(),
                           d_ne.as<int32_t>(), (uint32_t)n);
        (void)hipEventRecord(ctx->ev[3], ctx->stream);
        ctx->prof.nb_first = 0; ctx->prof.n_first = (uint32_t)n; ctx->prof.nb_redo = 0; ctx->prof.n_redo = 0;
        ctx->prof.align_redo_ms = 0.f;
    } else {
        // bit-vector array: 2 bits per processed cell in a per-wavefront scratch area, walked by the same wavefront.
        // First launch: every pair, narrow window, scratch sized for it (so more wavefronts fit the budget); second
        // launch: the pairs that came back uncertified, reference band.
        ctx->prof.nb_first = (uint32_t)pl.nb1; ctx->prof.n_first = (uint32_t)n;
        ctx->prof.nb_redo = 0; ctx->prof.n_redo = 0; ctx->prof.align_redo_ms = 0.f;
        // checkpoints + recomputation (default) or every step's words streamed to HBM (PBA_TRACE_STREAM=1: the round-1 form,
        // kept for comparison)
        const char *e_stream = getenv("PBA_TRACE_STREAM");
        const bool ck = !(e_stream && atoi(e_stream) != 0);
        std::vector<uint32_t> redo;
        for (int pass = 0; pass < 2; ++pass) {
            const int nb = pass ? pl.nb2 : pl.nb1;
            const uint32_t cnt = pass ? (uint32_t)redo.size() : (uint32_t)n;
            uint64_t cap_words = 128;
            for (uint32_t k = 0; k < cnt; ++k) {
                const pba_pair &p = pairs[pass ? redo[k] : k];
                cap_words = std::max(cap_words, trace_words_of(p.a_len, p.b_len, R, nb, pass != 0, ck));
            }
            cap_words = (cap_words + 63) & ~63ull;
            const uint64_t wave_words = cap_words + ((ops_max + 64 + 255) & ~255ull) / 4;
            uint32_t grid = persistent_grid(ctx, cnt, 4, pl.lds);
            grid = (uint32_t)std::min<uint64_t>(grid, budget / (wave_words * 4 * 4));
            if (grid == 0) PBA_FAIL(PBA_E_NOMEM, "one wavefront's parent bits exceed the traceback budget");
            DevBuf d_ids;
            const size_t need = (size_t)grid * 4 * wave_words * 4;
            if (need > ctx->scratch_bytes) {
                if (ctx->d_scratch) { HIPCHK(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_scratch); }
                ctx->d_scratch = nullptr; ctx->scratch_bytes = 0;
                HIPCHK(hipMalloc(&ctx->d_scratch, need));
                ctx->scratch_bytes = need;
            }
            uint32_t *const d_scr = (uint32_t *)ctx->d_scratch;
            const uint32_t *ids = nullptr;
            if (pass) {
                HIPCHK(hipMalloc(&d_ids.p, sizeof(uint32_t) * cnt));
                HIPCHK(hipMemcpyAsync(d_ids.p, redo.data(), sizeof(uint32_t) * cnt, hipMemcpyHostToDevice, ctx->stream));
                ids = d_ids.as<uint32_t>();
                pl.cfg.full_band = 1;
            }
            HIPCHK(hipMemsetAsync(ctx->d_queue, 0, sizeof(uint32_t), ctx->stream));
            (void)hipEventRecord(ctx->ev[pass ? 4 : 2], ctx->stream);
#define K_TRACE2(NBV, CKV)                                                                                            \
    if (vote)                                                                                                         \
        hipLaunchKernelGGL((k_vote_pairs<NBV, CKV>), dim3(grid), dim3(PBA_WAVE * 4), pl.lds * 4, ctx->stream, A->dev(), B->dev(), \
                           d_pairs.as<pba_pair>(), ids, cnt, pl.cfg, overlap_min, d_out.as<pba_result>(),              \
                           d_scr, wave_words, cap_words, vdev, vbeg, vpre, vpost, ctx->d_queue);        \
    else                                                                                                              \
        hipLaunchKernelGGL((k_trace_pairs<NBV, CKV>), dim3(grid), dim3(PBA_WAVE * 4), pl.lds * 4, ctx->stream, A->dev(), B->dev(), \
                           d_pairs.as<pba_pair>(), ids, cnt, pl.cfg, d_out.as<pba_result>(), d_scr,     \
                           wave_words, cap_words, d_ops.as<uint8_t>(), d_ooff.as<uint64_t>(), d_ne.as<int32_t>(),      \
                           ctx->d_queue)
#define K_TRACE(NBV)                                                                                                  \
    if (ck) { K_TRACE2(NBV, true); } else { K_TRACE2(NBV, false); }
            switch (nb) {
                case 1: K_TRACE(1); break;
                case 2: K_TRACE(2); break;
                case 3: K_TRACE(3); break;
                case 4: K_TRACE(4); break;
                case 6: K_TRACE(6); break;
                default: K_TRACE(8); break;
            }
#undef K_TRACE
#undef K_TRACE2
            (void)hipEventRecord(ctx->ev[pass ? 5 : 3], ctx->stream);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result) * n, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));           // the scratch area is freed at the end of this pass
            if (pass) { ctx->prof.nb_redo = (uint32_t)nb; ctx->prof.n_redo = cnt; break; }
            for (size_t q = 0; q < n; ++q)
                if (out[q].rc == PBA_RC_UNCERTIFIED) redo.push_back((uint32_t)q);
            if (redo.empty()) break;
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (!vote) {
        HIPCHK(hipMemcpyAsync(nedit, d_ne.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
        if (ops_total) HIPCHK(hipMemcpyAsync(ops + ops_off[0], d_ops.p, ops_total, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    prof_finish(ctx);
    return PBA_OK;
}

int pba_align_batch_trace(pba_ctx *ctx, const pba_seqs *A, const pba_seqs *B, const pba_pair *pairs, size_t n, double R,
                          int maxn, int maxm, int kernel, pba_result *out, uint8_t *ops, const uint64_t *ops_off,
                          int32_t *nedit) {
    return trace_batch(ctx, A, B, pairs, n, R, maxn, maxm, kernel, out, ops, ops_off, nedit, nullptr, 0);
}

}  // extern "C"
