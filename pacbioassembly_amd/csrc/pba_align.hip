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


static void tu_attrs() {
    static bool done = false;
    if (done) return;
    done = true;
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
    tu_attrs();
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

int pba_align_text(pba_ctx *ctx, const char *a, int a_fwd, int la, const char *b, int b_fwd, int lb, double R,
                   int maxn, int maxm, pba_result *out) {
    if (!ctx || !out || la < 0 || lb < 0 || (!a && la) || (!b && lb)) return PBA_E_INVALID;
    if (la > kMaxSeqLen || lb > kMaxSeqLen) PBA_FAIL(PBA_E_TOOLONG, "pba_align_text");
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs();
    Plan pl;
    int st = make_plan(ctx, R, maxn, maxm, PBA_KERNEL_ROWSWEEP, max_dst_of(la, lb, R), &pl);
    if (st != PBA_OK) return st;
    // element k of a backward accessor is p[-k]: ship [p-(len-1), p] and point at its last byte
    const size_t oa = 0, ob = ((size_t)la + 31) & ~(size_t)15;
    DevBuf buf, d_out;
    HIPCHK(hipMalloc(&buf.p, ob + lb + 32));
    HIPCHK(hipMalloc(&d_out.p, sizeof(pba_result)));
    if (la) HIPCHK(hipMemcpyAsync(buf.as<uint8_t>() + oa, a_fwd ? a : a - (la - 1), la, hipMemcpyHostToDevice, ctx->stream));
    if (lb) HIPCHK(hipMemcpyAsync(buf.as<uint8_t>() + ob, b_fwd ? b : b - (lb - 1), lb, hipMemcpyHostToDevice, ctx->stream));
    const uint8_t *da = buf.as<uint8_t>() + oa + (a_fwd || !la ? 0 : la - 1);
    const uint8_t *db = buf.as<uint8_t>() + ob + (b_fwd || !lb ? 0 : lb - 1);
    hipLaunchKernelGGL(k_align_bytes, dim3(1), dim3(PBA_WAVE), pl.lds, ctx->stream, da, a_fwd ? 1 : -1, la, db,
                       b_fwd ? 1 : -1, lb, pl.cfg, d_out.as<pba_result>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

// ---- traceback
static const uint64_t kTraceBudget = 96ull << 30;      // parent codes / bits resident for one call (and at most 80 % of free HBM)

static uint64_t par_bytes_of(int la, int lb, double R) {           // (len_a + 1) * (2*max_dst + 1), seq_aligner.h:94-102
    const int md = max_dst_of(la, lb, R);
    const int len_a = lb >= la ? la : std::min(la, lb + md);
    return ((uint64_t)len_a + 1) * (2ull * md + 1);
}

int pba_align_text_trace(pba_ctx *ctx, const char *a, int a_fwd, int la, const char *b, int b_fwd, int lb, double R,
                         int maxn, int maxm, pba_result *out, uint8_t *ops, int32_t ops_cap, int32_t *nedit) {
    if (!ctx || !out || !nedit || la < 0 || lb < 0 || (!a && la) || (!b && lb) || (!ops && ops_cap) || ops_cap < 0)
        return PBA_E_INVALID;
    if (la > kMaxSeqLen || lb > kMaxSeqLen) PBA_FAIL(PBA_E_TOOLONG, "pba_align_text_trace");
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs();
    Plan pl;
    int st = make_plan(ctx, R, maxn, maxm, PBA_KERNEL_ROWSWEEP, max_dst_of(la, lb, R), &pl);
    if (st != PBA_OK) return st;
    const uint64_t pb = par_bytes_of(la, lb, R);
    if (pb > kTraceBudget) PBA_FAIL(PBA_E_NOMEM, "parent codes exceed the traceback budget");
    const size_t oa = 0, ob = ((size_t)la + 31) & ~(size_t)15;
    DevBuf buf, d_out, d_par, d_ops, d_off, d_ne;
    HIPCHK(hipMalloc(&buf.p, ob + lb + 32));
    HIPCHK(hipMalloc(&d_out.p, sizeof(pba_result)));
    HIPCHK(hipMalloc(&d_par.p, pb + 16));
    HIPCHK(hipMalloc(&d_ops.p, (size_t)ops_cap + 16));
    HIPCHK(hipMalloc(&d_off.p, 4 * sizeof(uint64_t)));
    HIPCHK(hipMalloc(&d_ne.p, sizeof(int32_t)));
    if (la) HIPCHK(hipMemcpyAsync(buf.as<uint8_t>() + oa, a_fwd ? a : a - (la - 1), la, hipMemcpyHostToDevice, ctx->stream));
    if (lb) HIPCHK(hipMemcpyAsync(buf.as<uint8_t>() + ob, b_fwd ? b : b - (lb - 1), lb, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t offs[4] = {0, (uint64_t)ops_cap, 0, 0};          // ops_off[0..1], par_off[0]
    HIPCHK(hipMemcpyAsync(d_off.p, offs, sizeof offs, hipMemcpyHostToDevice, ctx->stream));
    const uint8_t *da = buf.as<uint8_t>() + oa + (a_fwd || !la ? 0 : la - 1);
    const uint8_t *db = buf.as<uint8_t>() + ob + (b_fwd || !lb ? 0 : lb - 1);
    hipLaunchKernelGGL(k_align_bytes_trace, dim3(1), dim3(PBA_WAVE), pl.lds, ctx->stream, da, a_fwd ? 1 : -1, la, db,
                       b_fwd ? 1 : -1, lb, pl.cfg, d_out.as<pba_result>(), d_par.as<uint8_t>(), (uint16_t *)nullptr);
    hipLaunchKernelGGL(k_trace_walk, dim3(1), dim3(64), 0, ctx->stream, d_out.as<pba_result>(), d_par.as<uint8_t>(),
                       d_off.as<uint64_t>() + 2, d_ops.as<uint8_t>(), d_off.as<uint64_t>(), d_ne.as<int32_t>(), 1u);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(nedit, d_ne.p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int32_t ncopy = std::min(*nedit, ops_cap);
    if (ncopy > 0) HIPCHK(hipMemcpyAsync(ops, d_ops.p, (size_t)ncopy, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
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
    if (la > kMaxSeqLen || lb > kMaxSeqLen) PBA_FAIL(PBA_E_TOOLONG, "pba_align_text_matrix");
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs();
    Plan pl;
    int st = make_plan(ctx, R, maxn, maxm, PBA_KERNEL_ROWSWEEP, max_dst_of(la, lb, R), &pl);
    if (st != PBA_OK) return st;
    const int md = max_dst_of(la, lb, R);
    const uint64_t pb = par_bytes_of(la, lb, R);                  // cells: (len_a + 1) * (2*max_dst + 1)
    if (pb * 3 > kTraceBudget) PBA_FAIL(PBA_E_NOMEM, "the matrix exceeds the traceback budget");
    if (cap_cells < pb) PBA_FAIL(PBA_E_INVALID, "pba_align_text_matrix: cost / parent hold fewer than (len_a + 1) * (2*max_dst + 1) cells");
    const size_t oa = 0, ob = ((size_t)la + 31) & ~(size_t)15;
    DevBuf buf, d_out, d_par, d_cst;
    HIPCHK(hipMalloc(&buf.p, ob + lb + 32));
    HIPCHK(hipMalloc(&d_out.p, sizeof(pba_result)));
    HIPCHK(hipMalloc(&d_par.p, pb + 16));
    HIPCHK(hipMalloc(&d_cst.p, 2 * pb + 16));
    HIPCHK(hipMemsetAsync(d_par.p, 0, pb, ctx->stream));
    HIPCHK(hipMemsetAsync(d_cst.p, 0xFF, 2 * pb, ctx->stream));
    if (la) HIPCHK(hipMemcpyAsync(buf.as<uint8_t>() + oa, a_fwd ? a : a - (la - 1), la, hipMemcpyHostToDevice, ctx->stream));
    if (lb) HIPCHK(hipMemcpyAsync(buf.as<uint8_t>() + ob, b_fwd ? b : b - (lb - 1), lb, hipMemcpyHostToDevice, ctx->stream));
    const uint8_t *da = buf.as<uint8_t>() + oa + (a_fwd || !la ? 0 : la - 1);
    const uint8_t *db = buf.as<uint8_t>() + ob + (b_fwd || !lb ? 0 : lb - 1);
    hipLaunchKernelGGL(k_align_bytes_trace, dim3(1), dim3(PBA_WAVE), pl.lds, ctx->stream, da, a_fwd ? 1 : -1, la, db,
                       b_fwd ? 1 : -1, lb, pl.cfg, d_out.as<pba_result>(), d_par.as<uint8_t>(), d_cst.as<uint16_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(cost, d_cst.p, 2 * pb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(parent, d_par.p, pb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // init_cell (seq_aligner.h:139-150): row 0 is D(0,j) = j, INSERT, for j <= max_dst; the size guard leaves everything unwritten
    const uint64_t W = 2ull * md + 1;
    int swept = 0;
    if (!(maxn > 0 && (out->len_a >= maxn + maxm || md >= maxm))) {
        for (int j = 0; j <= md; ++j) { cost[(uint64_t)md + j] = (uint16_t)j; parent[(uint64_t)md + j] = j ? 2 : 0; }
        // rows swept: every row up to len_a, or up to the early failure -- the last row whose diagonal-side cell was written
        for (swept = out->len_a; swept > 0; --swept) {
            const int jlo = swept - md > 0 ? swept - md : 0;
            if (cost[(uint64_t)swept * W + (uint64_t)(jlo - swept + md)] != 0xFFFF) break;
        }
    }
    if (rows_swept) *rows_swept = swept;
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
    tu_attrs();
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
