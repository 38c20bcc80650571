// pba_cons.hip -- consensus voting and reference growth (the unlocked half of ref_seq): host side of csrc/consensus.h.
// One process per GPU, one pba_ctx per process, one HIP stream per ctx.  Everything here fails loudly
// (PBA_E_NODEVICE / PBA_E_HIP): there is no CPU path behind these entry points.
#include "pba_host.h"

extern "C" {

// ---------------------------------------------------------------------------------------------
// host API: consensus voting and reference growth (ref_seq.h, the unlocked half)
// ---------------------------------------------------------------------------------------------
struct pba_cons {
    int device;                           // (not the ctx: the object may outlive it, like pba_seqs / pba_index)
    int max_len, beg, end, pre, post;     // as in ref_seq (ref_seq.h:364-368), indices into the 3*max_len arrays
    int cur;                              // which of the two array sets is live (evolve ping-pongs)
    ConsDev set[2];
    int *d_n;
    int vote_ext;                         // votes of the running batch address the text of [pre, post) (pba_cons_round), not [beg, end)
};

static void cons_free_sets(pba_cons *c) {
    for (int k = 0; k < 2; ++k) {
        if (c->set[k].sel) (void)hipFree(c->set[k].sel);
        if (c->set[k].sup) (void)hipFree(c->set[k].sup);
        if (c->set[k].tot) (void)hipFree(c->set[k].tot);
        if (c->set[k].txt) (void)hipFree(c->set[k].txt);
    }
    if (c->d_n) (void)hipFree(c->d_n);
}

static int cons_fill(pba_ctx *ctx, pba_cons *c, int first, const char *text, int len, int weight) {
    if (len <= 0) return PBA_OK;
    DevBuf d_text;
    HIPCHK(hipMalloc(&d_text.p, (size_t)len));
    HIPCHK(hipMemcpyAsync(d_text.p, text, (size_t)len, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_cons_fill, dim3((uint32_t)((len + 255) / 256)), dim3(256), 0, ctx->stream, c->set[c->cur], first,
                       len, d_text.as<char>(), weight);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

int pba_cons_create(pba_ctx *ctx, const char *text, int len, int weight, int max_len, pba_cons **out) {
    if (!ctx || !out || len < 0 || (!text && len) || max_len < 1 || len > max_len || weight < 0 || weight > 0xFFFF)
        return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    pba_cons *c = new (std::nothrow) pba_cons();
    if (!c) return PBA_E_NOMEM;
    memset(c, 0, sizeof *c);
    c->device = ctx->device; c->max_len = max_len;
    c->beg = c->pre = max_len; c->end = c->post = max_len + len;
    const size_t cap = (size_t)3 * max_len + 64;
    bool ok = hipMalloc((void **)&c->d_n, sizeof(int)) == hipSuccess;
    for (int k = 0; k < 2 && ok; ++k)
        ok = hipMalloc((void **)&c->set[k].sel, cap * 8) == hipSuccess && hipMalloc((void **)&c->set[k].sup, cap * 8) == hipSuccess &&
             hipMalloc((void **)&c->set[k].tot, cap * 4) == hipSuccess && hipMalloc((void **)&c->set[k].txt, cap) == hipSuccess;
    if (!ok) { cons_free_sets(c); delete c; PBA_FAIL(PBA_E_NOMEM, "pba_cons_create"); }
    int st = cons_fill(ctx, c, c->beg, text, len, weight);                    // ref_seq.h:218-225
    if (st != PBA_OK) { cons_free_sets(c); delete c; return st; }
    *out = c;
    return PBA_OK;
}

void pba_cons_destroy(pba_cons *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    cons_free_sets(c);                    // hipFree waits for the work that uses the buffers
    delete c;
}

int pba_cons_extent(const pba_cons *c, int32_t *extent) {
    if (!c || !extent) return PBA_E_INVALID;
    extent[0] = c->pre - c->beg; extent[1] = c->post - c->beg; extent[2] = c->end - c->beg;
    return PBA_OK;
}

int pba_cons_append(pba_ctx *ctx, pba_cons *c, const char *seg, int len) {        // ref_seq.h:227-233
    if (!ctx || !c || len < 0 || (!seg && len)) return PBA_E_INVALID;
    if ((long long)c->post + len > 3ll * c->max_len) PBA_FAIL(PBA_E_TOOLONG, "pba_cons_append: reference grew past 2*max_len");
    HIPCHK(hipSetDevice(ctx->device));
    int st = cons_fill(ctx, c, c->post, seg, len, 1);
    if (st == PBA_OK) c->post += len;
    return st;
}

int pba_cons_prepend(pba_ctx *ctx, pba_cons *c, const char *seg, int len) {       // ref_seq.h:235-242
    if (!ctx || !c || len < 0 || (!seg && len)) return PBA_E_INVALID;
    if (c->pre - len < 0) PBA_FAIL(PBA_E_TOOLONG, "pba_cons_prepend: reference grew past max_len before its origin");
    HIPCHK(hipSetDevice(ctx->device));
    int st = cons_fill(ctx, c, c->pre - len, seg, len, 1);
    if (st == PBA_OK) c->pre -= len;
    return st;
}

int pba_cons_elect(pba_ctx *ctx, pba_cons *c, uint32_t n, const int32_t *pos, const uint8_t *fwd, const uint8_t *ops,
                   const char *vals, const uint64_t *ops_off, const int32_t *nedit) {
    if (!ctx || !c || (n && (!pos || !fwd || !ops || !vals || !ops_off || !nedit))) return PBA_E_INVALID;
    if (n == 0) return PBA_OK;
    for (uint32_t q = 0; q < n; ++q)
        if (nedit[q] < 0 || ops_off[q + 1] < ops_off[q] || (uint64_t)nedit[q] > ops_off[q + 1] - ops_off[q] ||
            c->beg + pos[q] < c->pre || c->beg + pos[q] >= c->post)              // "pos should be contained", ref_seq.h:351
            PBA_FAIL(PBA_E_INVALID, "pba_cons_elect: script outside its slot or position outside the reference");
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t total = ops_off[n] - ops_off[0];
    DevBuf d_pos, d_fwd, d_ops, d_vals, d_off, d_ne;
    HIPCHK(hipMalloc(&d_pos.p, sizeof(int32_t) * n));
    HIPCHK(hipMalloc(&d_fwd.p, n));
    HIPCHK(hipMalloc(&d_ops.p, total + 16));
    HIPCHK(hipMalloc(&d_vals.p, total + 16));
    HIPCHK(hipMalloc(&d_off.p, sizeof(uint64_t) * (n + 1)));
    HIPCHK(hipMalloc(&d_ne.p, sizeof(int32_t) * n));
    std::vector<uint64_t> rel(n + 1);
    for (uint32_t q = 0; q <= n; ++q) rel[q] = ops_off[q] - ops_off[0];
    HIPCHK(hipMemcpyAsync(d_pos.p, pos, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_fwd.p, fwd, n, hipMemcpyHostToDevice, ctx->stream));
    if (total) {
        HIPCHK(hipMemcpyAsync(d_ops.p, ops + ops_off[0], total, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(d_vals.p, vals + ops_off[0], total, hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(hipMemcpyAsync(d_off.p, rel.data(), sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_ne.p, nedit, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_cons_elect, dim3(n), dim3(PBA_WAVE), 0, ctx->stream, c->set[c->cur], c->beg, c->pre, c->post, n,
                       d_pos.as<int>(), d_fwd.as<uint8_t>(), d_ops.as<uint8_t>(), d_vals.as<char>(),
                       d_off.as<unsigned long long>(), d_ne.as<int>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

int cons_vote_view(const pba_cons *c, ConsDev *dev, int *beg, int *pre, int *post) {
    if (!c) return PBA_E_INVALID;
    *dev = c->set[c->cur]; *beg = c->vote_ext ? c->pre : c->beg; *pre = c->pre; *post = c->post;
    return PBA_OK;
}

// ref_seq::try_align's align + gate + elect (ref_seq.h:264-267) for a batch of pairs whose `a` is the reference:
// sweep, walk and vote on the device, no script in memory.  No growth (append / prepend are the caller's, after the
// batch): the batch form of a round of interior reads.
int pba_cons_vote_pairs(pba_ctx *ctx, pba_cons *c, const pba_seqs *A, uint32_t ref_seq, const pba_seqs *B,
                        const pba_pair *pairs, size_t n, double R, int maxn, int maxm, int overlap_min, pba_result *out) {
    if (!ctx || !c || !A || !B || (!pairs && n) || (!out && n) || ref_seq >= A->n) return PBA_E_INVALID;
    if ((int)A->h_len[ref_seq] != c->end - c->beg) PBA_FAIL(PBA_E_INVALID, "pba_cons_vote_pairs: A[ref_seq] is not the reference of these boxes");
    for (size_t q = 0; q < n; ++q) {
        const bool ab = (pairs[q].flags & PBA_A_BACKWARD) != 0, bb = (pairs[q].flags & PBA_B_BACKWARD) != 0;
        if (pairs[q].a_seq != ref_seq || ab != bb)               // try_align walks both accessors the same way (ref_seq.h:260-261)
            PBA_FAIL(PBA_E_INVALID, "pba_cons_vote_pairs: a must be the reference, both accessors in one direction");
    }
    return trace_batch(ctx, A, B, pairs, n, R, maxn, maxm, PBA_KERNEL_BITVEC, out, nullptr, nullptr, nullptr, c, overlap_min);
}

// the text of boxes [first, first+len) as a one-sequence set (packed on the device from the object's own text array)
static int cons_text_seqs(pba_ctx *ctx, const pba_cons *c, int first, int len, pba_seqs **out) {
    DevBuf d_offs;
    const uint64_t offs[2] = {0, (uint64_t)len};
    HIPCHK(hipMalloc(&d_offs.p, sizeof offs));
    HIPCHK(hipMemcpyAsync(d_offs.p, offs, sizeof offs, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return pba_seqs_from_device_text(ctx, c->set[c->cur].txt + first, d_offs.p, 1, (uint64_t)len, (uint32_t)len, out);
}

// One round of spaced_seed.cpp:420-446 against an UNLOCKED reference: the reads of `pool`, in order, each stopping at its
// first success -- and every success votes (elect) and may grow the text (ref_seq.h:259-276), which the reads after it
// then see.  Votes never change an alignment inside a round (the text changes in evolve), growth does, but only for a
// candidate whose reference accessor is shorter than len_b + max_dst (seq_aligner.h:94-102): it `touches` an end.
// So the round runs as a few batches: all pending reads are walked at once against the text as it stands
// (k_spaced_round reports which ends each read's candidates touched); the rows are then taken in pool order, and a row
// is the reference's as long as no earlier read of the batch has grown -- or been put back for -- an end it touches.
// The accepted successes vote from their traceback walk (k_vote_pairs) against the batch's text, then the growths are
// applied, and what was put back is the next batch.  The first pending read is always accepted, so it ends; a batch
// takes at most one growth per end.
int pba_cons_round(pba_ctx *ctx, pba_cons *c, const pba_seqs *reads, const uint32_t *pool, uint32_t n_pool, uint32_t mask,
                   double R, int max_trial, int overlap_min, int buggy_seed_at, int kernel, int maxn, int maxm,
                   pba_ss_row *rows, pba_cons_round_stats *stats) {
    if (!ctx || !c || !reads || (!pool && n_pool) || !rows || max_trial < 0) return PBA_E_INVALID;
    for (uint32_t k = 0; k < n_pool; ++k)
        if (pool[k] >= reads->n) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    pba_cons_round_stats S;
    memset(&S, 0, sizeof S);
    pba_seqs *base = nullptr;
    pba_index *ix = nullptr;
    int st = cons_text_seqs(ctx, c, c->beg, c->end - c->beg, &base);                 // get_seedmap reads [beg, end), ref_seq.h:291-311
    if (st == PBA_OK) st = pba_index_build(ctx, base, 0, mask, PBA_INDEX_HEAD_TAIL, &ix);
    if (st != PBA_OK) { pba_seqs_destroy(base); return st; }
    S.n_index = (uint32_t)pba_index_entries(ix);
    std::vector<uint32_t> pending(pool, pool + n_pool), deferred;
    std::vector<uint8_t> touch(reads->n);
    std::vector<char> rtext;
    struct Growth { uint32_t read; bool fwd; int j, matlen_b; };
    while (st == PBA_OK && !pending.empty()) {
        ++S.n_batches;
        pba_seqs *ext = base;
        if (c->pre != c->beg || c->post != c->end) st = cons_text_seqs(ctx, c, c->pre, c->post - c->pre, &ext);
        if (st != PBA_OK) break;
        const int org = c->beg - c->pre, post_rel = c->post - c->beg, pre_rel = c->pre - c->beg;
        st = spaced_round_subset(ctx, ix, ext, 0, reads, R, max_trial, overlap_min, buggy_seed_at, kernel, pending.data(),
                                 (uint32_t)pending.size(), rows, org, maxn, maxm, touch.data());
        std::vector<pba_pair> vp;
        std::vector<uint32_t> vread;
        std::vector<Growth> grow;
        bool dirty_post = false, dirty_pre = false;
        deferred.clear();
        for (size_t k = 0; st == PBA_OK && k < pending.size(); ++k) {
            const uint32_t r = pending[k];
            if (((touch[r] & 1) && dirty_post) || ((touch[r] & 2) && dirty_pre)) {
                deferred.push_back(r);               // what it does once it is re-walked is unknown: it may grow either end
                dirty_post = dirty_pre = true;
                continue;
            }
            const pba_ss_row &w = rows[r];
            if (!w.found) continue;
            const bool fwd = w.dir == 1;
            const int slen = (int)reads->h_len[r], s_len = slen - w.j;                  // spaced_seed.cpp:274-275, both directions
            const int r_off = fwd ? w.ref_pos : w.ref_pos + 15;                         // spaced_seed.cpp:285
            const int la = fwd ? post_rel - r_off : r_off - pre_rel + 1;                // get_accessor, ref_seq.h:284-285
            pba_pair pr;
            memset(&pr, 0, sizeof pr);
            pr.a_seq = 0; pr.a_pos = r_off + org; pr.a_len = la;
            pr.b_seq = r; pr.b_pos = fwd ? w.j : slen - w.j - 1; pr.b_len = s_len;
            pr.flags = fwd ? 0u : (PBA_A_BACKWARD | PBA_B_BACKWARD);
            vp.push_back(pr); vread.push_back(r);
            ++S.n_found;
            if (w.matlen_a == la) {                                                     // ref_seq.h:268
                grow.push_back(Growth{r, fwd, w.j, w.matlen_b});
                if (fwd) dirty_post = true; else dirty_pre = true;
            }
        }
        if (st == PBA_OK && !vp.empty()) {                                              // elect, ref_seq.h:267
            std::vector<pba_result> out(vp.size());
            c->vote_ext = 1;
            st = trace_batch(ctx, ext, reads, vp.data(), vp.size(), R, maxn, maxm, PBA_KERNEL_BITVEC, out.data(), nullptr, nullptr,
                             nullptr, c, overlap_min);
            c->vote_ext = 0;
            for (size_t q = 0; st == PBA_OK && q < vp.size(); ++q) {
                const pba_ss_row &w = rows[vread[q]];
                if (out[q].rc < 0 || out[q].cost != w.cost || out[q].matlen_a != w.matlen_a || out[q].matlen_b != w.matlen_b) {
                    snprintf(ctx->err, sizeof ctx->err, "pba_cons_round: the voting walk of read %u disagrees with its round row", vread[q]);
                    st = PBA_E_HIP;
                }
            }
        }
        for (size_t g = 0; st == PBA_OK && g < grow.size(); ++g) {                      // ref_seq.h:268-275
            const Growth &G = grow[g];
            const int slen = (int)reads->h_len[G.read], add = (slen - G.j) - G.matlen_b;
            rtext.resize((size_t)slen + 1);
            st = pba_seqs_get_text(ctx, reads, G.read, rtext.data(), rtext.size());
            if (st != PBA_OK) break;
            if (G.fwd) { st = pba_cons_append(ctx, c, rtext.data() + G.j + G.matlen_b, add); ++S.n_grown_fwd; }
            else { st = pba_cons_prepend(ctx, c, rtext.data(), add); ++S.n_grown_bwd; }
        }
        if (ext != base) pba_seqs_destroy(ext);
        S.n_deferred += (uint32_t)deferred.size();
        pending.swap(deferred);
    }
    pba_index_destroy(ix);
    pba_seqs_destroy(base);
    if (stats) *stats = S;
    return st;
}

// spaced_seed's main loop (spaced_seed.cpp:409-452) without -l: rounds of pba_cons_round over the reads not found yet,
// seeds drawn as in pba_spaced_multi, evolve after every round that does not end the loop.
int pba_cons_assemble(pba_ctx *ctx, pba_cons *c, const pba_seqs *reads, double R, int max_trial, int overlap_min,
                      int buggy_seed_at, int kernel, int maxn, int maxm, const uint32_t *masks, int n_masks,
                      const uint32_t *picks, int n_picks, int max_round, pba_ss_row *rows, int32_t *found_round,
                      pba_ss_round_log *log, int32_t *ref_len_log, int log_cap, int *n_rounds) {
    if (!ctx || !c || !reads || !masks || n_masks < 1 || !picks || n_picks < 1 || max_round < 0 || !rows || !found_round ||
        !n_rounds || log_cap < 0 || ((!log || !ref_len_log) && log_cap))
        return PBA_E_INVALID;
    const uint32_t n = reads->n;
    std::vector<uint32_t> pool(n);
    for (uint32_t r = 0; r < n; ++r) { pool[r] = r; found_round[r] = 0; memset(&rows[r], 0, sizeof rows[r]); rows[r].read = (int32_t)r; rows[r].j = -1; }
    int nfailure = 0, draws = 0, done = 0;
    for (int nround = 1; nround <= max_round; ++nround) {
        const uint32_t mask = nfailure == 0 ? masks[picks[draws++ % n_picks] % (uint32_t)n_masks] : masks[nfailure - 1];   // :412
        pba_cons_round_stats S;
        int st = pba_cons_round(ctx, c, reads, pool.data(), (uint32_t)pool.size(), mask, R, max_trial, overlap_min, buggy_seed_at,
                                kernel, maxn, maxm, rows, &S);
        if (st != PBA_OK) return st;
        std::vector<uint32_t> rest;
        rest.reserve(pool.size());
        for (uint32_t r : pool) {
            if (rows[r].found) found_round[r] = nround;                                     // erased from the pool, :443
            else rest.push_back(r);
        }
        if (done < log_cap) { log[done].round = nround; log[done].mask = mask; log[done].n_tried = (int32_t)pool.size(); log[done].n_found = S.n_found; }
        pool.swap(rest);
        bool last = false;
        if (S.n_found != 0) nfailure = 0;                                                   // :448-449
        else if (++nfailure == n_masks) last = true;                                        // :450: break before evolve
        if (!last) {
            int32_t new_len = 0;
            st = pba_cons_evolve(ctx, c, nullptr, 0, &new_len);                             // :451
            if (st != PBA_OK) return st;
            if (done < log_cap) ref_len_log[done] = new_len;
        } else if (done < log_cap) ref_len_log[done] = c->post - c->pre;
        ++done;
        if (last) break;
    }
    *n_rounds = done;
    return PBA_OK;
}

int pba_cons_evolve(pba_ctx *ctx, pba_cons *c, char *text_out, int cap, int32_t *new_len) {   // ref_seq.h:317-349
    if (!ctx || !c || !new_len || cap < 0 || (!text_out && cap)) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    const int nxt = c->cur ^ 1;
    hipLaunchKernelGGL(k_cons_evolve, dim3(1), dim3(1024), 0, ctx->stream, c->set[c->cur], c->set[nxt], c->pre, c->post,
                       c->max_len, c->d_n);
    HIPCHK(hipGetLastError());
    int n = 0;
    HIPCHK(hipMemcpyAsync(&n, c->d_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    c->cur = nxt;
    c->beg = c->pre = c->max_len;
    c->end = c->post = c->max_len + n;
    *new_len = n;
    const int ncopy = std::min(n, cap);
    if (ncopy > 0) {
        HIPCHK(hipMemcpyAsync(text_out, c->set[c->cur].txt + c->beg, (size_t)ncopy, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return PBA_OK;
}

int pba_cons_dump(pba_ctx *ctx, const pba_cons *c, uint16_t *sel, uint16_t *sup, int32_t *tot, int cap, int32_t *n) {
    if (!ctx || !c || !n || cap < 0 || (cap && (!sel || !sup || !tot))) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    *n = c->post - c->pre;
    const int k = std::min(*n, cap);
    if (k > 0) {
        const ConsDev &d = c->set[c->cur];
        HIPCHK(hipMemcpyAsync(sel, d.sel + c->pre, (size_t)k * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(sup, d.sup + c->pre, (size_t)k * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(tot, d.tot + c->pre, (size_t)k * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return PBA_OK;
}

int pba_cons_text(pba_ctx *ctx, const pba_cons *c, char *out, int cap, int32_t *n) {
    if (!ctx || !c || !n || cap < 0 || (cap && !out)) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    *n = c->post - c->pre;
    const int k = std::min(*n, cap);
    if (k > 0) {
        HIPCHK(hipMemcpyAsync(out, c->set[c->cur].txt + c->pre, (size_t)k, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return PBA_OK;
}


}  // extern "C"
