// pba_drivers.hip -- the reference's ordered first-success loops (locator.cpp:70-92, spaced_seed.cpp:420-437) on the GPU.
// One process per GPU, one pba_ctx per process, one HIP stream per ctx.  Everything here fails loudly
// (PBA_E_NODEVICE / PBA_E_HIP): there is no CPU path behind these entry points.
#include "pba_host.h"
#include "prefilter.h"

// ---------------------------------------------------------------------------------------------
// kernels: drivers.  One wavefront per read walks the reference's ordered candidate loop and
// stops at the first success, so the pairs it aligns are exactly the pairs the reference aligns.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ long long pair_cells(const AlnOut &o) {
    return band_cells(o.len_b, o.max_dst, o.fail_row ? o.fail_row : o.len_a);
}

// per-read side outputs of k_locate
struct LocAux {
    long long cells;     // band cells the reference would evaluate for this read
    int probe_hits;      // probes that found their key
    int redo;            // 1: a pair came back PBA_RC_UNCERTIFIED, the read must be re-run at full band
};

// locator.cpp:70-92
template <int NB>
__global__ void __launch_bounds__(PBA_WAVE * Wpb<NB>::v, Wpb<NB>::occ)
k_locate(IndexDev ix, SeqSetDev T, uint32_t tseq, SeqSetDev Rd, const uint32_t *ids, uint32_t n, int trials,
         int min_len, AlignCfg cfg, pba_loc_row *rows, LocAux *aux, uint32_t *queue) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    __shared__ int2 s_grp[Wpb<NB>::v][PBA_WAVE];      // the hit group in flight: position, band cells of a hit the prefilter failed
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / PBA_WAVE));   // wave-uniform on purpose: keeps the walk in SGPRs
    uint8_t *lds = lds_all + (size_t)wave * cfg.row_cap * 2;
    const PreThresholds pre_t(cfg.R);
    for (;;) {                                // persistent wavefront: pull the next read until the queue is dry
    const uint32_t slot = next_slot(queue);
    if (slot >= n) break;
    // (what a read's walk works with is the same in every lane; saying so keeps it in scalar registers instead of vector
    // registers that the aligner's state then pushes out to scratch memory)
    const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ids ? ids[slot] : slot));
    const int len = __builtin_amdgcn_readfirstlane((int)Rd.len[r]);
    int found = 0, fj = -1, fpos = -1, fcost = -1, fma = 0, fmb = 0, fdiag = -1, npairs = 0, nhit = 0, redo = 0;
    long long ncell = 0;
    if (len >= min_len) {                                                   // locator.cpp:72
        const PackedFetch rbase = fetch_of(Rd, r, 0, 1), tbase = fetch_of_uniform(T, tseq, 0, 1);
        const uint8_t *rseq = rbase.seq;
        const int clen = __builtin_amdgcn_readfirstlane((int)T.len[tseq]);
        for (int j = 0; j < trials && j < len && !found && !redo; ++j) {    // locator.cpp:74
            const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)(window_key(rseq, (uint32_t)j, (uint32_t)len) & ix.mask));   // locator.cpp:75
            if (key == 0) continue;                                         // never inserted, locator.cpp:64
            uint32_t beg, cnt;
            ix_find(ix, key, beg, cnt);                                     // locator.cpp:76
            beg = (uint32_t)__builtin_amdgcn_readfirstlane((int)beg);
            cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
            if (cnt == 0) continue;
            ++nhit;
            // locator.cpp:79, 64 hits at a time: every lane runs the first 32 rows of its hit (prefilter.h), then the
            // hits are walked in list order -- the ones that failed there are done, the others get the wavefront
            for (uint32_t h0 = 0; h0 < cnt && !found && !redo; h0 += PBA_WAVE) {
                const uint32_t lane = threadIdx.x & (PBA_WAVE - 1), ng = min((uint32_t)PBA_WAVE, cnt - h0);
                const bool act = lane < ng;
                const int mypos = act ? ix_pos_of(ix, (uint32_t)ix.ent[beg + h0 + lane]) : 0;
                int myfr = 0;
                long long mycells = 0;
                if constexpr (NB != 0) {
                    AlnOut po;
                    myfr = prefilter32(act, rbase.at(j, 1), len - j, tbase.at(mypos, 1), clen - mypos, cfg.R,
                                       cfg.maxn, cfg.maxm, pre_t, po);
                    mycells = myfr ? band_cells(po.len_b, po.max_dst, myfr) : 0;
                }
                // The group's per-lane state goes to LDS before the array takes the wavefront (kept in registers it was
                // spilled to scratch memory around every alignment): the hit position and the band cells of a hit that
                // failed in its first 32 rows (<= 32 x 9001, so a group's sum fits 32 bits) -- the hits between two
                // survivors are counted from there, in list order, up to the first success.
                const uint64_t fmask = __builtin_amdgcn_ballot_w64(myfr != 0);
                uint64_t surv = __builtin_amdgcn_ballot_w64(act && myfr == 0);
                __builtin_amdgcn_wave_barrier();                        // (the previous group's reads are done)
                s_grp[wave][lane] = make_int2(mypos, (int)mycells);
                __builtin_amdgcn_wave_barrier();
                uint32_t from = 0;
                auto count_failed = [&](uint32_t to) {                  // hits [from, to) that failed: seq_aligner.h:185
                    if (to > from) {
                        const uint64_t m = fmask & (to >= PBA_WAVE ? ~0ull : (1ull << to) - 1ull) & ~((1ull << from) - 1ull);
                        if (m) {
                            const uint32_t l = threadIdx.x & (PBA_WAVE - 1);
                            npairs += __builtin_popcountll(m);
                            ncell += (unsigned)wave_sum_i32(l >= from && l < to ? s_grp[wave][l].y : 0);
                        }
                    }
                };
                while (surv) {
                    const uint32_t hh = (uint32_t)__builtin_ctzll(surv);
                    surv &= surv - 1ull;
                    count_failed(hh);
                    from = hh + 1;
                    const int pos = __builtin_amdgcn_readfirstlane(s_grp[wave][hh].x);
                    const PackedFetch fa = rbase.at(j, 1);                  // a = read from j   (locator.cpp:78)
                    const PackedFetch fb = tbase.at(pos, 1);                // b = contig from pos (locator.cpp:80)
                    AlnOut o;
                    align_dispatch<NB>(fa, len - j, fb, clen - pos, cfg, lds, o, true);
                    if (o.rc == PBA_RC_UNCERTIFIED) { redo = 1; break; }
                    ++npairs;
                    ncell += pair_cells(o);
                    if (o.rc > 0) {                                         // locator.cpp:82
                        found = 1; fj = j; fpos = pos; fcost = o.cost; fma = o.matlen_a; fmb = o.matlen_b; fdiag = o.diag;
                        break;
                    }
                }
                if (!found && !redo) count_failed(ng);
            }
        }
    }
    if ((threadIdx.x & (PBA_WAVE - 1)) == 0) {
        pba_loc_row *row = rows + r;          // read / nseq are filled by the host
        row->found = found; row->j = fj; row->pos = fpos; row->cost = fcost;
        row->seglen = found ? len - fj : 0; row->matlen_a = fma; row->matlen_b = fmb; row->n_pairs = npairs;
        row->diag_cost = fdiag;
        aux[r].cells = ncell; aux[r].probe_hits = nhit; aux[r].redo = redo;
    }
    }
}

// one locked round of spaced_seed.cpp:420-437 (try_align :262-298, ref_seq::try_align ref_seq.h:259-265)
__device__ __forceinline__ uint32_t seed_at_dev(const uint8_t *payload, int pos, uint32_t len, int buggy) {
    if (buggy && (pos & 3) == 0) return ld_u32(payload + pos);   // dna_seq.h:64: pos used as a byte offset
    return window_key(payload, (uint32_t)pos, len);
}

struct SsState {
    int found, dir, ref_pos, cost, ma, mb, ntrials, npairs, redo;
    int touch;      // 1: a forward candidate whose reference accessor ends within reach of the alignment (its outcome depends
                    // on where the reference text ends: ref_seq.h:268 growth), 2: a backward one (where it begins)
};

template <int NB>
__device__ __forceinline__ bool ss_try(const IndexDev &ix, const PackedFetch &refb, int ref_len, int ref_org, const PackedFetch &readb,
                                       int slen, int pos, int dir, int overlap_min, int buggy, const AlignCfg &cfg,
                                       const PreThresholds &pre_t, void *lds, SsState &st) {
    const uint8_t *rseq = readb.seq;
    if (pos < 0 || pos + 16 > slen) return false;   // the reference only keeps reads > 500 bases
    const uint32_t key = seed_at_dev(rseq, pos, (uint32_t)slen, buggy) & ix.mask;   // spaced_seed.cpp:265
    if (key == 0) return false;
    uint32_t beg, cnt;
    ix_find(ix, key, beg, cnt);
    if (cnt == 0) return false;
    ++st.ntrials;
    const bool fwd = dir == 1;
    const int s_off = fwd ? pos : pos + 15;                        // spaced_seed.cpp:274
    const int s_len = fwd ? slen - s_off : s_off + 1;              // spaced_seed.cpp:275
    if (s_len < overlap_min) return false;                         // spaced_seed.cpp:280
    for (uint32_t h0 = 0; h0 < cnt; h0 += PBA_WAVE) {                 // 64 hits at a time through the prefilter, then in list order
        const uint32_t lane = threadIdx.x & (PBA_WAVE - 1), ng = min((uint32_t)PBA_WAVE, cnt - h0);
        const bool act = lane < ng;
        const int myhit = act ? ix_pos_of(ix, (uint32_t)ix.ent[beg + h0 + lane]) + ref_org : 0;   // index positions count from `beg`
        int myfr = 0;
        {   // seq_aligner.h:94-102: the accessor's own length matters only below len_b + max_dst
            const int my_rlen = fwd ? ref_len - myhit : myhit + 16;
            if (__builtin_amdgcn_ballot_w64(act && my_rlen <= s_len + 1 + (int)((double)s_len * cfg.R)) != 0ull) st.touch |= fwd ? 1 : 2;
        }
        if constexpr (NB != 0) {
            const int r_off = fwd ? myhit : myhit + 15;
            AlnOut po;
            myfr = prefilter32(act, refb.at(r_off, fwd ? 1 : -1), fwd ? ref_len - r_off : r_off + 1,
                               readb.at(s_off, fwd ? 1 : -1), s_len, cfg.R, cfg.maxn, cfg.maxm, pre_t, po);
        }
        for (uint32_t hh = 0; hh < ng; ++hh) {
            if (__builtin_amdgcn_readlane(myfr, (int)hh)) { ++st.npairs; continue; }   // failed within its first 32 rows
            const int hit = __builtin_amdgcn_readlane(myhit, (int)hh);
            const int r_off = fwd ? hit : hit + 15;                    // spaced_seed.cpp:285
            const int r_len = fwd ? ref_len - r_off : r_off + 1;       // ref_seq.h:284-285
            const PackedFetch fa = refb.at(r_off, fwd ? 1 : -1);       // a = reference (ref_seq.h:264)
            const PackedFetch fb = readb.at(s_off, fwd ? 1 : -1);
            AlnOut o;
            align_dispatch<NB>(fa, r_len, fb, s_len, cfg, lds, o);
            if (o.rc == PBA_RC_UNCERTIFIED) { st.redo = 1; return true; }
            ++st.npairs;
            if (o.rc < 0) continue;                                    // ref_seq.h:264
            if (o.matlen_a < overlap_min) continue;                    // ref_seq.h:265
            st.found = 1; st.dir = dir; st.ref_pos = hit - ref_org; st.cost = o.cost; st.ma = o.matlen_a; st.mb = o.matlen_b;
            return true;
        }
    }
    return false;
}

// ref_org: where position 0 of the index (ref_seq's `beg`) sits inside Rf[rseq_id] -- 0 for a locked reference; beg - pre
// when the text is an unlocked reference that has grown before its origin (ref_seq.h:235-242).
// redo[r]: bit 0 = re-run at the reference band, bits 2:1 = SsState::touch.
template <int NB>
__global__ void __launch_bounds__(PBA_WAVE * Wpb<NB>::v, Wpb<NB>::occ)
k_spaced_round(IndexDev ix, SeqSetDev Rf, uint32_t rseq_id, int ref_org, SeqSetDev Rd, const uint32_t *ids, uint32_t n,
               int max_trial, int overlap_min, int buggy, AlignCfg cfg, pba_ss_row *rows, int *redo, uint32_t *queue) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / PBA_WAVE));   // wave-uniform on purpose: keeps the walk in SGPRs
    uint8_t *lds = lds_all + (size_t)wave * cfg.row_cap * 2;
    const PreThresholds pre_t(cfg.R);
    for (;;) {
    const uint32_t slot = next_slot(queue);
    if (slot >= n) break;
    const uint32_t r = ids ? ids[slot] : slot;
    const PackedFetch ref = fetch_of(Rf, rseq_id, 0, 1), rseq = fetch_of(Rd, r, 0, 1);
    const int ref_len = (int)Rf.len[rseq_id];
    const int slen = (int)Rd.len[r];
    SsState st = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int fj = -1;
    for (int j = 0; j < max_trial; ++j) {                          // spaced_seed.cpp:424-426
        if (ss_try<NB>(ix, ref, ref_len, ref_org, rseq, slen, j, 1, overlap_min, buggy, cfg, pre_t, lds, st) ||
            ss_try<NB>(ix, ref, ref_len, ref_org, rseq, slen, slen - j - 16, -1, overlap_min, buggy, cfg, pre_t, lds, st)) {
            fj = j;
            break;
        }
    }
    if ((threadIdx.x & (PBA_WAVE - 1)) == 0) {
        pba_ss_row *row = rows + r;
        row->read = (int32_t)r; row->found = st.found; row->j = st.found ? fj : -1; row->dir = st.dir;
        row->ref_pos = st.ref_pos; row->cost = st.cost; row->matlen_a = st.ma; row->matlen_b = st.mb;
        row->n_trials = st.ntrials; row->n_pairs = st.npairs;
        redo[r] = st.redo | (st.touch << 1);
    }
    }
}


// (the attribute belongs to the current device: remembered per ctx, so a second ctx on another GPU sets it there too)
static void tu_attrs(pba_ctx *ctx) {
    if (ctx->attr_done & 2u) return;
    ctx->attr_done |= 2u;
    PBA_BIG_LDS(k_locate<0>);
    PBA_BIG_LDS(k_spaced_round<0>);
}

extern "C" {

// ---------------------------------------------------------------------------------------------
// host API: drivers
// ---------------------------------------------------------------------------------------------
int pba_locate(pba_ctx *ctx, const pba_index *ix, const pba_seqs *target, uint32_t target_seq, const pba_seqs *reads,
               double R, int trials, int min_len, int maxn, int maxm, int kernel, pba_loc_row *rows,
               pba_loc_stats *stats) {
    if (!ctx || !ix || !target || !reads || !rows || target_seq >= target->n || trials < 0) return PBA_E_INVALID;
    if (ix->mode != PBA_INDEX_ALL || ix->seq_len != target->h_len[target_seq])
        PBA_FAIL(PBA_E_INVALID, "pba_locate needs a PBA_INDEX_ALL index of the target sequence");
    if (reads->max_len > (uint32_t)kMaxSeqLen) PBA_FAIL(PBA_E_TOOLONG, "read longer than the engine limit");
    if (target->non_acgt || reads->non_acgt)   // locator.cpp compares raw bytes (an 'N' only matches an 'N'); codes would match it to T
        PBA_FAIL(PBA_E_ALPHABET, "pba_locate: a sequence set holds bytes outside ACGT");
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs(ctx);
    const uint32_t n = reads->n;
    Plan pl;
    int st = make_plan(ctx, R, maxn, maxm, kernel, 1 + (int)(reads->max_len * R), &pl);
    if (st != PBA_OK) return st;
    BufRef d_rows, d_aux, d_ids;                               // (pooled in the ctx: two hipMalloc / hipFree pairs were 1 ms of a 50 ms step)
    POOL(POOL_LOC_ROWS, sizeof(pba_loc_row) * ((size_t)n + 1), d_rows.p);
    POOL(POOL_LOC_AUX, sizeof(LocAux) * ((size_t)n + 1), d_aux.p);
    std::vector<LocAux> aux(n + 1);
#define K_LOC(NBV)                                                                                                   \
    (void)hipMemsetAsync(ctx->d_queue, 0, 4, ctx->stream);                                                           \
    hipLaunchKernelGGL(k_locate<NBV>, dim3(persistent_grid(ctx, cnt, Wpb<NBV>::v, pl.lds)),                           \
                       dim3(PBA_WAVE * Wpb<NBV>::v), pl.lds * Wpb<NBV>::v, ctx->stream, ix->dev(), target->dev(),     \
                       target_seq, reads->dev(), ids, cnt, trials, min_len, pl.cfg, d_rows.as<pba_loc_row>(),        \
                       d_aux.as<LocAux>(), ctx->d_queue)
    if (n) {
        const uint32_t cnt = n;
        const uint32_t *ids = nullptr;
        (void)hipEventRecord(ctx->ev[2], ctx->stream);
        PBA_DISPATCH_NB(pl.nb1, K_LOC);
        (void)hipEventRecord(ctx->ev[3], ctx->stream);
        ctx->prof.nb_first = (uint32_t)pl.nb1; ctx->prof.n_first = cnt; ctx->prof.nb_redo = 0; ctx->prof.n_redo = 0;
        ctx->prof.align_redo_ms = 0.f;
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(aux.data(), d_aux.p, sizeof(LocAux) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::vector<uint32_t> redo;
        for (uint32_t r = 0; r < n; ++r)
            if (aux[r].redo) redo.push_back(r);
        if (!redo.empty()) {      // reads with an uncertified pair: walk them again at the reference band
            POOL(POOL_LOC_IDS, sizeof(uint32_t) * redo.size(), d_ids.p);
            HIPCHK(hipMemcpyAsync(d_ids.p, redo.data(), sizeof(uint32_t) * redo.size(), hipMemcpyHostToDevice, ctx->stream));
            pl.cfg.full_band = 1;
            const uint32_t cnt = (uint32_t)redo.size();
            const uint32_t *ids = d_ids.as<uint32_t>();
            (void)hipEventRecord(ctx->ev[4], ctx->stream);
            PBA_DISPATCH_NB(pl.nb2, K_LOC);
            (void)hipEventRecord(ctx->ev[5], ctx->stream);
            ctx->prof.nb_redo = (uint32_t)pl.nb2; ctx->prof.n_redo = cnt;
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(aux.data(), d_aux.p, sizeof(LocAux) * n, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHK(hipMemcpyAsync(rows, d_rows.p, sizeof(pba_loc_row) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
#undef K_LOC
    if (n) prof_finish(ctx);
    pba_loc_stats s = {0, 0, 0, 0, 0};
    int nseq = 0;
    for (uint32_t r = 0; r < n; ++r) {
        rows[r].read = (int32_t)r;
        rows[r].nseq = (int)reads->h_len[r] < min_len ? -1 : nseq++;      // locator.cpp:72,91
        if (rows[r].nseq >= 0) ++s.n_reads_kept;
        s.n_pairs += rows[r].n_pairs;
        s.n_located += rows[r].found;
        s.n_probe_hits += aux[r].probe_hits;
        s.n_cells += aux[r].cells;
    }
    if (stats) *stats = s;
    return PBA_OK;
}

// One locked round over the reads `subset` (host ids; nullptr = every read).  rows is indexed by read id: rows of
// reads outside the subset are left untouched.
// Unlocked rounds (pba_cons_round) pass the grown text: ref_org = the index's position 0 inside it, maxn / maxm = the
// size guard of the caller's t_aligner, touch[read id] = SsState::touch of the reads walked.
int spaced_round_subset(pba_ctx *ctx, const pba_index *ix, const pba_seqs *ref, uint32_t ref_seq, const pba_seqs *reads,
                        double R, int max_trial, int overlap_min, int buggy_seed_at, int kernel,
                        const uint32_t *subset, uint32_t n_subset, pba_ss_row *rows, int ref_org, int maxn, int maxm,
                        uint8_t *touch) {
    if (!ctx || !ix || !ref || !reads || !rows || ref_seq >= ref->n || max_trial < 0) return PBA_E_INVALID;
    if (ix->mode != PBA_INDEX_HEAD_TAIL || (!touch && ix->seq_len != ref->h_len[ref_seq]) || ref_org < 0 ||
        (uint64_t)ref_org + ix->seq_len > ref->h_len[ref_seq])
        PBA_FAIL(PBA_E_INVALID, "pba_spaced_round needs a PBA_INDEX_HEAD_TAIL index of the reference sequence");
    if (reads->max_len > (uint32_t)kMaxSeqLen || ref->h_len[ref_seq] > 0x7FFFFFF0u)
        PBA_FAIL(PBA_E_TOOLONG, "sequence longer than the engine limit");
    if (ref->non_acgt || reads->non_acgt) PBA_FAIL(PBA_E_ALPHABET, "pba_spaced_round: a sequence set holds bytes outside ACGT");
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs(ctx);
    const uint32_t n = reads->n, n_first = subset ? n_subset : n;
    Plan pl;
    // a = reference window, b = read window: the shorter side bounds max_dst (seq_aligner.h:94-102)
    int st = make_plan(ctx, R, maxn, maxm, kernel, 1 + (int)(reads->max_len * R), &pl);
    if (st != PBA_OK) return st;
    DevBuf d_rows, d_redo, d_ids, d_sub;
    HIPCHK(hipMalloc(&d_rows.p, sizeof(pba_ss_row) * (n + 1)));
    HIPCHK(hipMalloc(&d_redo.p, sizeof(int) * (n + 1)));
    HIPCHK(hipMemsetAsync(d_redo.p, 0, sizeof(int) * (n + 1), ctx->stream));
    if (subset && n_subset) {
        HIPCHK(hipMalloc(&d_sub.p, sizeof(uint32_t) * n_subset));
        HIPCHK(hipMemcpyAsync(d_sub.p, subset, sizeof(uint32_t) * n_subset, hipMemcpyHostToDevice, ctx->stream));
    }
#define K_SS(NBV)                                                                                                    \
    (void)hipMemsetAsync(ctx->d_queue, 0, 4, ctx->stream);                                                           \
    hipLaunchKernelGGL(k_spaced_round<NBV>, dim3(persistent_grid(ctx, cnt, Wpb<NBV>::v, pl.lds)),                     \
                       dim3(PBA_WAVE * Wpb<NBV>::v), pl.lds * Wpb<NBV>::v, ctx->stream, ix->dev(), ref->dev(),       \
                       ref_seq, ref_org, reads->dev(), ids, cnt, max_trial, overlap_min, buggy_seed_at, pl.cfg,      \
                       d_rows.as<pba_ss_row>(), d_redo.as<int>(), ctx->d_queue)
    if (n_first) {
        std::vector<int> h_redo(n);
        std::vector<pba_ss_row> h_rows(n);
        const uint32_t cnt = n_first;
        const uint32_t *ids = subset ? d_sub.as<uint32_t>() : nullptr;
        (void)hipEventRecord(ctx->ev[2], ctx->stream);
        PBA_DISPATCH_NB(pl.nb1, K_SS);
        (void)hipEventRecord(ctx->ev[3], ctx->stream);
        ctx->prof.nb_first = (uint32_t)pl.nb1; ctx->prof.n_first = cnt; ctx->prof.nb_redo = 0; ctx->prof.n_redo = 0;
        ctx->prof.align_redo_ms = 0.f;
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_redo.data(), d_redo.p, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::vector<uint32_t> redo;
        for (uint32_t r = 0; r < n; ++r)
            if (h_redo[r] & 1) redo.push_back(r);
        if (!redo.empty()) {
            HIPCHK(hipMalloc(&d_ids.p, sizeof(uint32_t) * redo.size()));
            HIPCHK(hipMemcpyAsync(d_ids.p, redo.data(), sizeof(uint32_t) * redo.size(), hipMemcpyHostToDevice, ctx->stream));
            pl.cfg.full_band = 1;
            const uint32_t cnt = (uint32_t)redo.size();
            const uint32_t *ids = d_ids.as<uint32_t>();
            (void)hipEventRecord(ctx->ev[4], ctx->stream);
            PBA_DISPATCH_NB(pl.nb2, K_SS);
            (void)hipEventRecord(ctx->ev[5], ctx->stream);
            ctx->prof.nb_redo = (uint32_t)pl.nb2; ctx->prof.n_redo = cnt;
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(h_rows.data(), d_rows.p, sizeof(pba_ss_row) * n, hipMemcpyDeviceToHost, ctx->stream));
        if (touch && !redo.empty()) HIPCHK(hipMemcpyAsync(h_redo.data(), d_redo.p, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (touch) for (uint32_t r = 0; r < n; ++r) touch[r] = (uint8_t)((h_redo[r] >> 1) & 3);
        if (!subset) memcpy(rows, h_rows.data(), sizeof(pba_ss_row) * n);
        else for (uint32_t k = 0; k < n_subset; ++k) rows[subset[k]] = h_rows[subset[k]];
        prof_finish(ctx);
    }
#undef K_SS
    return PBA_OK;
}

int pba_spaced_round(pba_ctx *ctx, const pba_index *ix, const pba_seqs *ref, uint32_t ref_seq, const pba_seqs *reads,
                     double R, int max_trial, int overlap_min, int buggy_seed_at, int kernel, pba_ss_row *rows) {
    return spaced_round_subset(ctx, ix, ref, ref_seq, reads, R, max_trial, overlap_min, buggy_seed_at, kernel, nullptr, 0, rows);
}

// spaced_seed.cpp:409-452 for a locked reference (-l): rounds over the reads not found yet, the seed of a round drawn
// like the reference draws it (a fresh draw after a round that found something, else the seeds in file order), stop
// when every seed has failed in a row or after max_round.  picks[] stands in for the values rand() returns.
int pba_spaced_multi(pba_ctx *ctx, const pba_seqs *ref, uint32_t ref_seq, const pba_seqs *reads, double R, int max_trial,
                     int overlap_min, int buggy_seed_at, int kernel, const uint32_t *masks, int n_masks,
                     const uint32_t *picks, int n_picks, int max_round, pba_ss_row *rows, int32_t *found_round,
                     pba_ss_round_log *log, int log_cap, int *n_rounds) {
    if (!ctx || !ref || !reads || !masks || n_masks < 1 || !picks || n_picks < 1 || max_round < 0 || !rows || !found_round ||
        !n_rounds || log_cap < 0 || (!log && log_cap) || ref_seq >= ref->n)
        return PBA_E_INVALID;
    const uint32_t n = reads->n;
    std::vector<uint32_t> pool(n);
    for (uint32_t r = 0; r < n; ++r) { pool[r] = r; found_round[r] = 0; memset(&rows[r], 0, sizeof rows[r]); rows[r].read = (int32_t)r; rows[r].j = -1; }
    int nfailure = 0, draws = 0, done = 0;
    for (int nround = 1; nround <= max_round; ++nround) {
        const uint32_t mask = nfailure == 0 ? masks[picks[draws++ % n_picks] % (uint32_t)n_masks] : masks[nfailure - 1];   // :412
        pba_index *ix = nullptr;
        int st = pba_index_build(ctx, ref, ref_seq, mask, PBA_INDEX_HEAD_TAIL, &ix);        // get_seedmap, :415
        if (st != PBA_OK) return st;
        st = spaced_round_subset(ctx, ix, ref, ref_seq, reads, R, max_trial, overlap_min, buggy_seed_at, kernel, pool.data(),
                                 (uint32_t)pool.size(), rows);
        pba_index_destroy(ix);
        if (st != PBA_OK) return st;
        int nmatches = 0;
        std::vector<uint32_t> rest;
        rest.reserve(pool.size());
        for (uint32_t r : pool) {
            if (rows[r].found) { found_round[r] = nround; ++nmatches; }                     // erased from the pool, :443
            else rest.push_back(r);
        }
        if (done < log_cap) { log[done].round = nround; log[done].mask = mask; log[done].n_tried = (int32_t)pool.size(); log[done].n_found = nmatches; }
        ++done;
        pool.swap(rest);
        if (nmatches != 0) nfailure = 0;                                                    // :448-451
        else if (++nfailure == n_masks) break;
    }
    *n_rounds = done;
    return PBA_OK;
}

}  // extern "C"
