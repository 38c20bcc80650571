// pba_overlap.hip -- all-vs-all overlap: host side of csrc/overlap.h.
// One process per GPU, one pba_ctx per process, one HIP stream per ctx.  Everything here fails loudly
// (PBA_E_NODEVICE / PBA_E_HIP): there is no CPU path behind these entry points.
#include "pba_host.h"
#include "overlap.h"

// (the attribute belongs to the current device: remembered per ctx, so a second ctx on another GPU sets it there too)
static void tu_attrs(pba_ctx *ctx) {
    if (ctx->attr_done & 4u) return;
    ctx->attr_done |= 4u;
    PBA_BIG_LDS(k_ovl_walk<0>);
}

extern "C" {

// ---------------------------------------------------------------------------------------------
// host API: all-vs-all overlap
// ---------------------------------------------------------------------------------------------
#define PBA_OVL_WALK(NBV)                                                                                             \
    hipLaunchKernelGGL((k_ovl_walk<NBV>), dim3(persistent_grid(ctx, n_items, (NBV) ? 4 : 1, lds)),                        \
                       dim3(PBA_WAVE * ((NBV) ? 4 : 1)), lds * ((NBV) ? 4 : 1), ctx->stream, reads->dev(), t_lo, n_items,  \
                       items, d_woff, d_wcnt, d_cand.as<uint64_t>(), ocfg, full_band,                                    \
                       redo_in, d_redo.as<uint2>(),                                                                      \
                       (unsigned long long)redo_cap, d_cnt64.as<unsigned long long>() + 2, d_out.as<pba_overlap>(),     \
                       (unsigned long long)cap, d_cnt64.as<unsigned long long>(), d_cnt64.as<unsigned long long>() + 1,  \
                       ctx->d_queue)

// the probe table of a read set (overlap.h: ProbeTab), built once and scanned by every target range
struct pba_probe_table {
    int device;
    ProbeTab T;
    bool hashed;
    uint32_t t2;
    uint64_t n_entries;
    float build_ms;
    // what a call learned from its sample about this read set under (R, ring): whether the narrow window certifies its
    // overlaps.  The next target ranges against the same table skip the sample (three launches and their tails per call).
    mutable uint32_t slice_max;    // largest / average candidate slice of a target in the last counted range of >= 1 024 targets (0: none yet)
    mutable double slice_avg;
    mutable int wide_known;        // -1: not sampled yet, 0: start narrow, 1: start in the middle ring / at the reference band
    mutable double wide_R;
    mutable int wide_nb1;
};

int pba_overlap_probes(pba_ctx *ctx, const pba_seqs *reads, uint32_t q_lo, uint32_t q_hi, uint32_t mask, int max_trial,
                       void *d_entries, uint64_t cap, uint64_t *n_out) {
    if (!ctx || !reads || !d_entries || !n_out || q_lo > q_hi || q_hi > reads->n) return PBA_E_INVALID;
    if (max_trial < 1 || 2 * max_trial >= (1 << PBA_OVL_JD_BITS)) PBA_FAIL(PBA_E_INVALID, "max_trial must be in [1, 63]");
    if (reads->n >= (1u << 24)) PBA_FAIL(PBA_E_TOOLONG, "at most 2^24 reads");
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs(ctx);
    const uint32_t t2 = 2u * (uint32_t)max_trial;
    const uint64_t slots = (uint64_t)(q_hi - q_lo) * t2;
    DevBuf counter;
    HIPCHK(hipMalloc(&counter.p, 8));
    HIPCHK(hipMemsetAsync(counter.p, 0, 8, ctx->stream));
    if (slots)
        hipLaunchKernelGGL(k_probe_emit, dim3((uint32_t)((slots + 255) / 256)), dim3(256), 0, ctx->stream, reads->dev(), q_lo,
                           q_hi - q_lo, t2, mask, (uint64_t *)d_entries, (unsigned long long)cap,
                           counter.as<unsigned long long>());
    unsigned long long h_n = 0;
    HIPCHK(hipMemcpyAsync(&h_n, counter.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    if (h_n > cap) PBA_FAIL(PBA_E_INVALID, "pba_overlap_probes: entry buffer too small");
    *n_out = h_n;
    return PBA_OK;
}

int pba_overlap_all(pba_ctx *ctx, const pba_seqs *reads, uint32_t t_lo, uint32_t t_hi, uint32_t mask, double R,
                    int max_trial, int overlap_min, int kernel, pba_overlap *out, uint64_t cap, uint64_t *n_out,
                    pba_overlap_stats *stats) {
    if (!ctx || !reads || !n_out) return PBA_E_INVALID;
    if (max_trial < 1 || 2 * max_trial >= (1 << PBA_OVL_JD_BITS)) PBA_FAIL(PBA_E_INVALID, "max_trial must be in [1, 63]");
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs(ctx);
    // 1. probe table of every read, built here (single-GPU form)
    const uint64_t pcap = (uint64_t)reads->n * 2u * (uint32_t)max_trial;
    DevBuf d_pent;
    HIPCHK(hipMalloc(&d_pent.p, sizeof(uint64_t) * (pcap + 1)));
    uint64_t n_pent = 0;
    int rc = pba_overlap_probes(ctx, reads, 0, reads->n, mask, max_trial, d_pent.p, pcap, &n_pent);
    if (rc != PBA_OK) return rc;
    return pba_overlap_all_probes(ctx, reads, t_lo, t_hi, d_pent.p, n_pent, mask, R, max_trial, overlap_min, kernel, out, cap,
                                  n_out, stats);
}

int pba_overlap_all_probes(pba_ctx *ctx, const pba_seqs *reads, uint32_t t_lo, uint32_t t_hi, const void *d_probe_entries,
                           uint64_t n_probe_slots, uint32_t mask, double R, int max_trial, int overlap_min, int kernel,
                           pba_overlap *out, uint64_t cap, uint64_t *n_out, pba_overlap_stats *stats) {
    if (!ctx || !n_out) return PBA_E_INVALID;
    pba_probe_table *tab = nullptr;
    int rc = pba_probe_table_create(ctx, d_probe_entries, n_probe_slots, mask, max_trial, &tab);
    if (rc != PBA_OK) return rc;
    rc = pba_overlap_all_table(ctx, reads, t_lo, t_hi, tab, R, overlap_min, kernel, out, cap, n_out, stats);
    pba_probe_table_destroy(tab);
    return rc;
}

void pba_probe_table_destroy(pba_probe_table *t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->T.start) (void)hipFree(t->T.start);
    if (t->T.pid) (void)hipFree(t->T.pid);
    if (t->T.pkey) (void)hipFree(t->T.pkey);
    if (t->T.presence) (void)hipFree(t->T.presence);
    if (t->T.rec) (void)hipFree(t->T.rec);
    delete t;
}

uint64_t pba_probe_table_entries(const pba_probe_table *t) { return t ? t->n_entries : 0; }

int pba_probe_table_create(pba_ctx *ctx, const void *d_probe_entries, uint64_t n_probe_slots, uint32_t mask, int max_trial,
                           pba_probe_table **out) {
    if (!ctx || !out || (!d_probe_entries && n_probe_slots)) return PBA_E_INVALID;
    *out = nullptr;
    if (max_trial < 1 || 2 * max_trial >= (1 << PBA_OVL_JD_BITS)) PBA_FAIL(PBA_E_INVALID, "max_trial must be in [1, 63]");
    if (n_probe_slots >= PBA_OVL_MAX_PROBES) PBA_FAIL(PBA_E_TOOLONG, "probe table: 2^32 probe slots or more (reads x 2 x max_trial)");
    HIPCHK(hipSetDevice(ctx->device));
    pba_probe_table *t = new (std::nothrow) pba_probe_table();
    if (!t) PBA_FAIL(PBA_E_NOMEM, "pba_probe_table");
    memset(t, 0, sizeof *t);
    t->device = ctx->device; t->t2 = 2u * (uint32_t)max_trial; t->wide_known = -1;
    struct Guard { pba_probe_table *p; ~Guard() { pba_probe_table_destroy(p); } } guard{t};
    const int care = __builtin_popcount(mask);
    t->hashed = care > PBA_PT_MAX_BITS;
    ProbeTab &T = t->T;
    T.mask = mask; T.bits = t->hashed ? PBA_PT_MAX_BITS : care;
    if (!t->hashed) {
        uint32_t m = mask, mk = ~m << 1;                         // Hacker's Delight 7-4: the move masks of compress(x, m)
        for (int i = 0; i < 5; ++i) {
            uint32_t mp = mk ^ (mk << 1);
            mp ^= mp << 2; mp ^= mp << 4; mp ^= mp << 8; mp ^= mp << 16;
            const uint32_t mv = mp & m;
            T.mv[i] = mv;
            m = (m ^ mv) | (mv >> (1 << i));
            mk &= ~mp;
        }
    }
    const uint64_t B = 1ull << T.bits, pres_words = std::max<uint64_t>(1, B / 32);
    HIPCHK(hipMalloc((void **)&T.start, sizeof(uint32_t) * (B + 1)));
    HIPCHK(hipMalloc((void **)&T.presence, sizeof(uint32_t) * pres_words));
    HIPCHK(hipMemsetAsync(T.start, 0, sizeof(uint32_t) * (B + 1), ctx->stream));
    HIPCHK(hipMemsetAsync(T.presence, 0, sizeof(uint32_t) * pres_words, ctx->stream));
    (void)hipEventRecord(ctx->ev[0], ctx->stream);
    const uint64_t n = n_probe_slots;
    const uint32_t grid = (uint32_t)((n + 255) / 256);
    const uint64_t *ent = (const uint64_t *)d_probe_entries;
    if (grid) {
        if (t->hashed) hipLaunchKernelGGL(k_pt_count<true>, dim3(grid), dim3(256), 0, ctx->stream, ent, n, T);
        else hipLaunchKernelGGL(k_pt_count<false>, dim3(grid), dim3(256), 0, ctx->stream, ent, n, T);
    }
    // start[b + 1] = entries of bucket b  ->  inclusive scan  ->  start[b] = first entry of bucket b
    const uint32_t n_tiles = (uint32_t)((B + PBA_SCAN_TILE - 1) / PBA_SCAN_TILE);
    DevBuf d_tiles, d_cursor;
    HIPCHK(hipMalloc(&d_tiles.p, sizeof(uint32_t) * n_tiles));
    hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(256), 0, ctx->stream, T.start + 1, B, d_tiles.as<uint32_t>());
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, d_tiles.as<uint32_t>(), n_tiles);
    hipLaunchKernelGGL(k_scan_add, dim3(n_tiles), dim3(256), 0, ctx->stream, T.start + 1, B, d_tiles.as<uint32_t>());
    uint32_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, T.start + B, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    t->n_entries = total;
    HIPCHK(hipMalloc((void **)&T.pid, sizeof(uint32_t) * ((uint64_t)total + 1)));
    if (t->hashed) HIPCHK(hipMalloc((void **)&T.pkey, sizeof(uint32_t) * ((uint64_t)total + 1)));
    if (total) {
        HIPCHK(hipMalloc(&d_cursor.p, sizeof(uint32_t) * B));
        HIPCHK(hipMemcpyAsync(d_cursor.p, T.start, sizeof(uint32_t) * B, hipMemcpyDeviceToDevice, ctx->stream));
        if (t->hashed) hipLaunchKernelGGL(k_pt_fill<true>, dim3(grid), dim3(256), 0, ctx->stream, ent, n, T, d_cursor.as<uint32_t>(), t->t2);
        else hipLaunchKernelGGL(k_pt_fill<false>, dim3(grid), dim3(256), 0, ctx->stream, ent, n, T, d_cursor.as<uint32_t>(), t->t2);
    }
    HIPCHK(hipMalloc((void **)&T.rec, sizeof(uint2) * (B + 2)));
    HIPCHK(hipMemsetAsync(T.rec + B, 0, sizeof(uint2) * 2, ctx->stream));
    hipLaunchKernelGGL(k_pt_pack, dim3((uint32_t)((B + 1 + 255) / 256)), dim3(256), 0, ctx->stream, T, B + 1, total);
    (void)hipEventRecord(ctx->ev[1], ctx->stream);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    (void)hipEventElapsedTime(&t->build_ms, ctx->ev[0], ctx->ev[1]);
    guard.p = nullptr;
    *out = t;
    return PBA_OK;
}

int pba_overlap_all_table(pba_ctx *ctx, const pba_seqs *reads, uint32_t t_lo, uint32_t t_hi, const pba_probe_table *tab, double R,
                          int overlap_min, int kernel, pba_overlap *out, uint64_t cap, uint64_t *n_out, pba_overlap_stats *stats) {
    if (!ctx || !reads || !tab || !n_out || (!out && cap) || t_lo > t_hi || t_hi > reads->n) return PBA_E_INVALID;
    if (reads->n >= PBA_OVL_MAX_READS) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: at most 2^24 reads");
    if ((uint64_t)reads->n * tab->t2 >= PBA_OVL_MAX_PROBES) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: reads x 2 x max_trial must stay below 2^32");
    if (reads->max_len > (uint32_t)kMaxSeqLen) PBA_FAIL(PBA_E_TOOLONG, "read longer than the engine limit");
    if (reads->non_acgt) PBA_FAIL(PBA_E_ALPHABET, "pba_overlap_all: the read set holds bytes outside ACGT");
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs(ctx);
    *n_out = 0;
    pba_overlap_stats st;
    memset(&st, 0, sizeof st);
    st.n_probe_entries = tab->n_entries;
    st.table_ms = tab->build_ms;
    const uint32_t n = reads->n, nt = t_hi - t_lo, t2 = tab->t2;
    if (nt == 0 || n < 2) { if (stats) *stats = st; return PBA_OK; }
    Plan pl;
    int rc = make_plan(ctx, R, 0, 0, kernel, 1 + (int)(reads->max_len * R), &pl);
    if (rc != PBA_OK) return rc;
    const ProbeTab &T = tab->T;
    DevBuf d_cnt64;
    HIPCHK(hipMalloc(&d_cnt64.p, 32));

    // 1. count: the slice of the candidate array every target needs -- or, for a later range of a table whose slices are
    //    known to be big and even (capacity mode), no count pass: every target gets the same room, 1.25 x the largest slice
    //    seen, and the fill pass reports what it needed; a target that outgrows its room sends the range through the
    //    counted way after all.  Capacity mode leaves gaps in the candidate array, which only the pre-sort stage (2b) can
    //    read, so it implies that stage.
    BufRef d_slice, d_off, d_valid, d_cand, d_tmp, d_out, d_small;
    // (the big arrays of a call live in the ctx's pool: at a million reads a target range needs 11 GB of candidates twice,
    // and mapping those anew for each of the 40 ranges took longer than everything the kernels do)
    POOL(POOL_OVL_SMALL, sizeof(uint32_t) * 5 * ((size_t)nt + 1), d_small.p);
    d_slice.p = d_small.as<uint32_t>(); d_off.p = d_small.as<uint32_t>() + (nt + 1); d_valid.p = d_small.as<uint32_t>() + 2 * ((size_t)nt + 1);
    uint32_t *const d_end = d_small.as<uint32_t>() + 4 * ((size_t)nt + 1);       // where every target's slice ends
    std::vector<uint32_t> h_slice(nt + 1), h_off(nt + 1), h_valid(nt + 1), h_end(nt + 1);
    uint64_t total = 0, extent = 0, max_cand = PBA_OVL_MAX_CANDIDATES;
    if (const char *e = getenv("PBA_OVL_MAX_CANDIDATES")) max_cand = std::min<uint64_t>(max_cand, (uint64_t)atoll(e));   // test hook: the limit at test sizes
    uint64_t prekeep_min = 1ull << 29;                           // (2b)
    if (const char *e = getenv("PBA_OVL_PREKEEP_MIN")) prekeep_min = (uint64_t)std::max(0LL, atoll(e));   // test hook: small inputs through the stage (or none)
    uint32_t slice_cap = 0;
    if (pl.nb1 != 0 && tab->slice_max > 0) {
        int pct = 125;
        if (const char *e = getenv("PBA_OVL_CAPFILL_PCT")) pct = atoi(e);         // test hook: 0 = never, small = overflow and fall back
        const uint64_t c = (uint64_t)tab->slice_max * (uint64_t)std::max(0, pct) / 100 + (pct > 0 ? 64 : 0);
        if (pct > 0 && (uint64_t)(tab->slice_avg * nt) >= prekeep_min && c * nt < max_cand && c < 0xFFFFFFFFull) slice_cap = (uint32_t)c;
    }
    uint32_t biggest_small = 2;
    std::vector<uint32_t> big;                                   // targets whose slice outgrows one LDS sort
    bool cap_mode = false;
    (void)hipEventRecord(ctx->ev[2], ctx->stream);
    for (int attempt = 0; attempt < 2; ++attempt) {
        cap_mode = attempt == 0 && slice_cap != 0;
        if (attempt == 0 && !cap_mode) continue;
        total = 0;
        if (cap_mode) {
            for (uint32_t i = 0; i <= nt; ++i) h_off[i] = (uint32_t)((uint64_t)i * slice_cap);
            extent = (uint64_t)nt * slice_cap;
        } else {
            if (tab->hashed) hipLaunchKernelGGL(k_ovl_count<true>, dim3(nt), dim3(PBA_WAVE * PBA_OVL_WAVES), 0, ctx->stream, T, reads->dev(), t_lo, nt, d_slice.as<uint32_t>());
            else hipLaunchKernelGGL(k_ovl_count<false>, dim3(nt), dim3(PBA_WAVE * PBA_OVL_WAVES), 0, ctx->stream, T, reads->dev(), t_lo, nt, d_slice.as<uint32_t>());
            HIPCHK(hipMemcpyAsync(h_slice.data(), d_slice.p, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            HIPCHK(hipGetLastError());
            uint32_t mx = 0;
            for (uint32_t i = 0; i < nt; ++i) {
                h_off[i] = (uint32_t)total;
                total += h_slice[i];
                mx = std::max(mx, h_slice[i]);
                if (total >= max_cand) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: 2^32 candidates or more in one target range; use smaller ranges");
                if (h_slice[i] <= PBA_IX_LDS_SORT_CAP) biggest_small = std::max(biggest_small, h_slice[i]);
                else big.push_back(i);
            }
            h_off[nt] = (uint32_t)total;
            extent = total;
            if (nt >= 1024) { tab->slice_max = mx; tab->slice_avg = (double)total / nt; }   // what a later range goes by
        }
        HIPCHK(hipMemcpyAsync(d_off.p, h_off.data(), sizeof(uint32_t) * (nt + 1), hipMemcpyHostToDevice, ctx->stream));
        POOL(POOL_OVL_CAND, sizeof(uint64_t) * (extent + 1), d_cand.p);

        // 2. fill: the candidates (all-ones where a slot belongs to the target's own probe or to another key)
        if (extent) {
            uint32_t *const d_written = cap_mode ? d_slice.as<uint32_t>() : nullptr;
            // runs of >= 6 probes on average: emit cooperatively (overlap.h).  Measured, second passes: a million reads (3.8
            // per run) scan 0.88 s one run per lane / 0.91 s cooperatively; two million (7.6 per run) 3.41 / 3.12 s; four
            // million (15 per run) 16.3 / 11.4 s
            uint32_t coop_avg = 6;
            if (const char *e = getenv("PBA_OVL_COOP_AVG")) coop_avg = (uint32_t)std::max(0, atoi(e));   // test hook: 0 = always
            if (tab->hashed) hipLaunchKernelGGL(k_ovl_fill<true>, dim3(nt), dim3(PBA_WAVE * PBA_OVL_WAVES), 0, ctx->stream, T, reads->dev(), t_lo, nt, d_off.as<uint32_t>(), d_cand.as<uint64_t>(), d_valid.as<uint32_t>(), cap_mode ? slice_cap : 0u, d_written, coop_avg);
            else hipLaunchKernelGGL(k_ovl_fill<false>, dim3(nt), dim3(PBA_WAVE * PBA_OVL_WAVES), 0, ctx->stream, T, reads->dev(), t_lo, nt, d_off.as<uint32_t>(), d_cand.as<uint64_t>(), d_valid.as<uint32_t>(), cap_mode ? slice_cap : 0u, d_written, coop_avg);
            HIPCHK(hipMemcpyAsync(h_valid.data(), d_valid.p, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost, ctx->stream));
            if (cap_mode) HIPCHK(hipMemcpyAsync(h_slice.data(), d_slice.p, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost, ctx->stream));
        } else {
            HIPCHK(hipMemsetAsync(d_valid.p, 0, sizeof(uint32_t) * (nt + 1), ctx->stream));
        }
        if (cap_mode) {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            HIPCHK(hipGetLastError());
            bool over = false;
            for (uint32_t i = 0; i < nt; ++i) { over = over || h_slice[i] > slice_cap; total += h_slice[i]; }
            if (over) { st.cap_overflow = 1; continue; }         // a slice outgrew its room: once more, counted
        }
        break;
    }
    (void)hipEventRecord(ctx->ev[3], ctx->stream);

    // 2b. big calls (>= 2^29 candidates) on the bit-vector kernels: the first prefilter stage before the sort (overlap.h: k_ovl_pre / k_ovl_keep).
    //     What is kept goes packed into the second buffer; the sort below moves it back, sorted; everything after sees the
    //     packed list (h_koff / h_kept) where it saw the slices (h_off / h_valid).  The time is the sort's in the statistics.
    // (measured, calls of 50 000 targets: 23 M candidates 0.081 s with the stage / 0.065 s without -- the few thousand dense
    // items that are left balance badly over 8 192 wavefronts --, 143 M 0.157 / 0.151, 573 M 0.333 / 0.330, 2.3 G 0.718 / 0.743,
    // 9.2 G 1.68 / 1.99, 57 G 5.8 / 7.9: prekeep_min = 2^29, above)
    const bool prekeep = total > 0 && (cap_mode || (pl.nb1 != 0 && total >= prekeep_min));
    st.cap_fill = cap_mode ? 1 : 0;
    for (uint32_t i = 0; i < nt; ++i) h_end[i] = h_off[i] + h_slice[i];
    HIPCHK(hipMemcpyAsync(d_end, h_end.data(), sizeof(uint32_t) * nt, hipMemcpyHostToDevice, ctx->stream));
    OvlCfg ocfg;
    ocfg.R = R; ocfg.overlap_min = overlap_min; ocfg.row_cap = pl.cfg.row_cap; ocfg.t2 = t2; ocfg.chunk = prekeep ? 1u : 0u;
    std::vector<uint32_t> h_kept, h_koff;                        // per target: candidates kept, and where they start in the packed list
    uint32_t *d_koff = nullptr;
    uint64_t n_dropped = 0;
    BufRef d_items;
    if (prekeep) {
        HIPCHK(hipStreamSynchronize(ctx->stream));               // h_valid
        std::vector<uint32_t> h_ipre1(nt + 1);
        uint64_t n_it = 0;
        for (uint32_t i = 0; i < nt; ++i) { h_ipre1[i] = (uint32_t)n_it; n_it += (h_slice[i] + PBA_WAVE - 1) / PBA_WAVE; }
        if (n_it >= 0xFFFFFFFFull) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: too many work items per call, use a smaller target range");
        h_ipre1[nt] = (uint32_t)n_it;
        uint32_t *d_ipre1 = d_small.as<uint32_t>() + 3 * ((size_t)nt + 1);
        HIPCHK(hipMemcpyAsync(d_ipre1, h_ipre1.data(), sizeof(uint32_t) * (nt + 1), hipMemcpyHostToDevice, ctx->stream));
        POOL(POOL_OVL_ITEMS, sizeof(uint2) * (n_it + 1), d_items.p);
        hipLaunchKernelGGL(k_ovl_items, dim3((uint32_t)((n_it + 255) / 256)), dim3(256), 0, ctx->stream, d_ipre1, d_off.as<uint32_t>(), nt,
                           (uint32_t)n_it, d_items.as<uint2>());
        // per target: the Bloom words, its blanked count; per item: the kept counts (+1: exclusive prefix sums); per target + 1:
        // the offsets into the packed list
        BufRef d_bloom;
        int bloom_bits = PBA_OVL_BLOOM_MIN_BITS;                 // ~2 x the average candidates of a target, a power of two
        while (bloom_bits < PBA_OVL_BLOOM_MAX_BITS && (1ull << bloom_bits) < 2 * (total / nt)) ++bloom_bits;
        const size_t bloom_words = (size_t)nt << (bloom_bits - 5);
        const uint32_t n_tiles = (uint32_t)((n_it + PBA_SCAN_TILE - 1) / PBA_SCAN_TILE);
        POOL(POOL_OVL_BLOOM, sizeof(uint32_t) * (bloom_words + 2 * ((size_t)nt + 1) + n_it + 1 + n_tiles) + sizeof(uint16_t) * (extent + 2), d_bloom.p);
        HIPCHK(hipMemsetAsync(d_bloom.p, 0, sizeof(uint32_t) * (bloom_words + 2 * ((size_t)nt + 1) + 1), ctx->stream));
        uint32_t *const d_blanked = d_bloom.as<uint32_t>() + bloom_words;
        d_koff = d_blanked + (nt + 1);
        uint32_t *const d_before = d_koff + (nt + 1);            // [n_it + 1], [0] = 0 (the memset above reaches it)
        uint32_t *const d_tiles = d_before + n_it + 1;
        uint16_t *const d_slot = (uint16_t *)(d_tiles + n_tiles);    // [total]: the Bloom slot of every candidate
        POOL(POOL_OVL_TMP, sizeof(uint64_t) * (total + 1), d_tmp.p);
        BufRef d_ends;                                           // the ends of every read side by side (overlap.h: OvlEnd)
        POOL(POOL_OVL_ENDS, sizeof(OvlEnd) * 2 * (size_t)n, d_ends.p);
        hipLaunchKernelGGL(k_ovl_ends, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, reads->dev(), n, d_ends.as<OvlEnd>());
        const uint32_t g = (uint32_t)((n_it + 3) / 4);
        hipLaunchKernelGGL(k_ovl_pre, dim3(g), dim3(256), 0, ctx->stream, reads->dev(), t_lo, (uint32_t)n_it, d_items.as<uint2>(),
                           d_end, d_cand.as<uint64_t>(), ocfg, PreThresholds::on_host(R), d_ends.as<OvlEnd>(), d_bloom.as<uint32_t>(), bloom_bits, d_blanked, d_slot);
        hipLaunchKernelGGL(k_ovl_keep_count, dim3(g), dim3(256), 0, ctx->stream, (uint32_t)n_it, d_items.as<uint2>(), d_end,
                           d_slot, d_bloom.as<uint32_t>(), bloom_bits, d_before + 1);
        hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(256), 0, ctx->stream, d_before + 1, n_it, d_tiles);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, d_tiles, n_tiles);
        hipLaunchKernelGGL(k_scan_add, dim3(n_tiles), dim3(256), 0, ctx->stream, d_before + 1, n_it, d_tiles);
        hipLaunchKernelGGL(k_ovl_keep_write, dim3(g), dim3(256), 0, ctx->stream, (uint32_t)n_it, d_items.as<uint2>(), d_end,
                           d_cand.as<uint64_t>(), d_slot, d_bloom.as<uint32_t>(), bloom_bits, d_before, d_tmp.as<uint64_t>());
        hipLaunchKernelGGL(k_ovl_keep_offsets, dim3((nt + 256) / 256), dim3(256), 0, ctx->stream, d_ipre1, d_before, nt, d_koff);
        h_koff.resize(nt + 1);
        std::vector<uint32_t> h_blanked(nt + 1);
        HIPCHK(hipMemcpyAsync(h_koff.data(), d_koff, sizeof(uint32_t) * (nt + 1), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(h_blanked.data(), d_blanked, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipGetLastError());
        h_kept.resize(nt + 1);
        for (uint32_t i = 0; i < nt; ++i) {
            h_kept[i] = h_koff[i + 1] - h_koff[i];
            n_dropped += (uint64_t)h_valid[i] - h_blanked[i] - h_kept[i];
        }
        st.n_prefiltered = n_dropped;
    }

    // 3. sort every target's slice = the reference's try order inside every (target, query): in LDS, in place; the big
    //    ones piece by piece through a second buffer
    if (total) {
        // k_seg_sort (seed_index.h): buckets by query range, every bucket sorted in wavefront registers.  What it reports
        // back (a slice with one bucket beyond 256 entries: one query with hundreds of candidates on the target) goes
        // through the global bitonic pass.
        const uint32_t ov_cap = 4096;
        DevBuf d_ov;
        HIPCHK(hipMalloc(&d_ov.p, sizeof(uint32_t) * (1 + ov_cap) * 2));
        uint32_t *const ov_small = d_ov.as<uint32_t>(), *const ov_piece = ov_small + 1 + ov_cap;
        HIPCHK(hipMemsetAsync(d_ov.p, 0, sizeof(uint32_t) * (1 + ov_cap) * 2, ctx->stream));
        std::vector<SegRef> h_pieces;
        if (prekeep) {
            // the packed list, target after target, from the second buffer into the candidate array, sorted; a target that kept
            // more than one sort holds (tandem repeats) is copied by the kernel and listed for the global pass
            uint32_t biggest = 2;
            for (uint32_t i = 0; i < nt; ++i) biggest = std::max(biggest, h_kept[i]);
            launch_seg_sort(ctx, d_tmp.as<uint64_t>(), d_cand.as<uint64_t>(), d_koff, nullptr, nt,
                            std::min<uint32_t>(biggest, PBA_IX_LDS_SORT_CAP), seg_bkt_range(), ov_small, ov_cap);
        } else
        launch_seg_sort(ctx, d_cand.as<uint64_t>(), d_cand.as<uint64_t>(), d_off.as<uint32_t>(), nullptr, nt,
                        big.empty() ? biggest_small : 0xFFFFFFFFu, seg_bkt_range(), ov_small, ov_cap);
        if (!prekeep && !big.empty()) {
            st.n_big_targets = (uint32_t)big.size();
            DevBuf d_big, d_pieces, d_pc;
            const uint32_t sub_mul = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, ((uint64_t)PBA_OVL_SUB << 32) / n);   // fine bucket = umulhi(q, sub_mul)
            POOL(POOL_OVL_TMP, sizeof(uint64_t) * (total + 1), d_tmp.p);
            HIPCHK(hipMalloc(&d_big.p, sizeof(uint32_t) * big.size()));
            HIPCHK(hipMalloc(&d_pieces.p, sizeof(OvlPiece) * big.size() * PBA_OVL_SUB));
            HIPCHK(hipMalloc(&d_pc.p, 8));
            HIPCHK(hipMemsetAsync(d_pc.p, 0, 8, ctx->stream));
            HIPCHK(hipMemcpyAsync(d_big.p, big.data(), sizeof(uint32_t) * big.size(), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_ovl_split, dim3((uint32_t)big.size()), dim3(1024), 0, ctx->stream, d_big.as<uint32_t>(), d_off.as<uint32_t>(),
                               d_cand.as<uint64_t>(), d_tmp.as<uint64_t>(), sub_mul, d_pieces.as<OvlPiece>(), d_pc.as<uint32_t>(),
                               d_pc.as<uint32_t>() + 1);
            uint32_t h_pc[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(h_pc, d_pc.p, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            HIPCHK(hipGetLastError());
            launch_seg_sort(ctx, d_tmp.as<uint64_t>(), d_cand.as<uint64_t>(), nullptr, d_pieces.as<SegRef>(), h_pc[0], 0xFFFFFFFFu,
                            seg_bkt_range(), ov_piece, ov_cap);
            h_pieces.resize(h_pc[0]);
            HIPCHK(hipMemcpyAsync(h_pieces.data(), d_pieces.p, sizeof(SegRef) * h_pc[0], hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            HIPCHK(hipGetLastError());
        }
        std::vector<uint32_t> h_ov((1 + ov_cap) * 2, 0);
        HIPCHK(hipMemcpyAsync(h_ov.data(), d_ov.p, sizeof(uint32_t) * (1 + ov_cap) * 2, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipGetLastError());
        // (a full list means "check everything")
        auto listed = [&](const uint32_t *ov, uint64_t n_seg) {
            std::vector<uint32_t> v;
            if (ov[0] > ov_cap) { v.resize(n_seg); for (uint64_t i = 0; i < n_seg; ++i) v[i] = (uint32_t)i; }
            else v.assign(ov + 1, ov + 1 + ov[0]);
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
            return v;
        };
        for (uint32_t i : listed(h_ov.data(), nt))                    // small slices with an overfull bucket (the big ones are cut into pieces)
            if (prekeep ? h_kept[i] > 1 : (h_slice[i] > 1 && h_slice[i] <= PBA_IX_LDS_SORT_CAP)) {
                rc = prekeep ? sort_partition_global(ctx, d_cand.as<uint64_t>() + h_koff[i], h_kept[i])
                             : sort_partition_global(ctx, d_cand.as<uint64_t>() + h_off[i], h_slice[i]);
                if (rc != PBA_OK) return rc;
            }
        for (uint32_t i : listed(h_ov.data() + 1 + ov_cap, h_pieces.size()))   // pieces beyond one sort, or with an overfull bucket
            if (i < h_pieces.size() && h_pieces[i].n > 1) {
                rc = sort_partition_global(ctx, d_cand.as<uint64_t>() + h_pieces[i].off, h_pieces[i].n);
                if (rc != PBA_OK) return rc;
            }
    }
    (void)hipEventRecord(ctx->ev[4], ctx->stream);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    uint64_t n_valid = 0;
    for (uint32_t i = 0; i < nt; ++i) n_valid += h_valid[i];
    st.n_candidates = n_valid;

    // 4. walk: persistent wavefronts, one target at a time, narrow window; then the parked (target, query) runs
    //    at the reference band
    POOL(POOL_OVL_OUT, sizeof(pba_overlap) * (cap + 1), d_out.p);
    HIPCHK(hipMemsetAsync(d_cnt64.p, 0, 32, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_queue, 0, 4, ctx->stream));
    const size_t lds = pl.lds;
    BufRef d_redo;
    const std::vector<uint32_t> &h_wcnt = prekeep ? h_kept : h_valid;      // what the walk sees of every target
    // (the walk takes a count per target: with the packed list, target t's is the distance to the next offset)
    BufRef d_kcnt;
    if (prekeep) {
        d_kcnt.p = d_small.as<uint32_t>() + 2 * ((size_t)nt + 1);            // d_valid's place: its host copy is what is used from here on
        HIPCHK(hipMemcpyAsync(d_kcnt.p, h_kept.data(), sizeof(uint32_t) * nt, hipMemcpyHostToDevice, ctx->stream));
    }
    const uint32_t *const d_wcnt = prekeep ? d_kcnt.as<uint32_t>() : d_valid.as<uint32_t>();
    const uint32_t *const d_woff = prekeep ? d_koff : d_off.as<uint32_t>();
    uint64_t n_walk = 0;
    for (uint32_t i = 0; i < nt; ++i) n_walk += h_wcnt[i];
    // every (target, query) run can park at most once per stage, and there are no more runs than candidates
    const uint64_t redo_cap = std::max<uint64_t>(1024, n_walk);
    POOL(POOL_OVL_REDO, sizeof(uint2) * redo_cap, d_redo.p);
    // work items: (target, first candidate of a group of 64), expanded on the device from the per-target item counts
    // (a million reads make 22 M items per call: building and copying them from the host took longer than a scan pass)
    std::vector<uint32_t> h_ipre(nt + 1);
    uint64_t n_items64 = 0;
    for (uint32_t i = 0; i < nt; ++i) { h_ipre[i] = (uint32_t)n_items64; n_items64 += (h_wcnt[i] + PBA_WAVE - 1) / PBA_WAVE; }
    if (n_items64 >= 0xFFFFFFFFull) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: too many work items per call, use a smaller target range");
    h_ipre[nt] = (uint32_t)n_items64;
    BufRef d_ipre;
    d_ipre.p = d_small.as<uint32_t>() + 3 * ((size_t)nt + 1);
    HIPCHK(hipMemcpyAsync(d_ipre.p, h_ipre.data(), sizeof(uint32_t) * (nt + 1), hipMemcpyHostToDevice, ctx->stream));
    POOL(POOL_OVL_ITEMS, sizeof(uint2) * (n_items64 + 1), d_items.p);
    if (n_items64)
        hipLaunchKernelGGL(k_ovl_items, dim3((uint32_t)((n_items64 + 255) / 256)), dim3(256), 0, ctx->stream, d_ipre.as<uint32_t>(),
                           d_woff, nt, (uint32_t)n_items64, d_items.as<uint2>());
    HIPCHK(hipStreamSynchronize(ctx->stream));                   // h_ipre must outlive its copy
    // one launch of the walk: items [lo, hi) of the group list (redo_in == nullptr) or n_redo parked runs
    auto walk = [&](int nb, const uint2 *items, uint32_t n_items, int full_band, const uint2 *redo_in) -> int {
        HIPCHK(hipMemsetAsync(ctx->d_queue, 0, 4, ctx->stream));
        HIPCHK(hipMemsetAsync(d_cnt64.as<unsigned long long>() + 2, 0, 8, ctx->stream));
        PBA_DISPATCH_NB(nb, PBA_OVL_WALK);
        HIPCHK(hipGetLastError());
        return PBA_OK;
    };
    // narrow window for items [lo, hi), then the runs it parked at the reference band; returns the number parked
    uint64_t parked_total = 0;
    // items [lo, hi) in the ring nb_first with its first-pass window, then what that parked in the widest ring below the
    // reference band's (its window takes all the room the ring has, bv_pass1_w: at 15 kb NB = 3 holds 4 072 of max_dst
    // 4 501), then what is still parked at the reference band; returns the number parked by the first stage
    int nb_mid = 0;
    for (int nb : {1, 2, 3, 4, 6})
        if (nb > pl.nb1 && nb < pl.nb2) nb_mid = nb;
    auto narrow_then_redo = [&](int nb_first, size_t lo, size_t hi, uint64_t *parked) -> int {
        *parked = 0;
        if (hi <= lo) return PBA_OK;
        int rc2 = walk(nb_first, d_items.as<uint2>() + lo, (uint32_t)(hi - lo), 0, nullptr);
        if (rc2 != PBA_OK) return rc2;
        for (int stage = 0; stage < 2; ++stage) {
            unsigned long long h_redo = 0;
            HIPCHK(hipMemcpyAsync(&h_redo, d_cnt64.as<unsigned long long>() + 2, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (h_redo > redo_cap) PBA_FAIL(PBA_E_NOMEM, "pba_overlap_all: more uncertified (target, query) runs than the redo list holds");
            if (stage == 0) *parked = h_redo;
            if (!h_redo) return PBA_OK;
            if (stage == 0 && nb_mid <= nb_first) continue;      // no ring between this one and the reference band's
            BufRef d_in;
            POOL(POOL_OVL_REDO_IN, sizeof(uint2) * h_redo, d_in.p);
            HIPCHK(hipMemcpyAsync(d_in.p, d_redo.p, sizeof(uint2) * h_redo, hipMemcpyDeviceToDevice, ctx->stream));
            rc2 = stage == 0 ? walk(nb_mid, nullptr, (uint32_t)h_redo, 0, d_in.as<uint2>())
                             : walk(pl.nb2, nullptr, (uint32_t)h_redo, 1, d_in.as<uint2>());
            if (rc2 != PBA_OK) return rc2;
            HIPCHK(hipStreamSynchronize(ctx->stream));
        }
        return PBA_OK;
    };
    // Whether the narrow window pays depends on how far the reads are from each other (two 15 % reads differ by ~27 %:
    // nothing certifies below the reference band), which only the data tells: a sample of the items goes through
    // narrow-then-redo, and if most of its successful runs had to be parked the rest starts wider: in the widest ring
    // below the reference band's (its first-pass window takes all the room that ring has, bv_pass1_w -- at 15 kb NB = 3
    // holds a window of 4 072, which certifies every overlap but the longest), or straight at the reference band.
    const size_t n_all = (size_t)n_items64;
    size_t sample_min = 4096;
    if (const char *e = getenv("PBA_OVL_SAMPLE_MIN")) sample_min = (size_t)std::max(1L, atol(e));   // test hook: small inputs through the sampled decision
    const bool decided = tab->wide_known >= 0 && tab->wide_R == R && tab->wide_nb1 == pl.nb1 && !getenv("PBA_OVL_SAMPLE_MIN");
    const size_t n_sample = pl.nb1 == 0 ? n_all : (decided ? 0 : std::min(n_all, std::max<size_t>(sample_min, n_all / 32)));
    uint64_t parked = 0;
    rc = narrow_then_redo(pl.nb1, 0, n_sample, &parked);
    if (rc != PBA_OK) return rc;
    parked_total += parked;
    if (n_sample < n_all) {
        unsigned long long h_ov = 0;
        if (!decided) {
            HIPCHK(hipMemcpyAsync(&h_ov, d_cnt64.p, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (n_sample >= 4096) {                              // a sample worth remembering
                tab->wide_known = 2 * parked > h_ov ? 1 : 0; tab->wide_R = R; tab->wide_nb1 = pl.nb1;
            }
        }
        if (decided ? tab->wide_known == 1 : 2 * parked > h_ov) {   // most overlaps of the sample needed more than the narrow window
            st.wide_first = 1;
            if (nb_mid) {
                rc = narrow_then_redo(nb_mid, n_sample, n_all, &parked);
                parked_total += parked;
            } else rc = walk(pl.nb2, d_items.as<uint2>() + n_sample, (uint32_t)(n_all - n_sample), 1, nullptr);
        } else {
            rc = narrow_then_redo(pl.nb1, n_sample, n_all, &parked);
            parked_total += parked;
        }
        if (rc != PBA_OK) return rc;
    }
    st.n_redo = parked_total;
    (void)hipEventRecord(ctx->ev[5], ctx->stream);
    HIPCHK(hipGetLastError());
    unsigned long long h_cnt2[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h_cnt2, d_cnt64.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint64_t got = std::min<uint64_t>(h_cnt2[0], cap);
    if (got) HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_overlap) * got, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::sort(out, out + got, [](const pba_overlap &x, const pba_overlap &y) {
        return x.target != y.target ? x.target < y.target : x.query < y.query;
    });
    *n_out = h_cnt2[0];
    st.n_overlaps = h_cnt2[0];
    st.n_pairs = h_cnt2[1] + n_dropped;
    (void)hipEventElapsedTime(&st.scan_ms, ctx->ev[2], ctx->ev[3]);
    (void)hipEventElapsedTime(&st.sort_ms, ctx->ev[3], ctx->ev[4]);
    (void)hipEventElapsedTime(&st.walk_ms, ctx->ev[4], ctx->ev[5]);
    if (stats) *stats = st;
    return PBA_OK;
}

}  // extern "C"
