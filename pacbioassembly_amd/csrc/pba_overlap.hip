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
                       (unsigned long long)dev_cap, d_cnt64.as<unsigned long long>(), d_cnt64.as<unsigned long long>() + 1, \
                       ctx->d_queue)

// the probe table of a read set (overlap.h: ProbeTab), built once and scanned by every target range
struct pba_probe_table {
    int device;
    mutable ProbeTab T;            // (prec is attached on first use)
    bool hashed;
    uint32_t t2;
    uint64_t n_entries;
    float build_ms;
    // the read set prec[] was filled from (rec_reads == nullptr: not yet -- on first use, overlap.h: k_pt_ctx); a table belongs to
    // one read set, and the arena's address and size are compared too, should a set have been replaced at the same address
    mutable const pba_seqs *rec_reads;
    mutable const uint8_t *rec_packed;
    mutable uint64_t rec_bytes;
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
    if (t->T.prec) (void)hipFree(t->T.prec);
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
    DevBuf d_cnt64;                // [0] overlaps, [1] pairs (the row-sweep form's walk), [2] parked runs; [4] seed matches, [5] candidates
    HIPCHK(hipMalloc(&d_cnt64.p, 64));   // past the gate (k_ovl_scan), [6] candidates behind a success (k_ovl_after)
    HIPCHK(hipMemsetAsync(d_cnt64.p, 0, 64, ctx->stream));
    const bool fused = pl.nb1 != 0;          // the bit-vector kernels: first 32 rows in the scan, survivors only in memory

    BufRef d_slice, d_off, d_valid, d_cand, d_tmp, d_out, d_small;
    // (the big arrays of a call live in the ctx's pool: mapping gigabytes anew for each target range of a table took longer
    // than everything the kernels do)
    POOL(POOL_OVL_SMALL, sizeof(uint32_t) * 5 * ((size_t)nt + 1), d_small.p);
    d_slice.p = d_small.as<uint32_t>(); d_off.p = d_small.as<uint32_t>() + (nt + 1); d_valid.p = d_small.as<uint32_t>() + 2 * ((size_t)nt + 1);
    std::vector<uint32_t> h_slice(nt + 1), h_off(nt + 1), h_valid(nt + 1);
    uint64_t total = 0, max_cand = PBA_OVL_MAX_CANDIDATES;
    if (const char *e = getenv("PBA_OVL_MAX_CANDIDATES")) max_cand = std::min<uint64_t>(max_cand, (uint64_t)atoll(e));   // test hook: the limit at test sizes
    uint32_t biggest_small = 2;
    std::vector<uint32_t> big;                                   // row-sweep form: targets whose slice outgrows one LDS sort
    OvlCfg ocfg;
    ocfg.R = R; ocfg.overlap_min = overlap_min; ocfg.row_cap = pl.cfg.row_cap; ocfg.t2 = t2; ocfg.chunk = fused ? 1u : 0u; ocfg.fused = fused ? 1 : 0;
    uint64_t n_cand = 0, n_ok = 0;
    (void)hipEventRecord(ctx->ev[2], ctx->stream);
    if (fused) {
        // 1. the records of the probe table from this read set, once per table
        if (tab->rec_reads != reads || tab->rec_packed != reads->d_packed || tab->rec_bytes != reads->packed_bytes) {
            if (!tab->T.prec) HIPCHK(hipMalloc((void **)&tab->T.prec, sizeof(uint4) * ((uint64_t)tab->n_entries + 1)));
            if (tab->n_entries)
                hipLaunchKernelGGL(k_pt_ctx, dim3((uint32_t)((tab->n_entries + 255) / 256)), dim3(256), 0, ctx->stream, T, reads->dev(),
                                   (uint32_t)tab->n_entries);
            HIPCHK(hipGetLastError());
            tab->rec_reads = reads; tab->rec_packed = reads->d_packed; tab->rec_bytes = reads->packed_bytes;
        }
        // 2. the scan.  How much room a target's survivors need is not known before its candidates have been through their
        //    32 rows: the first range of a table runs the scan once without writing (needed[] only) and then with exact
        //    slices; later ranges give every target the same room -- 1.25 x the largest need seen, + 64 -- and fall back to
        //    exact slices (needed[] of the clipped run) when a target outgrows it.
        uint32_t room = 0;
        int pct = 125;
        if (const char *e = getenv("PBA_OVL_CAPFILL_PCT")) pct = atoi(e);         // test hook: 0 = never equal room (a full census per range), small = overflow and fall back
        if (tab->slice_max > 0) {
            const uint64_t c = (uint64_t)tab->slice_max * (uint64_t)std::max(0, pct) / 100 + 64;
            if (pct > 0 && c * nt < max_cand) room = (uint32_t)c;
        }
        if (const char *e = getenv("PBA_OVL_ROOM"))              // test hook: this much room for every target of every range (and so the overflow path at will)
            if (atoi(e) > 0) room = (uint32_t)atoi(e);
        const PreChecks pre_t = PreChecks::on_host(R);
        const size_t plane_lds = sizeof(uint32_t) * 2 * ((size_t)reads->max_len / 32 + 2);     // the target's bit planes (k_ovl_scan)
        auto scan = [&](bool write, uint32_t cap_slots, uint32_t grid, uint32_t stride) -> int {
            HIPCHK(hipMemsetAsync(d_cnt64.as<unsigned long long>() + 4, 0, 16, ctx->stream));
            const uint32_t *so = write ? d_off.as<uint32_t>() : nullptr;
            if (tab->hashed) hipLaunchKernelGGL(k_ovl_scan<true>, dim3(grid), dim3(PBA_WAVE * PBA_OVL_WAVES), plane_lds, ctx->stream, T, reads->dev(), t_lo, stride, so, d_cand.as<uint64_t>(), cap_slots, d_slice.as<uint32_t>(), ocfg, pre_t, d_cnt64.as<unsigned long long>() + 4);
            else hipLaunchKernelGGL(k_ovl_scan<false>, dim3(grid), dim3(PBA_WAVE * PBA_OVL_WAVES), plane_lds, ctx->stream, T, reads->dev(), t_lo, stride, so, d_cand.as<uint64_t>(), cap_slots, d_slice.as<uint32_t>(), ocfg, pre_t, d_cnt64.as<unsigned long long>() + 4);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(h_slice.data(), d_slice.p, sizeof(uint32_t) * grid, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            return PBA_OK;
        };
        bool have_exact = false;
        if (room == 0) {
            // census: all targets of a small range, every k-th of a big one (a sixteenth of the work; what it misses the
            // overflow path catches)
            uint32_t n_s = pct > 0 ? std::min<uint32_t>(nt, std::max<uint32_t>(64u, nt / 16u)) : nt;
            const uint32_t stride = nt / n_s;
            if (stride == 1) n_s = nt;                           // (no sample worth the name: every target)
            rc = scan(false, 0, n_s, stride);
            if (rc != PBA_OK) return rc;
            if (stride == 1) have_exact = true;
            else {
                uint32_t mx = 0;
                for (uint32_t i = 0; i < n_s; ++i) mx = std::max(mx, h_slice[i]);
                room = mx + mx / 2 + 64;
                if ((uint64_t)room * nt >= max_cand) { rc = scan(false, 0, nt, 1); if (rc != PBA_OK) return rc; have_exact = true; }   // (too much room to hand out blindly)
            }
        }
        for (int attempt = 0; attempt < 2; ++attempt) {
            uint64_t extent = 0;
            if (have_exact) {
                uint32_t mx = 0;
                for (uint32_t i = 0; i < nt; ++i) {
                    h_off[i] = (uint32_t)extent; extent += h_slice[i]; mx = std::max(mx, h_slice[i]);
                    if (extent >= max_cand) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: 2^32 candidates or more in one target range; use smaller ranges");
                }
                if (nt >= 1024 || tab->slice_max == 0) { tab->slice_max = std::max(mx, 1u); tab->slice_avg = (double)extent / nt; }   // what a later range goes by
            } else {
                for (uint32_t i = 0; i < nt; ++i) h_off[i] = (uint32_t)((uint64_t)i * room);
                extent = (uint64_t)nt * room;
            }
            h_off[nt] = (uint32_t)extent;
            HIPCHK(hipMemcpyAsync(d_off.p, h_off.data(), sizeof(uint32_t) * (nt + 1), hipMemcpyHostToDevice, ctx->stream));
            POOL(POOL_OVL_CAND, sizeof(uint64_t) * (extent + 1), d_cand.p);
            rc = scan(true, have_exact ? 0xFFFFFFFFu : room, nt, 1);
            if (rc != PBA_OK) return rc;
            if (have_exact) break;
            bool over = false;
            uint32_t mx = 0;
            for (uint32_t i = 0; i < nt; ++i) { over = over || h_slice[i] > room; mx = std::max(mx, h_slice[i]); }
            if (!over) {
                st.cap_fill = 1;
                if (nt >= 1024 || tab->slice_max == 0) tab->slice_max = std::max(mx, 1u);     // what a later range goes by
                break;
            }
            st.cap_overflow = 1;                                 // a target outgrew its room: once more, with what each one needed
            have_exact = true;
        }
        unsigned long long h_tot[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(h_tot, d_cnt64.as<unsigned long long>() + 4, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        n_cand = h_tot[0]; n_ok = h_tot[1];
        for (uint32_t i = 0; i < nt; ++i) { h_valid[i] = h_slice[i]; total += h_slice[i]; }
        HIPCHK(hipMemcpyAsync(d_valid.p, h_valid.data(), sizeof(uint32_t) * nt, hipMemcpyHostToDevice, ctx->stream));
        st.n_prefiltered = n_ok - total;
    } else {
        // 1. count: the slice of the candidate array every target needs
        if (tab->hashed) hipLaunchKernelGGL(k_ovl_count<true>, dim3(nt), dim3(PBA_WAVE * PBA_OVL_WAVES), 0, ctx->stream, T, reads->dev(), t_lo, nt, d_slice.as<uint32_t>());
        else hipLaunchKernelGGL(k_ovl_count<false>, dim3(nt), dim3(PBA_WAVE * PBA_OVL_WAVES), 0, ctx->stream, T, reads->dev(), t_lo, nt, d_slice.as<uint32_t>());
        HIPCHK(hipMemcpyAsync(h_slice.data(), d_slice.p, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipGetLastError());
        for (uint32_t i = 0; i < nt; ++i) {
            h_off[i] = (uint32_t)total;
            total += h_slice[i];
            if (total >= max_cand) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: 2^32 candidates or more in one target range; use smaller ranges");
            if (h_slice[i] <= PBA_IX_LDS_SORT_CAP) biggest_small = std::max(biggest_small, h_slice[i]);
            else big.push_back(i);
        }
        h_off[nt] = (uint32_t)total;
        HIPCHK(hipMemcpyAsync(d_off.p, h_off.data(), sizeof(uint32_t) * (nt + 1), hipMemcpyHostToDevice, ctx->stream));
        POOL(POOL_OVL_CAND, sizeof(uint64_t) * (total + 1), d_cand.p);
        // 2. fill: the candidates (all-ones where a slot belongs to the target's own probe or to another key)
        if (total) {
            if (tab->hashed) hipLaunchKernelGGL(k_ovl_fill<true>, dim3(nt), dim3(PBA_WAVE * PBA_OVL_WAVES), 0, ctx->stream, T, reads->dev(), t_lo, nt, d_off.as<uint32_t>(), d_cand.as<uint64_t>(), d_valid.as<uint32_t>());
            else hipLaunchKernelGGL(k_ovl_fill<false>, dim3(nt), dim3(PBA_WAVE * PBA_OVL_WAVES), 0, ctx->stream, T, reads->dev(), t_lo, nt, d_off.as<uint32_t>(), d_cand.as<uint64_t>(), d_valid.as<uint32_t>());
            HIPCHK(hipMemcpyAsync(h_valid.data(), d_valid.p, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost, ctx->stream));
        } else {
            HIPCHK(hipMemsetAsync(d_valid.p, 0, sizeof(uint32_t) * (nt + 1), ctx->stream));
        }
    }
    (void)hipEventRecord(ctx->ev[3], ctx->stream);

    // 3. sort every target's slice = the reference's try order inside every (target, query): in LDS, in place; the big
    //    ones (row-sweep form only: a million reads leave ~1 000 survivors per target, not 57 000 candidates) piece by piece
    //    through a second buffer
    if (total) {
        // k_seg_sort (seed_index.h): buckets by query range, every bucket sorted in wavefront registers.  What it reports
        // back (a slice with one bucket beyond 256 entries: one query with hundreds of candidates on the target; a slice
        // beyond one workgroup) goes through the global bitonic pass.
        const uint32_t ov_cap = 4096;
        DevBuf d_ov;
        HIPCHK(hipMalloc(&d_ov.p, sizeof(uint32_t) * (1 + ov_cap) * 2));
        uint32_t *const ov_small = d_ov.as<uint32_t>(), *const ov_piece = ov_small + 1 + ov_cap;
        HIPCHK(hipMemsetAsync(d_ov.p, 0, sizeof(uint32_t) * (1 + ov_cap) * 2, ctx->stream));
        std::vector<SegRef> h_pieces;
        if (fused) {
            // (equal-room slices have gaps: the segments are given one by one)
            std::vector<SegRef> h_seg(nt);
            uint32_t biggest = 2;
            for (uint32_t i = 0; i < nt; ++i) { h_seg[i] = SegRef{h_off[i], h_valid[i]}; biggest = std::max(biggest, h_valid[i]); }
            BufRef d_seg;
            POOL(POOL_OVL_TMP, sizeof(SegRef) * ((size_t)nt + 1), d_seg.p);
            HIPCHK(hipMemcpyAsync(d_seg.p, h_seg.data(), sizeof(SegRef) * nt, hipMemcpyHostToDevice, ctx->stream));
            launch_seg_sort(ctx, d_cand.as<uint64_t>(), d_cand.as<uint64_t>(), nullptr, d_seg.as<SegRef>(), nt,
                            std::min<uint32_t>(biggest, PBA_IX_LDS_SORT_CAP), seg_bkt_range(), ov_small, ov_cap);
            HIPCHK(hipStreamSynchronize(ctx->stream));           // h_seg
        } else
        launch_seg_sort(ctx, d_cand.as<uint64_t>(), d_cand.as<uint64_t>(), d_off.as<uint32_t>(), nullptr, nt,
                        big.empty() ? biggest_small : 0xFFFFFFFFu, seg_bkt_range(), ov_small, ov_cap);
        if (!fused && !big.empty()) {
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
        for (uint32_t i : listed(h_ov.data(), nt))                    // slices with an overfull bucket, or beyond one workgroup (the row-sweep form cuts those into pieces)
            if (fused ? h_valid[i] > 1 : (h_slice[i] > 1 && h_slice[i] <= PBA_IX_LDS_SORT_CAP)) {
                rc = sort_partition_global(ctx, d_cand.as<uint64_t>() + h_off[i], fused ? h_valid[i] : h_slice[i]);
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
    st.n_candidates = fused ? n_cand : n_valid;
    st.n_listed = n_valid;

    // 4. walk: persistent wavefronts, one target at a time, narrow window; then the parked (target, query) runs
    //    at the reference band
    // (the device list holds every success -- there are no more of them than listed candidates -- whatever the caller's cap:
    // k_ovl_after goes through all of them)
    const uint64_t dev_cap = fused ? std::max<uint64_t>(cap, n_valid) : cap;
    POOL(POOL_OVL_OUT, sizeof(pba_overlap) * (dev_cap + 1), d_out.p);
    HIPCHK(hipMemsetAsync(d_cnt64.p, 0, 32, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_queue, 0, 4, ctx->stream));
    const size_t lds = pl.lds;
    BufRef d_redo;
    const std::vector<uint32_t> &h_wcnt = h_valid;               // what the walk sees of every target
    const uint32_t *const d_wcnt = d_valid.as<uint32_t>();
    const uint32_t *const d_woff = d_off.as<uint32_t>();
    BufRef d_items;
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
    if (const char *e = getenv("PBA_OVL_WIDE")) {                // tuning hook: 0 / 1 = start every range narrow / in the wider ring, no sample
        tab->wide_known = atoi(e) != 0 ? 1 : 0; tab->wide_R = R; tab->wide_nb1 = pl.nb1;
    }
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
    HIPCHK(hipGetLastError());
    unsigned long long h_cnt2[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h_cnt2, d_cnt64.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // 5. pairs.  Row-sweep form: the walk counted what it tried.  Fused form: the scan counted every candidate past the gate
    //    as a pair; what lies behind the first success of a run was never tried (overlap.h: k_ovl_after)
    unsigned long long h_after = 0;
    if (fused && h_cnt2[0]) {
        const uint32_t n_ov = (uint32_t)std::min<uint64_t>(h_cnt2[0], dev_cap);
        hipLaunchKernelGGL(k_ovl_after, dim3((n_ov + 3) / 4), dim3(PBA_WAVE * 4), 0, ctx->stream, reads->dev(), d_out.as<pba_overlap>(), n_ov,
                           T.mask, t2, overlap_min, d_cnt64.as<unsigned long long>() + 6);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&h_after, d_cnt64.as<unsigned long long>() + 6, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    (void)hipEventRecord(ctx->ev[5], ctx->stream);
    const uint64_t got = std::min<uint64_t>(h_cnt2[0], cap);
    if (got) HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_overlap) * got, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // rows by (target, query): the walk's wavefronts emit in the order they finish.  A counting pass over the range's targets,
    // then the few rows of a target by query (a comparison sort of the whole list was 12 ms per 200 000 rows on the host: a
    // tenth of a 200 k-read pass)
    if (got > 1) {
        std::vector<uint32_t> first(nt + 1, 0);
        bool in_range = true;
        for (uint64_t i = 0; i < got; ++i) {
            const uint32_t tl = (uint32_t)out[i].target - t_lo;
            if (tl >= nt) { in_range = false; break; }
            ++first[tl + 1];
        }
        if (in_range) {
            for (uint32_t t = 0; t < nt; ++t) first[t + 1] += first[t];
            std::vector<pba_overlap> tmp(out, out + got);
            std::vector<uint32_t> at(first.begin(), first.end() - 1);
            for (uint64_t i = 0; i < got; ++i) out[at[(uint32_t)tmp[i].target - t_lo]++] = tmp[i];
            for (uint32_t t = 0; t < nt; ++t)
                if (first[t + 1] - first[t] > 1)
                    std::sort(out + first[t], out + first[t + 1], [](const pba_overlap &x, const pba_overlap &y) { return x.query < y.query; });
        } else
            std::sort(out, out + got, [](const pba_overlap &x, const pba_overlap &y) {
                return x.target != y.target ? x.target < y.target : x.query < y.query;
            });
    }
    *n_out = h_cnt2[0];
    st.n_overlaps = h_cnt2[0];
    st.n_pairs = fused ? n_ok - h_after : h_cnt2[1];
    (void)hipEventElapsedTime(&st.scan_ms, ctx->ev[2], ctx->ev[3]);
    (void)hipEventElapsedTime(&st.sort_ms, ctx->ev[3], ctx->ev[4]);
    (void)hipEventElapsedTime(&st.walk_ms, ctx->ev[4], ctx->ev[5]);
    if (stats) *stats = st;
    return PBA_OK;
}

}  // extern "C"
