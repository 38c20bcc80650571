// pba_overlap.hip -- all-vs-all overlap: host side of csrc/overlap.h.
// One process per GPU, one pba_ctx per process, one HIP stream per ctx.  Everything here fails loudly
// (PBA_E_NODEVICE / PBA_E_HIP): there is no CPU path behind these entry points.
#include "pba_host.h"
#include "overlap.h"

static void tu_attrs() {
    static bool done = false;
    if (done) return;
    done = true;
    PBA_BIG_LDS(k_part_sort);
    PBA_BIG_LDS(k_ovl_walk<0>);
}

extern "C" {

// ---------------------------------------------------------------------------------------------
// host API: all-vs-all overlap
// ---------------------------------------------------------------------------------------------
#define PBA_OVL_WALK(NBV)                                                                                             \
    hipLaunchKernelGGL((k_ovl_walk<NBV>), dim3(persistent_grid(ctx, n_items, (NBV) ? 4 : 1, lds)),                        \
                       dim3(PBA_WAVE * ((NBV) ? 4 : 1)), lds * ((NBV) ? 4 : 1), ctx->stream, reads->dev(), t_lo, n_items,  \
                       items, d_off.as<uint32_t>(), d_cand.as<uint64_t>(), ocfg, full_band, redo_in,                     \
                       d_redo.as<uint2>(),                                                                              \
                       (unsigned long long)redo_cap, d_cnt64.as<unsigned long long>() + 2, d_out.as<pba_overlap>(),     \
                       (unsigned long long)cap, d_cnt64.as<unsigned long long>(), d_cnt64.as<unsigned long long>() + 1,  \
                       ctx->d_queue)

int pba_overlap_probes(pba_ctx *ctx, const pba_seqs *reads, uint32_t q_lo, uint32_t q_hi, uint32_t mask, int max_trial,
                       void *d_entries, uint64_t cap, uint64_t *n_out) {
    if (!ctx || !reads || !d_entries || !n_out || q_lo > q_hi || q_hi > reads->n) return PBA_E_INVALID;
    if (max_trial < 1 || 2 * max_trial >= (1 << PBA_OVL_JD_BITS)) PBA_FAIL(PBA_E_INVALID, "max_trial must be in [1, 63]");
    if (reads->n >= (1u << 24)) PBA_FAIL(PBA_E_TOOLONG, "at most 2^24 reads");
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs();
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
    tu_attrs();
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
    if (!ctx || !reads || !n_out || (!out && cap) || t_lo > t_hi || t_hi > reads->n || (!d_probe_entries && n_probe_slots))
        return PBA_E_INVALID;
    if (max_trial < 1 || 2 * max_trial >= (1 << PBA_OVL_JD_BITS)) PBA_FAIL(PBA_E_INVALID, "max_trial must be in [1, 63]");
    if (reads->n >= (1u << 24)) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: at most 2^24 reads");
    if (reads->max_len > (uint32_t)kMaxSeqLen) PBA_FAIL(PBA_E_TOOLONG, "read longer than the engine limit");
    if (reads->non_acgt) PBA_FAIL(PBA_E_ALPHABET, "pba_overlap_all: the read set holds bytes outside ACGT");
    HIPCHK(hipSetDevice(ctx->device));
    tu_attrs();
    *n_out = 0;
    pba_overlap_stats st;
    memset(&st, 0, sizeof st);
    const uint32_t n = reads->n, nt = t_hi - t_lo, t2 = 2u * (uint32_t)max_trial;
    if (nt == 0 || n < 2) { if (stats) *stats = st; return PBA_OK; }
    Plan pl;
    int rc = make_plan(ctx, R, 0, 0, kernel, 1 + (int)(reads->max_len * R), &pl);
    if (rc != PBA_OK) return rc;

    // 1. probe table: the (gathered) probe entries, partitioned and sorted like a seed index
    DevBuf d_cnt64;
    HIPCHK(hipMalloc(&d_cnt64.p, 32));
    (void)hipEventRecord(ctx->ev[2], ctx->stream);
    pba_index *pix = nullptr;
    rc = pba_index_from_entries(ctx, d_probe_entries, n_probe_slots, mask, PBA_INDEX_ALL, 0, &pix);   // identity ordinal -> value: the probe id
    if (rc != PBA_OK) return rc;
    struct IxGuard { pba_index *p; ~IxGuard() { pba_index_destroy(p); } } guard{pix};
    st.n_probe_entries = pix->n_entries;

    // 2. scan the targets' positions against the probe table: count, offsets, fill
    DevBuf d_off, d_cur, d_cand, d_out, d_pres;
    HIPCHK(hipMalloc(&d_pres.p, (size_t)1 << (PBA_OVL_PRES_LOG - 3)));
    HIPCHK(hipMemsetAsync(d_pres.p, 0, (size_t)1 << (PBA_OVL_PRES_LOG - 3), ctx->stream));
    if (pix->n_entries)
        hipLaunchKernelGGL(k_ovl_presence, dim3((uint32_t)((pix->n_entries + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t *)pix->d_ent, (uint64_t)pix->n_entries, d_pres.as<uint32_t>());
    // a direct-address directory of the probe keys (seed_index.h: KeyDir), when the mask's care bits allow one
    KeyDir kd;
    memset(&kd, 0, sizeof kd);
    DevBuf d_dir;
    const int care = __builtin_popcount(mask);
    if (pix->n_entries && pix->n_entries < 0xFFFFFFFFull && care <= PBA_DIR_MAX_BITS) {
        kd.mask = mask; kd.n_entries = (uint32_t)pix->n_entries;
        uint32_t m = mask, mk = ~m << 1;                         // Hacker's Delight 7-4: the move masks of compress(x, m)
        for (int i = 0; i < 5; ++i) {
            uint32_t mp = mk ^ (mk << 1);
            mp ^= mp << 2; mp ^= mp << 4; mp ^= mp << 8; mp ^= mp << 16;
            const uint32_t mv = mp & m;
            kd.mv[i] = mv;
            m = (m ^ mv) | (mv >> (1 << i));
            mk &= ~mp;
        }
        HIPCHK(hipMalloc(&d_dir.p, sizeof(uint32_t) << care));
        HIPCHK(hipMemsetAsync(d_dir.p, 0xFF, sizeof(uint32_t) << care, ctx->stream));
        hipLaunchKernelGGL(k_dir_build, dim3((kd.n_entries + 255) / 256), dim3(256), 0, ctx->stream, (const uint64_t *)pix->d_ent,
                           kd.n_entries, kd, d_dir.as<uint32_t>());
        kd.dir = d_dir.as<uint32_t>();
    }
    HIPCHK(hipMalloc(&d_off.p, sizeof(uint32_t) * (nt + 1)));
    // count per (target, bucket of consecutive queries): PBA_OVL_SUB buckets per target
    const uint64_t nsub = (uint64_t)nt * PBA_OVL_SUB;
    const uint32_t sub_mul = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, ((uint64_t)PBA_OVL_SUB << 32) / n);   // bucket = umulhi(q, sub_mul)
    DevBuf d_sub;
    HIPCHK(hipMalloc(&d_sub.p, sizeof(uint32_t) * (nsub + 1)));
    hipLaunchKernelGGL(k_ovl_scan<false>, dim3(nt), dim3(256), 0, ctx->stream, pix->dev(), kd, d_pres.as<uint32_t>(), reads->dev(),
                       t_lo, nt, t2, sub_mul, 0, d_sub.as<uint32_t>(), (uint64_t *)nullptr);
    std::vector<uint32_t> h_sub(nsub + 1), h_cnt(nt + 1), h_off(nt + 1);
    HIPCHK(hipMemcpyAsync(h_sub.data(), d_sub.p, sizeof(uint32_t) * nsub, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    // the coarsest split (2^shift neighbouring buckets merged) whose pieces all fit the LDS sort
    int shift = 0;
    for (int sh = 6; sh >= 0; --sh) {                            // PBA_OVL_SUB = 2^6
        bool fits = true;
        for (uint64_t p0 = 0; p0 < nsub && fits; p0 += (1ull << sh)) {
            uint64_t c = 0;
            for (uint64_t k = 0; k < (1ull << sh); ++k) c += h_sub[p0 + k];
            fits = c <= PBA_IX_LDS_SORT_CAP;
        }
        if (fits) { shift = sh; break; }
    }
    const uint32_t per_t = PBA_OVL_SUB >> shift;                  // sorted pieces per target
    const uint64_t npiece = (uint64_t)nt * per_t;
    std::vector<uint32_t> h_poff(npiece + 1);
    uint64_t total = 0;
    uint32_t biggest = 2;
    std::vector<uint32_t> oversize;                               // pieces that outgrow the LDS sort even at the finest split
    for (uint32_t i = 0; i < nt; ++i) {
        h_off[i] = (uint32_t)total;
        for (uint32_t pc = 0; pc < per_t; ++pc) {
            uint64_t c = 0;
            for (uint32_t k = 0; k < (1u << shift); ++k) c += h_sub[(uint64_t)i * PBA_OVL_SUB + ((uint64_t)pc << shift) + k];
            h_poff[(uint64_t)i * per_t + pc] = (uint32_t)total;
            if (c <= PBA_IX_LDS_SORT_CAP) biggest = std::max<uint32_t>(biggest, (uint32_t)c);
            else oversize.push_back((uint32_t)((uint64_t)i * per_t + pc));
            total += c;
            if (total > 0xFFFFFFF0ull) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: more than 2^32 candidates in one target range; shard it");
        }
        h_cnt[i] = (uint32_t)(total - h_off[i]);
    }
    h_off[nt] = (uint32_t)total;
    h_poff[npiece] = (uint32_t)total;
    st.n_candidates = total;
    DevBuf d_poff;
    HIPCHK(hipMalloc(&d_cand.p, sizeof(uint64_t) * (total + 1)));
    HIPCHK(hipMalloc(&d_poff.p, sizeof(uint32_t) * (npiece + 1)));
    HIPCHK(hipMalloc(&d_cur.p, sizeof(uint32_t) * (npiece + 1)));
    HIPCHK(hipMemcpyAsync(d_off.p, h_off.data(), sizeof(uint32_t) * (nt + 1), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_poff.p, h_poff.data(), sizeof(uint32_t) * (npiece + 1), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_cur.p, h_poff.data(), sizeof(uint32_t) * (npiece + 1), hipMemcpyHostToDevice, ctx->stream));
    if (total) {
        hipLaunchKernelGGL(k_ovl_scan<true>, dim3(nt), dim3(256), 0, ctx->stream, pix->dev(), kd, d_pres.as<uint32_t>(),
                           reads->dev(), t_lo, nt, t2, sub_mul, shift, d_cur.as<uint32_t>(), d_cand.as<uint64_t>());
        (void)hipEventRecord(ctx->ev[3], ctx->stream);
        // 3. sort every piece = the reference's try order inside every (target, query); the pieces of a target in
        //    order are its list in query order
        uint32_t pow2 = 2;
        while (pow2 < biggest) pow2 <<= 1;
        for (uint64_t p0 = 0; p0 < npiece; p0 += 0x40000000ull) {  // (grid dimension limit)
            const uint32_t chunk = (uint32_t)std::min<uint64_t>(npiece - p0, 0x40000000ull);
            // (a big piece takes most of a CU's LDS, so its workgroup is the only one there: 1 024 threads keep the CU busy)
            hipLaunchKernelGGL(k_part_sort, dim3(chunk), dim3(pow2 >= 4096 ? 1024 : 256), sizeof(uint64_t) * pow2, ctx->stream, d_cand.as<uint64_t>(),
                               d_poff.as<uint32_t>() + p0);
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipGetLastError());
        for (uint32_t pc : oversize) {                            // e.g. one query with > 16 384 candidates on a target
            rc = sort_partition_global(ctx, d_cand.as<uint64_t>() + h_poff[pc], h_poff[pc + 1] - h_poff[pc]);
            if (rc != PBA_OK) return rc;
        }
    } else {
        (void)hipEventRecord(ctx->ev[3], ctx->stream);
    }
    (void)hipEventRecord(ctx->ev[4], ctx->stream);

    // 4. walk: persistent wavefronts, one target at a time, narrow window; then the parked (target, query) runs
    //    at the reference band
    HIPCHK(hipMalloc(&d_out.p, sizeof(pba_overlap) * (cap + 1)));
    HIPCHK(hipMemsetAsync(d_cnt64.p, 0, 32, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_queue, 0, 4, ctx->stream));
    OvlCfg ocfg;
    ocfg.R = R; ocfg.overlap_min = overlap_min; ocfg.row_cap = pl.cfg.row_cap; ocfg.t2 = t2;
    const size_t lds = pl.lds;
    DevBuf d_redo, d_items;
    const uint64_t redo_cap = std::max<uint64_t>(1024, total / 4);
    HIPCHK(hipMalloc(&d_redo.p, sizeof(uint2) * redo_cap));
    // work items: (target, first candidate of a group of 64), expanded on the device from the per-target item counts
    // (a million reads make 22 M items per call: building and copying them from the host took longer than a scan pass)
    std::vector<uint32_t> h_ipre(nt + 1);
    uint64_t n_items64 = 0;
    for (uint32_t i = 0; i < nt; ++i) { h_ipre[i] = (uint32_t)n_items64; n_items64 += (h_cnt[i] + PBA_WAVE - 1) / PBA_WAVE; }
    if (n_items64 >= 0xFFFFFFFFull) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: too many work items per call, use a smaller target range");
    h_ipre[nt] = (uint32_t)n_items64;
    DevBuf d_ipre;
    HIPCHK(hipMalloc(&d_ipre.p, sizeof(uint32_t) * (nt + 1)));
    HIPCHK(hipMemcpyAsync(d_ipre.p, h_ipre.data(), sizeof(uint32_t) * (nt + 1), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMalloc(&d_items.p, sizeof(uint2) * (n_items64 + 1)));
    if (n_items64)
        hipLaunchKernelGGL(k_ovl_items, dim3((uint32_t)((n_items64 + 255) / 256)), dim3(256), 0, ctx->stream, d_ipre.as<uint32_t>(),
                           d_off.as<uint32_t>(), nt, (uint32_t)n_items64, d_items.as<uint2>());
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
            DevBuf d_in;
            HIPCHK(hipMalloc(&d_in.p, sizeof(uint2) * h_redo));
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
    const size_t n_sample = pl.nb1 == 0 ? n_all : std::min(n_all, std::max<size_t>(sample_min, n_all / 32));
    uint64_t parked = 0;
    rc = narrow_then_redo(pl.nb1, 0, n_sample, &parked);
    if (rc != PBA_OK) return rc;
    parked_total += parked;
    if (n_sample < n_all) {
        unsigned long long h_ov = 0;
        HIPCHK(hipMemcpyAsync(&h_ov, d_cnt64.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (2 * parked > h_ov) {                                 // most overlaps of the sample needed more than the narrow window
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
    st.n_pairs = h_cnt2[1];
    (void)hipEventElapsedTime(&st.scan_ms, ctx->ev[2], ctx->ev[3]);
    (void)hipEventElapsedTime(&st.sort_ms, ctx->ev[3], ctx->ev[4]);
    (void)hipEventElapsedTime(&st.walk_ms, ctx->ev[4], ctx->ev[5]);
    if (stats) *stats = st;
    return PBA_OK;
}

}  // extern "C"
