// pba_core.hip -- context, sequence sets and the seed-hit index: kernels and the device half of the C ABI (include/pba.h).
// One process per GPU, one pba_ctx per process, one HIP stream per ctx.  Everything here fails loudly
// (PBA_E_NODEVICE / PBA_E_HIP): there is no CPU path behind these entry points.
#include "pba_host.h"
#include "pba_internal.h"

// ---------------------------------------------------------------------------------------------
// kernels: packing
// ---------------------------------------------------------------------------------------------
// ASCII -> 2-bit, 16 chars per thread into one packed dword.  Thread t owns packed dword t of the
// whole set; its sequence is found by bisection over the (16-byte aligned) packed offsets.
__global__ void __launch_bounds__(256)
k_pack_text(const uint8_t *text, const uint64_t *text_off, const uint64_t *pk_off, const uint32_t *len, uint32_t n,
            uint64_t total_dwords, uint8_t *packed, int strict, uint32_t *bad) {
    // (grid-stride: a launch's global size is 32 bits -- a thread per dword of a 37 GB set would not fit; see elem_grid)
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total_dwords; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t byte = t * 4;
        uint32_t lo = 0, hi = n;            // last s with pk_off[s] <= byte
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pk_off[mid] <= byte) lo = mid; else hi = mid;
        }
        const uint32_t s = lo;
        const uint32_t L = len[s];
        const uint64_t w = (byte - pk_off[s]) >> 2;          // dword index inside the sequence
        if (w * 16 >= L) continue;                           // alignment padding
        const uint8_t *src = text + text_off[s] + w * 16;
        const uint32_t nb = (uint32_t)min((uint64_t)16, (uint64_t)L - w * 16);
        uint32_t word = 0, notacgt = 0;
        for (uint32_t k = 0; k < nb; ++k) {
            const uint32_t ch = src[k];
            const uint32_t code = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : 3u;   // C2I, dna_seq.h:21
            notacgt |= (code == 3u && ch != 'T');
            word |= code << (8 * (k >> 2) + 6 - 2 * (k & 3));   // byte k/4, first base in bits 7:6
        }
        *reinterpret_cast<uint32_t *>(packed + byte) = word;    // padding bytes of the last dword stay 0
        if (notacgt) atomicOr(bad, 1u);
    }
}

// Bit planes of a packed set (dev_common.h: SeqSetDev::plane): thread w owns plane word w of the whole set.
__global__ void __launch_bounds__(256)
k_make_planes(const uint8_t *packed, const uint64_t *off, const uint32_t *len, const uint64_t *poff, uint32_t n,
              uint64_t total_words, uint32_t *plane) {
    // (grid-stride: ten million 15 kb reads are 4.7 G plane words, more than a launch's 32-bit global size holds -- a thread per
    // word silently ran the first 2^32-th of them only, tools/rehearse_config4.py; see elem_grid)
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = n;            // last s with poff[s] <= w
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (poff[mid] <= w) lo = mid; else hi = mid;
        }
        const uint32_t s = lo;
        const uint64_t k = w - poff[s];     // word index inside the sequence
        const uint32_t L = len[s];
        uint32_t plo = 0, phi = 0;
        if (k * 32 < L) {
            planes_from_packed(packed + off[s], (int)(k * 32), plo, phi);
            const uint32_t valid = L - (uint32_t)(k * 32);
            if (valid < 32) { plo &= (1u << valid) - 1u; phi &= (1u << valid) - 1u; }   // nothing of the neighbour's bytes
        }
        plane[2 * w] = plo;                  // the two planes side by side: one line serves both (align_bitvec.h: load_planes32)
        plane[2 * w + 1] = phi;
    }
}


extern "C" {

int pba_ctx_create(int device_id, pba_ctx **out) {
    if (!out) return PBA_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PBA_E_NODEVICE;
    if (device_id < 0 || device_id >= ndev) return PBA_E_NODEVICE;
    pba_ctx *ctx = new (std::nothrow) pba_ctx();
    if (!ctx) return PBA_E_NOMEM;
    ctx->device = device_id;
    ctx->err[0] = 0;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device_id) != hipSuccess) {
        delete ctx;
        return PBA_E_NODEVICE;
    }
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {   // the code object holds gfx950 ISA only
        delete ctx;
        return PBA_E_NODEVICE;
    }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return PBA_E_HIP;
    }
    ctx->stream = ctx->own_stream;
    for (int i = 0; i < 6; ++i)
        if (hipEventCreate(&ctx->ev[i]) != hipSuccess) { delete ctx; return PBA_E_HIP; }
    if (hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming) != hipSuccess) { delete ctx; return PBA_E_HIP; }
    memset(&ctx->prof, 0, sizeof ctx->prof);
    if (hipMalloc((void **)&ctx->d_queue, 64) != hipSuccess) { delete ctx; return PBA_E_NOMEM; }
    ctx->d_scratch = nullptr; ctx->scratch_bytes = 0;
    memset(ctx->pool, 0, sizeof ctx->pool);
    ctx->h_stage = nullptr; ctx->h_stage_cap = 0; ctx->attr_done = 0;
    memset(&ctx->ix_cache, 0, sizeof ctx->ix_cache);
    // (kernels that take more than the default 64 KB of dynamic LDS are given the attribute by the translation unit
    // that launches them: tu_attrs() in each .hip)
    *out = ctx;
    return PBA_OK;
}

void pba_ctx_destroy(pba_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->own_stream);
    for (int i = 0; i < 6; ++i) (void)hipEventDestroy(ctx->ev[i]);
    (void)hipEventDestroy(ctx->ev_aux);
    (void)hipFree(ctx->d_queue);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    for (auto &b : ctx->pool) if (b.p) (void)hipFree(b.p);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->ix_cache.ent) (void)hipFree(ctx->ix_cache.ent);
    if (ctx->ix_cache.ent2) (void)hipFree(ctx->ix_cache.ent2);
    if (ctx->ix_cache.off) (void)hipFree(ctx->ix_cache.off);
    delete ctx;
}

int pba_ctx_trim(pba_ctx *ctx) {
    if (!ctx) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (auto &b : ctx->pool) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    if (ctx->h_stage) { (void)hipHostFree(ctx->h_stage); ctx->h_stage = nullptr; ctx->h_stage_cap = 0; }
    if (ctx->ix_cache.ent) (void)hipFree(ctx->ix_cache.ent);
    if (ctx->ix_cache.ent2) (void)hipFree(ctx->ix_cache.ent2);
    if (ctx->ix_cache.off) (void)hipFree(ctx->ix_cache.off);
    memset(&ctx->ix_cache, 0, sizeof ctx->ix_cache);
    if (ctx->d_scratch) { (void)hipFree(ctx->d_scratch); ctx->d_scratch = nullptr; ctx->scratch_bytes = 0; }
    return PBA_OK;
}

const char *pba_ctx_error(const pba_ctx *ctx) { return ctx ? ctx->err : "null ctx"; }

int pba_ctx_set_stream(pba_ctx *ctx, void *hip_stream) {
    if (!ctx) return PBA_E_INVALID;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return PBA_OK;
}

int pba_ctx_sync(pba_ctx *ctx) {
    if (!ctx) return PBA_E_INVALID;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

int pba_ctx_last_profile(const pba_ctx *ctx, pba_profile *out) {
    if (!ctx || !out) return PBA_E_INVALID;
    *out = ctx->prof;
    return PBA_OK;
}

int pba_ctx_device_info(const pba_ctx *ctx, char *name, size_t cap, int *n_cu, int *clock_mhz, uint64_t *hbm) {
    if (!ctx) return PBA_E_INVALID;
    if (name && cap) snprintf(name, cap, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = ctx->prop.clockRate / 1000;
    if (hbm) *hbm = (uint64_t)ctx->prop.totalGlobalMem;
    return PBA_OK;
}

// ---------------------------------------------------------------------------------------------
// host API: sequence sets
// ---------------------------------------------------------------------------------------------
static int seqs_alloc(pba_ctx *ctx, pba_seqs *s, uint64_t packed_bytes) {
    s->packed_bytes = packed_bytes;
    HIPCHK(hipMalloc((void **)&s->d_alloc, packed_bytes + 2 * kSlack));
    HIPCHK(memset_big(s->d_alloc, 0, packed_bytes + 2 * kSlack, ctx->stream));
    s->d_packed = s->d_alloc + kSlack;
    HIPCHK(hipMalloc((void **)&s->d_off, sizeof(uint64_t) * (s->n + 1)));
    HIPCHK(hipMalloc((void **)&s->d_len, sizeof(uint32_t) * (s->n + 1)));
    HIPCHK(hipMemcpyAsync(s->d_off, s->h_off.data(), sizeof(uint64_t) * s->n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(s->d_len, s->h_len.data(), sizeof(uint32_t) * s->n, hipMemcpyHostToDevice, ctx->stream));
    return PBA_OK;
}

// the bit planes of a set whose packed bytes, offsets and lengths are on the device (enqueued on the ctx's stream)
static int seqs_planes(pba_ctx *ctx, pba_seqs *s) {
    std::vector<uint64_t> poff(s->n + 1);
    uint64_t w = 0;
    for (uint32_t i = 0; i < s->n; ++i) { poff[i] = w; w += ((uint64_t)s->h_len[i] + 31) / 32; }
    poff[s->n] = w;
    s->plane_words = w + 2 * kPlaneSlack;
    HIPCHK(hipMalloc((void **)&s->d_planes, s->plane_words * 2 * sizeof(uint32_t)));
    HIPCHK(memset_big(s->d_planes, 0, s->plane_words * 2 * sizeof(uint32_t), ctx->stream));
    HIPCHK(hipMalloc((void **)&s->d_poff, sizeof(uint64_t) * (s->n + 1)));
    HIPCHK(hipMemcpyAsync(s->d_poff, poff.data(), sizeof(uint64_t) * (s->n + 1), hipMemcpyHostToDevice, ctx->stream));
    if (w) {
        hipLaunchKernelGGL(k_make_planes, dim3(elem_grid(w, 256)), dim3(256), 0, ctx->stream, s->d_packed, s->d_off, s->d_len,
                           s->d_poff, s->n, w, s->d_planes + 2 * kPlaneSlack);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));      // poff (host vector) must outlive the copy
    return PBA_OK;
}

void pba_seqs_destroy(pba_seqs *s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    if (s->d_planes) (void)hipFree(s->d_planes);
    if (s->d_poff) (void)hipFree(s->d_poff);
    if (s->d_alloc) (void)hipFree(s->d_alloc);
    if (s->d_off) (void)hipFree(s->d_off);
    if (s->d_len) (void)hipFree(s->d_len);
    delete s;
}

// shared tail of the two text constructors: d_text / d_toff are on the device, h_toff on the host
static int seqs_pack(pba_ctx *ctx, const uint8_t *d_text, const uint64_t *d_toff, const uint64_t *h_toff, uint32_t n,
                     int strict, pba_seqs **out) {
    pba_seqs *s = new (std::nothrow) pba_seqs();
    if (!s) PBA_FAIL(PBA_E_NOMEM, "pba_seqs");
    s->ctx = ctx; s->n = n; s->max_len = 0; s->non_acgt = false; s->d_alloc = nullptr; s->d_packed = nullptr; s->d_off = nullptr; s->d_len = nullptr; s->d_planes = nullptr; s->d_poff = nullptr; s->plane_words = 0;
    s->h_off.resize(n + 1); s->h_len.resize(n + 1);
    uint64_t pk = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (h_toff[i + 1] < h_toff[i] || h_toff[i + 1] - h_toff[i] > 0x7FFFFFF0ull) {
            delete s;
            PBA_FAIL(PBA_E_INVALID, "offsets must be non-decreasing and each sequence < 2^31 bases");
        }
        const uint32_t L = (uint32_t)(h_toff[i + 1] - h_toff[i]);
        s->h_off[i] = pk; s->h_len[i] = L;
        s->max_len = std::max(s->max_len, L);
        pk += (((uint64_t)L + 3) / 4 + 15) & ~15ull;     // every sequence starts 16-byte aligned
    }
    s->h_off[n] = pk; s->h_len[n] = 0;
    int st = seqs_alloc(ctx, s, pk);
    if (st != PBA_OK) { pba_seqs_destroy(s); return st; }
    DevBuf bad;
    if (hipMalloc(&bad.p, 4) != hipSuccess) { pba_seqs_destroy(s); PBA_FAIL(PBA_E_NOMEM, "hipMalloc"); }
    (void)hipMemsetAsync(bad.p, 0, 4, ctx->stream);
    // the kernel bisects over n+1 offsets: upload the end offset too
    (void)hipMemcpyAsync(s->d_off + n, &s->h_off[n], sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
    const uint64_t total_dwords = pk / 4;
    if (total_dwords) {
        hipLaunchKernelGGL(k_pack_text, dim3(elem_grid(total_dwords, 256)), dim3(256), 0, ctx->stream, d_text, d_toff, s->d_off,
                           s->d_len, n, total_dwords, s->d_packed, strict, bad.as<uint32_t>());
    }
    uint32_t h_bad = 0;
    hipError_t e = hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { pba_seqs_destroy(s); return ctx_fail(ctx, PBA_E_HIP, "k_pack_text", e); }
    if (strict && h_bad) { pba_seqs_destroy(s); PBA_FAIL(PBA_E_ALPHABET, "pba_seqs_from_text"); }
    s->non_acgt = h_bad != 0;
    st = seqs_planes(ctx, s);
    if (st != PBA_OK) { pba_seqs_destroy(s); return st; }
    *out = s;
    return PBA_OK;
}

int pba_seqs_from_text(pba_ctx *ctx, const char *text, const uint64_t *offsets, uint32_t n, int strict_acgt,
                       pba_seqs **out) {
    if (!ctx || !offsets || !out || (!text && n && offsets[n] > offsets[0])) return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t total = n ? offsets[n] : 0;
    DevBuf d_text, d_toff;
    HIPCHK(hipMalloc(&d_text.p, total + kSlack));
    HIPCHK(hipMalloc(&d_toff.p, sizeof(uint64_t) * (n + 1)));
    if (total) HIPCHK(hipMemcpyAsync(d_text.p, text, total, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_toff.p, offsets, sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    return seqs_pack(ctx, d_text.as<uint8_t>(), d_toff.as<uint64_t>(), offsets, n, strict_acgt, out);
}

int pba_seqs_from_device_text(pba_ctx *ctx, const void *d_text, const void *d_offsets, uint32_t n, uint64_t total_bytes,
                              uint32_t max_len, pba_seqs **out) {
    (void)total_bytes; (void)max_len;
    if (!ctx || !d_offsets || !out || (!d_text && n)) return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<uint64_t> h_toff(n + 1);
    HIPCHK(hipMemcpyAsync(h_toff.data(), d_offsets, sizeof(uint64_t) * (n + 1), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return seqs_pack(ctx, (const uint8_t *)d_text, (const uint64_t *)d_offsets, h_toff.data(), n, 0, out);
}

int pba_seqs_from_records(pba_ctx *ctx, const uint8_t *file, size_t file_len, uint32_t min_excl, uint32_t max_excl,
                          pba_seqs **out) {
    if (!ctx || !out || (!file && file_len)) return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    size_t total = 0;
    const size_t kept = pba_open_binary(file, file_len, min_excl, max_excl, nullptr, 0, &total);
    if (kept > 0x7FFFFFFFull) PBA_FAIL(PBA_E_TOOLONG, "too many records");
    std::vector<uint64_t> recs(kept + 1);
    pba_open_binary(file, file_len, min_excl, max_excl, recs.data(), kept, nullptr);
    pba_seqs *s = new (std::nothrow) pba_seqs();
    if (!s) PBA_FAIL(PBA_E_NOMEM, "pba_seqs");
    s->ctx = ctx; s->n = (uint32_t)kept; s->max_len = 0; s->non_acgt = false; s->d_alloc = nullptr; s->d_packed = nullptr; s->d_off = nullptr; s->d_len = nullptr; s->d_planes = nullptr; s->d_poff = nullptr; s->plane_words = 0;
    s->h_off.resize(kept + 1); s->h_len.resize(kept + 1);
    for (size_t i = 0; i < kept; ++i) {
        uint32_t L;
        memcpy(&L, file + recs[i], 4);
        if (recs[i] + 4 + ((uint64_t)L + 3) / 4 > file_len) { delete s; PBA_FAIL(PBA_E_INVALID, "truncated record"); }
        s->h_off[i] = recs[i] + 4;      // payload follows the u32 length (dna_seq.h:119-121)
        s->h_len[i] = L;
        s->max_len = std::max(s->max_len, L);
    }
    s->h_off[kept] = file_len; s->h_len[kept] = 0;
    // the file image goes up as it is (no re-packing); the slack after it is large enough for
    // seed_at's byte-offset reads to stay inside the allocation and read zeros
    const uint64_t slack = 65536;
    s->packed_bytes = file_len;
    hipError_t e = hipMalloc((void **)&s->d_alloc, file_len + slack + kSlack);
    if (e == hipSuccess) e = hipMemsetAsync(s->d_alloc, 0, file_len + slack + kSlack, ctx->stream);
    if (e == hipSuccess) s->d_packed = s->d_alloc + kSlack;
    if (e == hipSuccess && file_len) e = hipMemcpyAsync(s->d_packed, file, file_len, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_off, sizeof(uint64_t) * (kept + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_len, sizeof(uint32_t) * (kept + 1));
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_off, s->h_off.data(), sizeof(uint64_t) * (kept + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_len, s->h_len.data(), sizeof(uint32_t) * (kept + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { pba_seqs_destroy(s); return ctx_fail(ctx, PBA_E_HIP, "pba_seqs_from_records", e); }
    const int stp = seqs_planes(ctx, s);
    if (stp != PBA_OK) { pba_seqs_destroy(s); return stp; }
    *out = s;
    return PBA_OK;
}

int pba_seqs_export(pba_ctx *ctx, const pba_seqs *s, void *d_dst, uint64_t cap, uint64_t *offsets) {
    if (!ctx || !s || !offsets || (!d_dst && s->packed_bytes)) return PBA_E_INVALID;
    if (cap < s->packed_bytes) PBA_FAIL(PBA_E_INVALID, "pba_seqs_export: buffer smaller than pba_seqs_packed_bytes");
    HIPCHK(hipSetDevice(ctx->device));
    if (s->packed_bytes) HIPCHK(copy_d2d(d_dst, s->d_packed, s->packed_bytes, ctx->stream));
    for (uint32_t i = 0; i < s->n; ++i) offsets[i] = s->h_off[i];
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

int pba_seqs_from_device_packed(pba_ctx *ctx, const void *d_packed, uint64_t n_bytes, const uint64_t *offsets, const uint32_t *lengths,
                                uint32_t n, int non_acgt, pba_seqs **out) {
    if (!ctx || !out || (!d_packed && n_bytes) || (n && (!offsets || !lengths))) return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    pba_seqs *s = new (std::nothrow) pba_seqs();
    if (!s) PBA_FAIL(PBA_E_NOMEM, "pba_seqs");
    s->ctx = ctx; s->n = n; s->max_len = 0; s->non_acgt = non_acgt != 0; s->d_alloc = nullptr; s->d_packed = nullptr; s->d_off = nullptr; s->d_len = nullptr; s->d_planes = nullptr; s->d_poff = nullptr; s->plane_words = 0;
    s->h_off.resize((size_t)n + 1); s->h_len.resize((size_t)n + 1);
    for (uint32_t i = 0; i < n; ++i) {
        if (offsets[i] + ((uint64_t)lengths[i] + 3) / 4 > n_bytes) { delete s; PBA_FAIL(PBA_E_INVALID, "pba_seqs_from_device_packed: sequence outside the buffer"); }
        s->h_off[i] = offsets[i]; s->h_len[i] = lengths[i];
        s->max_len = std::max(s->max_len, lengths[i]);
    }
    s->h_off[n] = n_bytes; s->h_len[n] = 0;
    s->packed_bytes = n_bytes;
    // like a binary read file: the bytes go in as they are, offsets point into them; slack on both sides for the streaming reads
    hipError_t e = hipMalloc((void **)&s->d_alloc, n_bytes + 2 * kSlack);
    if (e == hipSuccess) e = hipMemsetAsync(s->d_alloc, 0, kSlack, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->d_alloc + kSlack + n_bytes, 0, kSlack, ctx->stream);
    if (e == hipSuccess) s->d_packed = s->d_alloc + kSlack;
    if (e == hipSuccess && n_bytes) e = copy_d2d(s->d_packed, d_packed, n_bytes, ctx->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_off, sizeof(uint64_t) * ((size_t)n + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_len, sizeof(uint32_t) * ((size_t)n + 1));
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_off, s->h_off.data(), sizeof(uint64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_len, s->h_len.data(), sizeof(uint32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { pba_seqs_destroy(s); return ctx_fail(ctx, PBA_E_HIP, "pba_seqs_from_device_packed", e); }
    const int stp = seqs_planes(ctx, s);
    if (stp != PBA_OK) { pba_seqs_destroy(s); return stp; }
    *out = s;
    return PBA_OK;
}

int pba_seqs_non_acgt(const pba_seqs *s) { return s && s->non_acgt ? 1 : 0; }

uint32_t pba_seqs_count(const pba_seqs *s) { return s ? s->n : 0; }
uint32_t pba_seqs_max_len(const pba_seqs *s) { return s ? s->max_len : 0; }
uint64_t pba_seqs_packed_bytes(const pba_seqs *s) { return s ? s->packed_bytes : 0; }

int pba_seqs_lengths(const pba_seqs *s, uint32_t *lengths, uint32_t cap) {
    if (!s || !lengths) return PBA_E_INVALID;
    for (uint32_t i = 0; i < s->n && i < cap; ++i) lengths[i] = s->h_len[i];
    return PBA_OK;
}

int pba_seqs_get_text(pba_ctx *ctx, const pba_seqs *s, uint32_t i, char *text, size_t cap) {
    if (!ctx || !s || !text || i >= s->n) return PBA_E_INVALID;
    const uint32_t L = s->h_len[i];
    if (cap <= L) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<uint8_t> pk(((size_t)L + 3) / 4 + 1);
    if (L) HIPCHK(hipMemcpyAsync(pk.data(), s->d_packed + s->h_off[i], ((size_t)L + 3) / 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    static const char base[4] = {'A', 'C', 'G', 'T'};
    for (uint32_t k = 0; k < L; ++k) text[k] = base[(pk[k >> 2] >> (6 - 2 * (k & 3))) & 3];
    text[L] = 0;
    return PBA_OK;
}

// ---------------------------------------------------------------------------------------------
// host API: seed index
// ---------------------------------------------------------------------------------------------
void pba_index_destroy(pba_index *ix) {
    if (!ix) return;
    pba_ctx *ctx = ix->ctx;
    (void)hipSetDevice(ctx->device);
    // the arrays go to the ctx's one-deep cache when it is empty (the work that used them was synchronised by the call
    // that returned its results), else back to the device
    if (ix->d_ent && !ctx->ix_cache.ent) { ctx->ix_cache.ent = ix->d_ent; ctx->ix_cache.ent_cap = ix->ent_cap; }
    else if (ix->d_ent) (void)hipFree(ix->d_ent);
    if (ix->d_part_off && !ctx->ix_cache.off) { ctx->ix_cache.off = ix->d_part_off; ctx->ix_cache.off_cap = ix->off_cap; }
    else if (ix->d_part_off) (void)hipFree(ix->d_part_off);
    delete ix;
}

// device array of at least `bytes` for a new index: the cached one if it is large enough
static int ix_alloc(pba_ctx *ctx, void **cache, size_t *cache_cap, size_t bytes, void **out, size_t *cap) {
    if (*cache && *cache_cap >= bytes) { *out = *cache; *cap = *cache_cap; *cache = nullptr; *cache_cap = 0; return PBA_OK; }
    HIPCHK(hipMalloc(out, bytes));
    *cap = bytes;
    return PBA_OK;
}

uint64_t pba_index_entries(const pba_index *ix) { return ix ? ix->n_entries : 0; }
uint32_t pba_index_visited(const pba_index *ix) { return ix ? ix->visited : 0; }

// sort one oversize partition in global memory
int sort_partition_global(pba_ctx *ctx, uint64_t *d_part, uint32_t n) {
    uint32_t N = 2;
    while (N < n) N <<= 1;
    DevBuf tmp;
    HIPCHK(hipMalloc(&tmp.p, sizeof(uint64_t) * N));
    HIPCHK(hipMemcpyAsync(tmp.p, d_part, sizeof(uint64_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
    if (N > n)
        hipLaunchKernelGGL(k_fill_u64, dim3((N - n + 255) / 256), dim3(256), 0, ctx->stream, tmp.as<uint64_t>(), n, N,
                           ~0ull);
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1)
            hipLaunchKernelGGL(k_bitonic_step, dim3((N / 2 + 255) / 256), dim3(256), 0, ctx->stream, tmp.as<uint64_t>(),
                               N, k, j);
    HIPCHK(hipMemcpyAsync(d_part, tmp.p, sizeof(uint64_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

// the reference's visiting order as at most two position segments (+ what get_seedmap returns)
struct VisitPlan {
    ScanSeg segs[2];
    int nseg;
    uint32_t visited, nhead;
    int32_t tail_top;
};

static VisitPlan visit_plan(uint32_t len, int mode) {
    VisitPlan v;
    v.nseg = 0; v.visited = 0; v.nhead = 0xFFFFFFFFu; v.tail_top = 0;
    if (mode == PBA_INDEX_ALL) {                       // locator.cpp:62: for i in [0, len)
        if (len) v.segs[v.nseg++] = ScanSeg{0, len, 0, 0};
        v.visited = len;
    } else {                                           // ref_seq.h:291-311, MAX_READ_LEN = 20000, N_SEQ_WORD = 16
        const long long L = len, nmax = L - 16;
        const long long nh = std::min(nmax, 20000ll);
        const long long nt = std::min(L - 20000 - 16, 20000ll);
        v.nhead = nh > 0 ? (uint32_t)nh : 0;
        v.tail_top = (int32_t)(L - 16);
        if (nh > 0) v.segs[v.nseg++] = ScanSeg{0, (uint32_t)nh, 0, 0};
        if (nt > 0) v.segs[v.nseg++] = ScanSeg{(uint32_t)(L - 16 - nt + 1), (uint32_t)(L - 16 + 1), v.nhead, 1};
        v.visited = (uint32_t)(nh + (nt < 0 ? 0 : nt));   // ref_seq.h:310 (a negative nhead is added as it is)
    }
    return v;
}

static uint32_t seg_grid(const ScanSeg &sg, int iters = PBA_IX_TILE_ITERS) {
    const uint64_t chunks = ((uint64_t)sg.hi + 15) / 16 - sg.lo / 16;
    return (uint32_t)((chunks + PBA_IX_TILE_THREADS * iters - 1) / (PBA_IX_TILE_THREADS * iters));
}
// small inputs: tiles of 4 096 entries (one chunk per thread) instead of 16 384 -- a 5 Mb target is 305 big tiles on 256 CUs
static bool index_small_tiles(uint64_t n_upper) { return n_upper <= (32ull << 20); }

// partitions of 1 024 .. 2 048 entries on average
static int index_logp(uint64_t n) {
    int logP = 0;
    while (logP < PBA_IX_MAX_LOGP && (n >> logP) > PBA_IX_PART_AVG) ++logP;
    return logP;
}

static pba_index *index_new(pba_ctx *ctx, uint32_t mask, uint32_t len, int mode, const VisitPlan &v) {
    pba_index *ix = new (std::nothrow) pba_index();
    if (!ix) return nullptr;
    ix->ctx = ctx; ix->mask = mask; ix->seq_len = len; ix->visited = v.visited; ix->nhead = v.nhead;
    ix->tail_top = v.tail_top; ix->mode = mode; ix->n_entries = 0; ix->d_ent = nullptr; ix->d_part_off = nullptr;
    ix->ent_cap = ix->off_cap = 0;
    ix->logP = 0;
    return ix;
}

// in-place inclusive scan of a[0 .. n) on the stream (seed_index.h: k_scan_*); tiles: scratch of n / PBA_SCAN_TILE + 1 u32
// shifted (nullable): see k_scan_small; true when it was written (else the caller copies)
static bool scan_inclusive(pba_ctx *ctx, uint32_t *a, uint64_t n, uint32_t *tiles, uint32_t *shifted = nullptr) {
    if (n && n <= PBA_SCAN_SMALL_MAX) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, ctx->stream, a, (uint32_t)n, shifted);
        return shifted != nullptr;
    }
    const uint32_t n_tiles = (uint32_t)((n + PBA_SCAN_TILE - 1) / PBA_SCAN_TILE);
    if (!n_tiles) return false;
    hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(256), 0, ctx->stream, a, n, tiles);
    if (n_tiles > 1) {
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, tiles, n_tiles);
        hipLaunchKernelGGL(k_scan_add, dim3(n_tiles), dim3(256), 0, ctx->stream, a, n, tiles);
    }
    return false;
}

// The partition levels and the sort.  n_upper: an upper bound of the entries (positions visited / slots of the gathered
// list).  level1(bits, cnt1, cursor, dst): dst == nullptr -> add the sizes of the 2^bits level-1 bins to cnt1[bin];
// else scatter the entries into dst through cursor[bin].  from_list (nullable): the exchange form, whose level 1 runs
// through the generic level kernels on this flat list (all-ones entries = padding).
static int index_levels(pba_ctx *ctx, pba_index *ix, uint64_t n_upper, const uint64_t *from_list,
                        const std::function<void(int, uint32_t *, uint32_t *, uint64_t *)> &level1) {
    const uint32_t tile = index_small_tiles(n_upper) ? 4096u : (uint32_t)PBA_IX_TILE_POS;
    if (n_upper > 0xFFFFFFF0ull) PBA_FAIL(PBA_E_TOOLONG, "more than 2^32 index entries");
    ix->logP = index_logp(n_upper);
    const int logP = ix->logP, n_levels = std::max(1, (logP + PBA_IX_LVL_BITS - 1) / PBA_IX_LVL_BITS);
    const uint64_t P = 1ull << logP;
    // device arrays: two entry buffers (the levels ping-pong; the last one written becomes the index's), the offsets of
    // every level (2^depth + 1 each), cursors, tile tables, scan scratch
    void *ent[2] = {nullptr, nullptr};
    size_t ent_cap[2] = {0, 0};
    int sta = ix_alloc(ctx, &ctx->ix_cache.ent, &ctx->ix_cache.ent_cap, sizeof(uint64_t) * (n_upper + 1), &ent[0], &ent_cap[0]);
    if (sta != PBA_OK) return sta;
    ix->d_ent = (uint64_t *)ent[0]; ix->ent_cap = ent_cap[0];                      // (the index owns it from here on: error paths free it with the index)
    // (the second entry buffer comes from the ctx's one-deep cache of index arrays too when it holds one -- in a loop of
    // build / destroy the two buffers just swap roles -- and goes back there; offsets and work tables are pooled)
    BufRef offs, work;
    size_t second_cap = 0;
    void *second = nullptr;
    if (n_levels > 1) {
        sta = ix_alloc(ctx, &ctx->ix_cache.ent2, &ctx->ix_cache.ent2_cap, sizeof(uint64_t) * (n_upper + 1), &second, &second_cap);
        if (sta != PBA_OK) return sta;
        ent[1] = second; ent_cap[1] = second_cap;
    }
    struct Spare { pba_ctx *ctx; void **p; size_t *cap; ~Spare() {          // the buffer the index does not keep: back to the cache
        if (*p) { if (!ctx->ix_cache.ent2) { ctx->ix_cache.ent2 = *p; ctx->ix_cache.ent2_cap = *cap; } else (void)hipFree(*p); }
    } } spare{ctx, &second, &second_cap};
    // offsets of every level (sum over the levels < 2 P + levels), {0, n} of a gathered list, then what the build wants zero
    // at its start beside them -- [zero_at]: largest partition, partitions beyond the LDS sort, entries; the sort's oversize
    // list -- all zeroed by ONE memset (a fill per array and level was 30 us of a 330 us build at 5 Mb)
    const uint32_t ov_cap = (uint32_t)std::min<uint64_t>(P, 4096);
    const uint64_t offs_words = 2 * P + 2 * (uint64_t)n_levels + 8, zero_at = offs_words, all_words = offs_words + 4 + 1 + ov_cap;
    POOL(POOL_IX_OFFS, sizeof(uint32_t) * all_words, offs.p);
    HIPCHK(hipMemsetAsync(offs.p, 0, sizeof(uint32_t) * all_words, ctx->stream));
    uint32_t *const stat = offs.as<uint32_t>() + zero_at, *const ov = stat + 4;
    POOL(POOL_IX_WORK, sizeof(uint32_t) * (3 * P + 64), work.p);                                     // cursor | tile_pre | scan scratch
    uint32_t *const cursor = work.as<uint32_t>(), *const tile_pre = cursor + P + 8, *const tiles = tile_pre + P + 8;
    uint32_t *off_prev = nullptr, *off_k = offs.as<uint32_t>();
    uint32_t *const list_off = offs.as<uint32_t>() + 2 * P + 2 * (uint64_t)n_levels + 4;    // {0, n}: the one group of a gathered list
    int done = 0, cur = 0;
    sta = stage_reserve(ctx, 64);
    if (sta != PBA_OK) return sta;
    uint32_t *const h_stat = (uint32_t *)ctx->h_stage;                             // pinned (read after the event below): [0] largest partition <= the LDS sort's cap, [1] partitions beyond it, [2] entries
    (void)hipEventRecord(ctx->ev[0], ctx->stream);
    for (int k = 0; k < n_levels; ++k) {
        const int bits = (logP - done + (n_levels - k) - 1) / (n_levels - k);      // the remaining bits, split evenly
        const uint64_t bins = 1ull << (done + bits);
        LvlSrc L;
        uint32_t grid = 0;
        if (k > 0 || from_list) {
            // tiles of the groups: tile_pre[g] = tiles of the groups before g
            L.src = k == 0 ? from_list : (const uint64_t *)ent[cur ^ 1]; L.done = done; L.bits = bits;
            L.n_groups = (uint32_t)(1ull << done); L.tile = tile;
            if (k == 0) {                                                          // one group: the whole list
                const uint32_t h[2] = {0u, (uint32_t)n_upper}, t[2] = {0u, (uint32_t)((n_upper + tile - 1) / tile)};
                HIPCHK(hipMemcpyAsync(list_off, h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(hipMemcpyAsync(tile_pre, t, sizeof t, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));                         // (h, t are locals)
                L.off_prev = list_off;
            } else {
                hipLaunchKernelGGL(k_lvl_tiles, dim3((L.n_groups + 255) / 256), dim3(256), 0, ctx->stream, off_prev, L.n_groups, tile, tile_pre + 1);
                scan_inclusive(ctx, tile_pre + 1, L.n_groups, tiles);
                L.off_prev = off_prev;
            }
            L.tile_pre = tile_pre;
            grid = (uint32_t)(n_upper / tile + L.n_groups + 1);                     // >= the tiles there are; the rest exit
        }
        if (grid) hipLaunchKernelGGL(k_lvl_count, dim3(grid), dim3(PBA_IX_TILE_THREADS), 0, ctx->stream, L, off_k + 1);
        else level1(bits, off_k + 1, nullptr, nullptr);
        if (!scan_inclusive(ctx, off_k + 1, bins, tiles, cursor))
            HIPCHK(hipMemcpyAsync(cursor, off_k, sizeof(uint32_t) * bins, hipMemcpyDeviceToDevice, ctx->stream));
        if (k == n_levels - 1) {
            // The partitions' sizes are final once the last level has been counted: the totals the host decides by (entries,
            // the largest partition the LDS sort takes, partitions beyond it) are queued for the host BEFORE the last scatter
            // and read while it runs -- the sort is launched without the chip waiting for a host round trip (75 us of a
            // 330 us build at 5 Mb).
            hipLaunchKernelGGL(k_part_max, dim3((uint32_t)((P + 255) / 256)), dim3(256), 0, ctx->stream, off_k, (uint32_t)P, stat);
            HIPCHK(hipMemcpyAsync(h_stat, stat, 12, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipEventRecord(ctx->ev_aux, ctx->stream));
        }
        if (grid) hipLaunchKernelGGL(k_lvl_scatter, dim3(grid), dim3(PBA_IX_TILE_THREADS), 0, ctx->stream, L, cursor, (uint64_t *)ent[cur]);
        else level1(bits, nullptr, cursor, (uint64_t *)ent[cur]);
        off_prev = off_k; off_k += bins + 1; done += bits; cur ^= 1;
    }
    cur ^= 1;                                                                      // the buffer the last level wrote
    // totals: entries, the largest partition the LDS sort takes, partitions beyond it (queued before the last scatter)
    HIPCHK(hipEventSynchronize(ctx->ev_aux));
    HIPCHK(hipGetLastError());
    const uint32_t total = h_stat[2];
    ix->n_entries = total;
    sta = ix_alloc(ctx, &ctx->ix_cache.off, &ctx->ix_cache.off_cap, sizeof(uint32_t) * (P + 1), (void **)&ix->d_part_off, &ix->off_cap);
    if (sta != PBA_OK) return sta;
    HIPCHK(hipMemcpyAsync(ix->d_part_off, off_prev, sizeof(uint32_t) * (P + 1), hipMemcpyDeviceToDevice, ctx->stream));
    // the index keeps the buffer the last level wrote; the other one is the temporary
    if (cur == 1) { ix->d_ent = (uint64_t *)ent[1]; ix->ent_cap = ent_cap[1]; second = ent[0]; second_cap = ent_cap[0]; }   // (the first is the spare now)
    if (total) {
        // every partition sorted by key, then insertion order: k_seg_sort (buckets by the key's gathered care bits, sorted in
        // wavefront registers); what it leaves -- partitions beyond 16 384 entries or with a bucket beyond 256: low-complexity
        // targets -- goes through the global bitonic pass
        launch_seg_sort(ctx, ix->d_ent, ix->d_ent, ix->d_part_off, nullptr, P, std::max(h_stat[0], h_stat[1] ? 0xFFFFFFFFu : 0u),
                        seg_bkt_key(ix->mask), ov, ov_cap);
        std::vector<uint32_t> h_ov(1 + ov_cap, 0);
        HIPCHK(hipMemcpyAsync(h_ov.data(), ov, sizeof(uint32_t) * (1 + ov_cap), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipGetLastError());
        if (h_ov[0]) {
            std::vector<uint32_t> h_off(P + 1);
            HIPCHK(hipMemcpyAsync(h_off.data(), ix->d_part_off, sizeof(uint32_t) * (P + 1), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            std::vector<uint32_t> todo;
            if (h_ov[0] > ov_cap) { for (uint64_t q = 0; q < P; ++q) todo.push_back((uint32_t)q); }   // (list overflow: check them all)
            else todo.assign(h_ov.begin() + 1, h_ov.begin() + 1 + h_ov[0]);
            std::sort(todo.begin(), todo.end());
            todo.erase(std::unique(todo.begin(), todo.end()), todo.end());
            for (uint32_t q : todo)
                if (h_off[q + 1] - h_off[q] > 1) {
                    int st = sort_partition_global(ctx, ix->d_ent + h_off[q], h_off[q + 1] - h_off[q]);
                    if (st != PBA_OK) return st;
                }
        }
    }
    (void)hipEventRecord(ctx->ev[1], ctx->stream);
    (void)hipEventSynchronize(ctx->ev[1]);
    HIPCHK(hipGetLastError());
    (void)hipEventElapsedTime(&ctx->prof.index_ms, ctx->ev[0], ctx->ev[1]);
    return PBA_OK;
}

int pba_index_build(pba_ctx *ctx, const pba_seqs *target, uint32_t seq, uint32_t mask, int mode, pba_index **out) {
    if (!ctx || !target || !out || seq >= target->n) return PBA_E_INVALID;
    if (mode != PBA_INDEX_ALL && mode != PBA_INDEX_HEAD_TAIL) return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t len = target->h_len[seq];
    if (len > 0x7FFFFFF0u) PBA_FAIL(PBA_E_TOOLONG, "target sequence");
    const VisitPlan v = visit_plan(len, mode);
    pba_index *ix = index_new(ctx, mask, len, mode, v);
    if (!ix) PBA_FAIL(PBA_E_NOMEM, "pba_index");
    uint64_t npos = 0;
    for (int s = 0; s < v.nseg; ++s) npos += v.segs[s].hi - v.segs[s].lo;
    const uint8_t *d_seq = target->d_packed + target->h_off[seq];
    int st = index_levels(ctx, ix, npos, nullptr, [&](int bits, uint32_t *cnt1, uint32_t *cursor, uint64_t *dst) {
        // (level 1 keeps its tiles of 16 384 positions at any size: with 4 096 the per-bin reservations of four times the
        // workgroups cost more than the idle CUs -- 13 -> 26 us and 35 -> 43 us at 5 Mb)
        for (int s = 0; s < v.nseg; ++s) {
            if (!dst) hipLaunchKernelGGL(k_seed_count<PBA_IX_TILE_ITERS>, dim3(seg_grid(v.segs[s])), dim3(PBA_IX_TILE_THREADS), 0, ctx->stream, d_seq, len,
                                         mask, v.segs[s], bits, cnt1);
            else hipLaunchKernelGGL(k_seed_scatter<PBA_IX_TILE_ITERS>, dim3(seg_grid(v.segs[s])), dim3(PBA_IX_TILE_THREADS), 0, ctx->stream, d_seq, len,
                                    mask, v.segs[s], bits, cursor, dst);
        }
    });
    if (st != PBA_OK) { pba_index_destroy(ix); return st; }
    *out = ix;
    return PBA_OK;
}

int pba_index_scan(pba_ctx *ctx, const pba_seqs *target, uint32_t seq, uint32_t mask, int mode, uint32_t part,
                   uint32_t nparts, void *d_entries, uint64_t cap, uint64_t *n_out) {
    if (!ctx || !target || !d_entries || !n_out || seq >= target->n || nparts == 0 || part >= nparts) return PBA_E_INVALID;
    if (mode != PBA_INDEX_ALL && mode != PBA_INDEX_HEAD_TAIL) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t len = target->h_len[seq];
    const VisitPlan v = visit_plan(len, mode);
    // this rank's contiguous slice of the ordinal space [0, visited)
    const uint64_t nv = v.nseg ? (uint64_t)v.segs[v.nseg - 1].ord0 + (v.segs[v.nseg - 1].hi - v.segs[v.nseg - 1].lo) : 0;
    const uint64_t o_lo = nv * part / nparts, o_hi = nv * (part + 1) / nparts;
    DevBuf counter;
    HIPCHK(hipMalloc(&counter.p, 8));
    HIPCHK(hipMemsetAsync(counter.p, 0, 8, ctx->stream));
    const uint8_t *d_seq = target->d_packed + target->h_off[seq];
    for (int s = 0; s < v.nseg; ++s) {
        const ScanSeg &g = v.segs[s];
        const uint64_t g_lo = g.ord0, g_hi = (uint64_t)g.ord0 + (g.hi - g.lo);
        const uint64_t a = std::max(o_lo, g_lo), b = std::min(o_hi, g_hi);
        if (a >= b) continue;
        ScanSeg c;
        c.descending = g.descending; c.ord0 = (uint32_t)a;
        if (!g.descending) { c.lo = g.lo + (uint32_t)(a - g_lo); c.hi = g.lo + (uint32_t)(b - g_lo); }
        else { c.lo = g.hi - (uint32_t)(b - g_lo); c.hi = g.hi - (uint32_t)(a - g_lo); }
        hipLaunchKernelGGL(k_seed_emit, dim3(seg_grid(c)), dim3(PBA_IX_TILE_THREADS), 0, ctx->stream, d_seq, len, mask, c,
                           (uint64_t *)d_entries, (unsigned long long)cap, counter.as<unsigned long long>());
    }
    HIPCHK(hipGetLastError());
    unsigned long long h_n = 0;
    HIPCHK(hipMemcpyAsync(&h_n, counter.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (h_n > cap) PBA_FAIL(PBA_E_INVALID, "pba_index_scan: entry buffer too small");
    *n_out = h_n;
    return PBA_OK;
}

int pba_index_from_entries(pba_ctx *ctx, const void *d_entries, uint64_t n, uint32_t mask, int mode, uint32_t seq_len,
                           pba_index **out) {
    if (!ctx || !out || (!d_entries && n)) return PBA_E_INVALID;
    if (mode != PBA_INDEX_ALL && mode != PBA_INDEX_HEAD_TAIL) return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    const VisitPlan v = visit_plan(seq_len, mode);
    pba_index *ix = index_new(ctx, mask, seq_len, mode, v);
    if (!ix) PBA_FAIL(PBA_E_NOMEM, "pba_index");
    int st = index_levels(ctx, ix, n, (const uint64_t *)d_entries, [](int, uint32_t *, uint32_t *, uint64_t *) {});
    if (st != PBA_OK) { pba_index_destroy(ix); return st; }
    *out = ix;
    return PBA_OK;
}

int pba_index_dump(pba_ctx *ctx, const pba_index *ix, uint32_t *keys, int32_t *pos, uint64_t cap, uint64_t *n) {
    if (!ctx || !ix || !n) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    *n = ix->n_entries;
    if (!keys || !pos) return PBA_OK;
    std::vector<uint64_t> ent(ix->n_entries + 1);
    if (ix->n_entries)
        HIPCHK(hipMemcpyAsync(ent.data(), ix->d_ent, sizeof(uint64_t) * ix->n_entries, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ent.resize(ix->n_entries);
    std::sort(ent.begin(), ent.end());          // partitions are sorted; this only merges them by key
    for (uint64_t i = 0; i < ent.size() && i < cap; ++i) {
        const uint32_t ord = (uint32_t)ent[i];
        keys[i] = (uint32_t)(ent[i] >> 32);
        pos[i] = ord < ix->nhead ? (int32_t)ord : ix->tail_top - (int32_t)(ord - ix->nhead);
    }
    return PBA_OK;
}

int pba_index_find(pba_ctx *ctx, const pba_index *ix, const uint32_t *keys, uint32_t n_keys, uint64_t *hit_off,
                   int32_t *hit_pos, uint64_t hit_cap) {
    if (!ctx || !ix || !hit_off || (!keys && n_keys)) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    hit_off[0] = 0;
    if (!n_keys) return PBA_OK;
    DevBuf d_keys, d_beg, d_cnt, d_off, d_pos;
    HIPCHK(hipMalloc(&d_keys.p, 4ull * n_keys));
    HIPCHK(hipMalloc(&d_beg.p, 4ull * n_keys));
    HIPCHK(hipMalloc(&d_cnt.p, 4ull * n_keys));
    HIPCHK(hipMemcpyAsync(d_keys.p, keys, 4ull * n_keys, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_find_count, dim3((n_keys + 255) / 256), dim3(256), 0, ctx->stream, ix->dev(),
                       d_keys.as<uint32_t>(), n_keys, d_beg.as<uint32_t>(), d_cnt.as<uint32_t>());
    std::vector<uint32_t> cnt(n_keys);
    HIPCHK(hipMemcpyAsync(cnt.data(), d_cnt.p, 4ull * n_keys, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (uint32_t q = 0; q < n_keys; ++q) hit_off[q + 1] = hit_off[q] + cnt[q];
    const uint64_t total = hit_off[n_keys];
    if (!hit_pos || !total) return PBA_OK;
    const uint64_t ncopy = std::min(total, hit_cap);
    HIPCHK(hipMalloc(&d_off.p, 8ull * (n_keys + 1)));
    HIPCHK(hipMalloc(&d_pos.p, 4ull * (ncopy + 1)));
    HIPCHK(hipMemcpyAsync(d_off.p, hit_off, 8ull * (n_keys + 1), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_find_fill, dim3((n_keys + 255) / 256), dim3(256), 0, ctx->stream, ix->dev(), d_beg.as<uint32_t>(),
                       d_off.as<uint64_t>(), n_keys, d_pos.as<int32_t>(), ncopy);
    HIPCHK(hipMemcpyAsync(hit_pos, d_pos.p, 4ull * ncopy, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    return PBA_OK;
}

}  // extern "C"
