// pba_synth.cpp -- deterministic synthetic workload (bench/test infrastructure, host only).
// Implements the generator SURVEY.md 8d / Appendix C describes (uniform genome, forward-strand
// reads at uniform starts, per-step insert / delete / substitute) on an integer counter RNG so
// that the CPU oracle run and the GPU run see identical bytes on every machine.
#include <stdint.h>
#include <thread>
#include <vector>

#include "pba.h"

namespace {

inline uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

const char kBase[4] = {'A', 'C', 'G', 'T'};

struct Stream {
    uint64_t s;
    explicit Stream(uint64_t seed, uint64_t idx) : s(mix64(seed ^ mix64(idx))) {}
    uint64_t next() { return mix64(s++); }
};

inline int code_of(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }

// reads [r0, r1) of the set; read r goes to slot r - slot0 of out / starts
void gen_reads(uint64_t seed, const char *genome, size_t L, uint32_t r0, uint32_t r1, uint32_t slot0, uint32_t rl, uint64_t t_ins,
               uint64_t t_del, uint64_t t_sub, char *out, uint32_t *starts) {
    const size_t span = (size_t)rl + rl / 2;
    const uint64_t nstart = L > span ? L - span : 1;
    for (uint32_t r = r0; r < r1; ++r) {
        Stream st(seed, r);
        size_t g = (size_t)(st.next() % nstart);
        if (starts) starts[r - slot0] = (uint32_t)g;
        char *o = out + (size_t)(r - slot0) * rl;
        uint32_t emitted = 0;
        while (emitted < rl) {
            const uint64_t x = st.next();
            const uint64_t u = x >> 32;
            const char gb = genome[g < L ? g : g % L];
            if (u < t_ins) {
                o[emitted++] = kBase[x & 3];
            } else if (u < t_del) {
                ++g;
            } else if (u < t_sub) {
                o[emitted++] = kBase[(code_of(gb) + 1 + ((x >> 2) & 0xFFFF) % 3) & 3];
                ++g;
            } else {
                o[emitted++] = gb;
                ++g;
            }
        }
    }
}

}  // namespace

extern "C" {

void pba_synth_genome(uint64_t seed, char *out, size_t n) {
    for (size_t i = 0; i < n; i += 32) {
        uint64_t x = mix64(seed * 0xD1342543DE82EF95ull + i / 32);
        for (size_t k = 0; k < 32 && i + k < n; ++k, x >>= 2) out[i + k] = kBase[x & 3];
    }
}

int pba_synth_reads_range(uint64_t seed, const char *genome, size_t L, uint32_t r_lo, uint32_t r_hi, uint32_t read_len, double p_ins,
                          double p_del, double p_sub, char *out, uint32_t *starts, int nthreads) {
    if (!genome || !out || L == 0 || read_len == 0 || r_lo > r_hi) return PBA_E_INVALID;
    if (p_ins < 0 || p_del < 0 || p_sub < 0 || p_ins + p_del + p_sub >= 1.0) return PBA_E_INVALID;
    const double two32 = 4294967296.0;
    const uint64_t t_ins = (uint64_t)(p_ins * two32);
    const uint64_t t_del = t_ins + (uint64_t)(p_del * two32);
    const uint64_t t_sub = t_del + (uint64_t)(p_sub * two32);
    const uint32_t n_reads = r_hi - r_lo;
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > n_reads) nthreads = n_reads ? (int)n_reads : 1;
    std::vector<std::thread> th;
    const uint32_t per = (n_reads + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        const uint32_t r0 = r_lo + (uint32_t)t * per, r1 = r0 + per < r_hi ? r0 + per : r_hi;
        if (r0 >= r1) break;
        th.emplace_back(gen_reads, seed, genome, L, r0, r1, r_lo, read_len, t_ins, t_del, t_sub, out, starts);
    }
    for (auto &x : th) x.join();
    return PBA_OK;
}

int pba_synth_reads(uint64_t seed, const char *genome, size_t L, uint32_t n_reads, uint32_t read_len, double p_ins,
                    double p_del, double p_sub, char *out, uint32_t *starts, int nthreads) {
    return pba_synth_reads_range(seed, genome, L, 0, n_reads, read_len, p_ins, p_del, p_sub, out, starts, nthreads);
}

}  // extern "C"
