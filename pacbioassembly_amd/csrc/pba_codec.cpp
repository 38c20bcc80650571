// pba_codec.cpp -- host side of the 2-bit codec, bit-compatible with the reference's
// dna_seq statics (/root/reference/src/dna_seq.h:21-176).  Pure functions, no GPU.
#include <string.h>

#include "pba.h"
#include "pba_internal.h"

namespace {

// dna_seq.h:147-159: up to four bases into one byte, first base in bits 7:6, short tail zero-filled
inline uint8_t pack_quad(const char *p, size_t n) {
    unsigned b = 0;
    for (size_t k = 0; k < 4 && k < n; ++k) b |= pba_c2i((unsigned char)p[k]) << (6 - 2 * k);
    return (uint8_t)b;
}

const char kBase[4] = {'A', 'C', 'G', 'T'};  // dna_seq.h:30

}  // namespace

extern "C" {

uint32_t pba_encode16(const char *t) {
    // dna_seq.h:86-96: the four packed bytes in memory order form a little-endian word
    return (uint32_t)pack_quad(t, 4) | (uint32_t)pack_quad(t + 4, 4) << 8 | (uint32_t)pack_quad(t + 8, 4) << 16 |
           (uint32_t)pack_quad(t + 12, 4) << 24;
}

void pba_decode16(uint32_t code, char *t) {
    for (int k = 0; k < 16; ++k) t[k] = kBase[(code >> (8 * (k >> 2) + 6 - 2 * (k & 3))) & 3];
}

size_t pba_text2bin(const char *text, size_t tlen, uint8_t *rec, size_t cap) {
    const size_t need = 4 + (tlen + 3) / 4;
    if (!text || !rec || cap < need || tlen > 0xFFFFFFFFull) return 0;
    const uint32_t l32 = (uint32_t)tlen;
    memcpy(rec, &l32, 4);  // host-endian, unaligned (dna_seq.h:119)
    for (size_t i = 0, o = 4; i < tlen; i += 4, ++o) rec[o] = pack_quad(text + i, tlen - i);
    return need;
}

size_t pba_bin2text(const uint8_t *rec, char *text, size_t cap) {
    if (!rec || !text) return 0;
    uint32_t tlen;
    memcpy(&tlen, rec, 4);
    if (cap <= tlen) return 0;  // dna_seq.h:138 asserts buflen > tlen
    const uint8_t *pb = rec + 4;
    for (uint32_t i = 0; i < tlen; ++i) text[i] = kBase[(pb[i >> 2] >> (6 - 2 * (i & 3))) & 3];
    text[tlen] = '\0';
    return tlen;
}

uint32_t pba_seed_at_fixed(const uint8_t *rec, int pos) {
    const uint8_t *p = rec + 4 + (pos >> 2);
    const unsigned ls = (unsigned)(pos & 3) * 2;
    uint32_t w = 0;
    for (int k = 0; k < 4; ++k) {
        const unsigned b = ls ? (((unsigned)p[k] << ls) | ((unsigned)p[k + 1] >> (8 - ls))) & 0xFFu : p[k];
        w |= b << (8 * k);
    }
    return w;
}

uint32_t pba_seed_at(const uint8_t *rec, int pos) {
    if ((pos & 3) == 0) {
        // dna_seq.h:64: `pos` is added as a byte offset, i.e. this is the window of base 4*pos
        uint32_t w;
        memcpy(&w, rec + 4 + pos, 4);
        return w;
    }
    return pba_seed_at_fixed(rec, pos);
}

uint32_t pba_mask_from_pattern(const char *pat) {
    char w[16];
    const size_t n = pat ? strnlen(pat, 16) : 0;
    for (size_t k = 0; k < 16; ++k) w[k] = (k < n && pat[k] == '1') ? 'T' : 'A';
    return pba_encode16(w);
}

char pba_value_at(uint8_t bv, int idx) { return kBase[(bv >> ((~idx & 3) << 1)) & 3]; }

size_t pba_open_binary(const uint8_t *file, size_t len, uint32_t min_excl, uint32_t max_excl, uint64_t *offs,
                       size_t cap, size_t *n_total) {
    size_t kept = 0, total = 0;
    for (size_t off = 0; file && off + 4 <= len;) {
        uint32_t sl;
        memcpy(&sl, file + off, 4);
        if (sl > min_excl && sl < max_excl) {  // spaced_seed.cpp:334
            if (offs && kept < cap) offs[kept] = off;
            ++kept;
        }
        ++total;
        off += 4 + ((size_t)sl + 3) / 4;  // spaced_seed.cpp:341
    }
    if (n_total) *n_total = total;
    return kept;
}

const char *pba_strerror(int st) {
    switch (st) {
        case PBA_OK: return "ok";
        case PBA_E_INVALID: return "invalid argument";
        case PBA_E_NOMEM: return "out of memory";
        case PBA_E_HIP: return "HIP runtime or kernel failure";
        case PBA_E_TOOLONG: return "sequence too long for the engine";
        case PBA_E_NODEVICE: return "no usable gfx950 device";
        case PBA_E_ALPHABET: return "byte outside ACGT in a strict sequence set";
        default: return "unknown status";
    }
}

}  // extern "C"
