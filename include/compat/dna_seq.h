// dna_seq.h -- the reference's dna_seq / seq_accessor API (/root/reference/src/dna_seq.h) on top of libpba.
#ifndef PBA_COMPAT_DNA_SEQ_H
#define PBA_COMPAT_DNA_SEQ_H

#include <assert.h>
#include <string.h>

#include "common.h"

#define C2I(x) ((x) == 'A' ? 0 : ((x) == 'C' ? 1 : ((x) == 'G' ? 2 : 3)))   // dna_seq.h:21
#define I2C(x) ((x) == 0 ? 'A' : ((x) == 1 ? 'C' : ((x) == 2 ? 'G' : 'T')))  // dna_seq.h:23
#define N_SEQ_WORD 16
#define N_SEQ_BYTE 4

class dna_seq {
public:
    bool parse(const char *) { return false; }                       // dna_seq.h:53-57: stubs in the reference too
    bool empty() { return true; }
    const unsigned *read(unsigned *) const { return NULL; }

    static t_seed seed_at(unsigned char *pbin, int pos) { return pba_seed_at(pbin, pos); }          // dna_seq.h:62 (bug-compatible)
    static char value_at(unsigned char bv, int idx) { return pba_value_at(bv, idx); }               // dna_seq.h:78
    static unsigned encode(const char *ptext) { return pba_encode16(ptext); }                       // dna_seq.h:86
    static void decode(unsigned code, char *ptext) { pba_decode16(code, ptext); }                   // dna_seq.h:101
    static unsigned text2bin(const char *ptext, unsigned char *pbin, unsigned buflen) {             // dna_seq.h:113
        size_t n = pba_text2bin(ptext, strlen(ptext), pbin, buflen);
        assert(n != 0);
        return (unsigned)n;
    }
    static unsigned bin2text(const unsigned char *pbin, char *ptext, unsigned buflen) {             // dna_seq.h:133
        unsigned tlen;
        memcpy(&tlen, pbin, 4);
        assert(buflen > tlen);
        return (unsigned)pba_bin2text(pbin, ptext, buflen);
    }
};

// dna_seq.h:185-233: forward/backward cursor over a caller-owned text buffer
class seq_accessor {
public:
    seq_accessor(char *p, bool f, int l) : pdna(p), pcur(p), len(l), cnt(0), forward(f) {}
    int length() { return len; }
    bool is_forward() { return forward; }
    bool has_more() { return cnt < len; }
    char next() { ++cnt; return forward ? *pcur++ : *pcur--; }
    void reset(int pos) { cnt = pos; pcur = forward ? pdna + pos : pdna - pos; }
    char at(int i) { return forward ? *(pdna + i) : *(pdna - i); }
    char *pt(int i) { return forward ? (pdna + i) : (pdna - i); }
private:
    char *pdna, *pcur;
    int len, cnt;
    bool forward;
};

#endif
