// ref_seq.h -- the hot-path half of the reference's ref_seq (/root/reference/src/ref_seq.h): the text buffer,
// get_accessor, get_seedmap (seed index built on the GPU) and the locked-mode try_align.  The voting /
// consensus half (base_vote, vote_box, elect, evolve, apply_edits) is out of scope.
#ifndef PBA_COMPAT_REF_SEQ_H
#define PBA_COMPAT_REF_SEQ_H

#include <string.h>
#include <vector>

#include "common.h"
#include "dna_seq.h"
#include "seq_aligner.h"

class ref_seq {
public:
    ref_seq(const t_bseq *pseq, bool lk = true) : locked(lk), txt_buf(3 * MAX_SEQ_LEN) {        // ref_seq.h:207
        beg = pre = MAX_SEQ_LEN;
        end = post = beg + (int)dna_seq::bin2text(pseq, &txt_buf[beg], MAX_SEQ_LEN);
    }
    ref_seq(const char *ptxt, int len, bool l = true, int /*w*/ = 1) : locked(l), txt_buf(3 * MAX_SEQ_LEN) {   // ref_seq.h:218
        beg = pre = MAX_SEQ_LEN;
        end = post = beg + len;
        memcpy(&txt_buf[beg], ptxt, len);
    }
    void append(char *pseg, int len) { memmove(&txt_buf[post], pseg, len); post += len; }        // ref_seq.h:227
    void prepend(char *pseg, int len) { pre -= len; memmove(&txt_buf[pre], pseg, len); }         // ref_seq.h:235
    bool contained(int pos) { return pos + beg >= pre && pos + beg < post; }                     // ref_seq.h:248
    unsigned length() { return end - beg; }                                                      // ref_seq.h:253

    seq_accessor get_accessor(int pos, bool forward) {                                           // ref_seq.h:282
        assert(contained(pos));
        return seq_accessor(&txt_buf[beg + pos], forward, forward ? post - beg - pos : pos + beg - pre + 1);
    }

    // ref_seq.h:259-265.  Only the locked behaviour exists here (no vote, no growth).
    bool try_align(t_aligner *paligner, int pos, seq_accessor *pac_seg) {
        bool forward = pac_seg->is_forward();
        seq_accessor ac_ref = get_accessor(pos, forward);
        if (paligner->align(&ac_ref, pac_seg) < 0) return false;      // the reference is `a`, the read is `b`
        if (paligner->matlen_a < OVERLAP_MIN) return false;
        return true;
    }

    // ref_seq.h:291-311: head ascending then tail descending, key 0 dropped; the windows are hashed and ordered
    // on the GPU (PBA_INDEX_HEAD_TAIL) and copied into the caller's table in the reference's list order
    unsigned get_seedmap(hash_table &seedmap, t_seed sd_pat) {
        seedmap.clear();
        pba_ctx *ctx = pba_compat_ctx();
        const uint64_t offs[2] = {0, (uint64_t)(end - beg)};
        pba_seqs *s = NULL;
        pba_index *ix = NULL;
        if (pba_seqs_from_text(ctx, &txt_buf[beg], offs, 1, 0, &s) != PBA_OK ||
            pba_index_build(ctx, s, 0, sd_pat, PBA_INDEX_HEAD_TAIL, &ix) != PBA_OK) {
            LOG("get_seedmap: %s\n", pba_ctx_error(ctx));
            exit(1);
        }
        uint64_t n = pba_index_entries(ix);
        std::vector<uint32_t> keys(n + 1);
        std::vector<int32_t> pos(n + 1);
        pba_index_dump(ctx, ix, keys.data(), pos.data(), n, &n);
        for (uint64_t i = 0; i < n; ++i) seedmap[keys[i]].push_back(pos[i]);
        unsigned rv = pba_index_visited(ix);
        pba_index_destroy(ix);
        pba_seqs_destroy(s);
        return rv;
    }
private:
    int beg, end, pre, post;
    bool locked;
    std::vector<char> txt_buf;
};

#endif
