// ref_seq.h -- the reference's ref_seq (/root/reference/src/ref_seq.h) over the C ABI: the text buffer,
// get_accessor, get_seedmap (seed index built on the GPU), try_align, and -- for an unlocked reference -- the
// voting half: the vote boxes live in HBM (pba_cons_*), elect / append / prepend / evolve run there.
#ifndef PBA_COMPAT_REF_SEQ_H
#define PBA_COMPAT_REF_SEQ_H

#include <string.h>
#include <vector>

#include "common.h"
#include "dna_seq.h"
#include "seq_aligner.h"

class ref_seq {
public:
    // (the reference's lk defaults to false; every caller in the reference passes it explicitly)
    ref_seq(const t_bseq *pseq, bool lk = false) : locked(lk), txt_buf(3 * MAX_SEQ_LEN), cons(NULL) {   // ref_seq.h:207
        beg = pre = MAX_SEQ_LEN;
        end = post = beg + (int)dna_seq::bin2text(pseq, &txt_buf[beg], MAX_SEQ_LEN);
        make_boxes(1);
    }
    ref_seq(const char *ptxt, int len, bool l, int w = 1) : locked(l), txt_buf(3 * MAX_SEQ_LEN), cons(NULL) {   // ref_seq.h:218
        beg = pre = MAX_SEQ_LEN;
        end = post = beg + len;
        memcpy(&txt_buf[beg], ptxt, len);
        make_boxes(w);
    }
    ~ref_seq() { pba_cons_destroy(cons); }
    void append(char *pseg, int len) {                                                           // ref_seq.h:227
        memmove(&txt_buf[post], pseg, len);
        post += len;
        if (cons) check(pba_cons_append(pba_compat_ctx(), cons, pseg, len));
    }
    void prepend(char *pseg, int len) {                                                          // ref_seq.h:235
        pre -= len;
        memmove(&txt_buf[pre], pseg, len);
        if (cons) check(pba_cons_prepend(pba_compat_ctx(), cons, pseg, len));
    }
    bool contained(int pos) { return pos + beg >= pre && pos + beg < post; }                     // ref_seq.h:248
    unsigned length() { return end - beg; }                                                      // ref_seq.h:253

    seq_accessor get_accessor(int pos, bool forward) {                                           // ref_seq.h:282
        assert(contained(pos));
        return seq_accessor(&txt_buf[beg + pos], forward, forward ? post - beg - pos : pos + beg - pre + 1);
    }

    // ref_seq.h:259-276
    bool try_align(t_aligner *paligner, int pos, seq_accessor *pac_seg) {
        bool forward = pac_seg->is_forward();
        seq_accessor ac_ref = get_accessor(pos, forward);
        paligner->want_edits(!locked);                                // an unlocked reference reads the script of every success (elect)
        if (paligner->align(&ac_ref, pac_seg) < 0) return false;      // the reference is `a`, the read is `b`
        if (paligner->matlen_a < OVERLAP_MIN) return false;
        if (locked) return true;
        elect(pos, paligner->edits, paligner->nedit, forward);
        if (paligner->matlen_a == ac_ref.length()) {
            int add_len = pac_seg->length() - paligner->matlen_b;
            if (forward) append(pac_seg->pt(paligner->matlen_b), add_len);
            else prepend(pac_seg->pt(pac_seg->length() - 1), add_len);
        }
        return true;
    }

    // ref_seq.h:352-362 (+ apply_edits :25-41): the script's votes, added to the boxes in HBM
    void elect(int pos, edit *pedit, int nedit, bool forward) {
        if (!cons || nedit <= 0) return;
        std::vector<uint8_t> ops(nedit);
        std::vector<char> vals(nedit);
        for (int k = 0; k < nedit; ++k) { ops[k] = (uint8_t)pedit[k].op; vals[k] = pedit[k].val; }
        const int32_t p = pos, ne = nedit;
        const uint8_t f = forward ? 1 : 0;
        const uint64_t off[2] = {0, (uint64_t)nedit};
        check(pba_cons_elect(pba_compat_ctx(), cons, 1, &p, &f, ops.data(), vals.data(), off, &ne));
    }

    // ref_seq.h:317-349: the votes become the next reference; the previous seedmap is stale afterwards
    void evolve() {
        if (locked) return;
        int32_t n = 0;
        beg = pre = end = MAX_SEQ_LEN;
        check(pba_cons_evolve(pba_compat_ctx(), cons, &txt_buf[beg], MAX_SEQ_LEN * 2, &n));
        end = post = beg + n;
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
    void check(int st) {
        if (st != PBA_OK) { LOG("ref_seq: %s\n", pba_ctx_error(pba_compat_ctx())); exit(1); }   // like handle_error, spaced_seed.cpp:40
    }
    void make_boxes(int w) {        // the constructor's push_back(vote_box(c, w)) loop, ref_seq.h:211-213,222-224
        if (!locked) check(pba_cons_create(pba_compat_ctx(), &txt_buf[beg], end - beg, w, MAX_SEQ_LEN, &cons));
    }
    int beg, end, pre, post;
    bool locked;
    std::vector<char> txt_buf;
    pba_cons *cons;                 // vote boxes in HBM; NULL for a locked reference
    ref_seq(const ref_seq &);       // not copyable (owns device memory)
    ref_seq &operator=(const ref_seq &);
};

#endif
