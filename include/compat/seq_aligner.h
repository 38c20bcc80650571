// seq_aligner.h -- the reference's seq_aligner<MAXN,MAXM> API (/root/reference/src/seq_aligner.h) with the
// banded DP and its traceback running on the MI355X (pba_align_text_trace: raw-byte comparison, one wavefront per pair).
#ifndef PBA_COMPAT_SEQ_ALIGNER_H
#define PBA_COMPAT_SEQ_ALIGNER_H

#include <vector>

#include "common.h"
#include "dna_seq.h"

enum OP { MATCH = 1, INSERT, DELETE };                              // seq_aligner.h:32-36
typedef struct { enum OP op; char val; } edit;                      // seq_aligner.h:41-44

template <int MAXN, int MAXM>
class seq_aligner {
public:
    seq_aligner() : R(MAXR), len_a(0), len_b(0), max_dst(0), matlen_a(0), matlen_b(0), edits(MAXN + MAXM), nedit(0),
                    cost_(0), ops_(MAXN + MAXM) {}
    seq_aligner(double r) : R(r), len_a(0), len_b(0), max_dst(0), matlen_a(0), matlen_b(0), edits(MAXN + MAXM), nedit(0),
                            cost_(0), ops_(MAXN + MAXM) {}
    double R;                    // seq_aligner.h:73-80, same names
    int len_a, len_b, max_dst, matlen_a, matlen_b;
    std::vector<edit> edits;     // edits[k].op / .val as in the reference (an array there, indexable the same way)
    int nedit;

    // seq_aligner.h:92: -1 on failure, matlen_b on success; results valid until the next call
    int align(seq_accessor *seg_a, seq_accessor *seg_b) {
        pba_result r;
        int32_t ne = 0;
        int st = pba_align_text_trace(pba_compat_ctx(), seg_a->pt(0), seg_a->is_forward(), seg_a->length(), seg_b->pt(0),
                                      seg_b->is_forward(), seg_b->length(), R, MAXN, MAXM, &r, ops_.data(),
                                      (int32_t)ops_.size(), &ne);
        if (st != PBA_OK) {
            LOG("pba_align_text_trace: %s\n", pba_ctx_error(pba_compat_ctx()));
            return -1;
        }
        len_a = r.len_a; len_b = r.len_b; max_dst = r.max_dst;
        if (len_a >= (MAXN + MAXM) || max_dst >= MAXM) LOG("segment too long: %d\n", len_a);   // seq_aligner.h:104-107
        matlen_a = r.matlen_a; matlen_b = r.matlen_b; cost_ = r.cost;
        if (r.rc >= 0) {                                            // seq_aligner.h:115-116, 214-233
            nedit = ne;
            for (int k = 0, j = 0; k < ne && k < (int)edits.size(); ++k) {
                edits[k].op = (OP)ops_[k];
                if (ops_[k] != DELETE) edits[k].val = seg_b->at(j++);   // MATCH / INSERT carry the b element they consume
            }
        }
        // the reference leaves the accessors' cursors where its sweep stopped; callers re-reset them before reuse
        return r.rc;
    }
    int final_cost() { return cost_; }                              // seq_aligner.h:130
    // only the goal cell is kept (the DP matrix never exists on the GPU); other cells: -1
    int get_cost(int i, int j) { return (i == matlen_a && j == matlen_b) ? cost_ : -1; }
private:
    int cost_;
    std::vector<uint8_t> ops_;
};

typedef seq_aligner<MAX_READ_LEN + MAX_DIFF_LEN, MAX_DIFF_LEN> t_aligner;   // seq_aligner.h:260

#endif
