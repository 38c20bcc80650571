// seq_aligner.h -- the reference's seq_aligner<MAXN,MAXM> API (/root/reference/src/seq_aligner.h) with the
// banded DP and its traceback running on the MI355X (pba_align_text / pba_align_text_trace: raw-byte semantics, one
// wavefront per pair -- ACGT-only pairs on the bit-vector array, anything else on the reference-shaped row sweep).
//
// The DP matrix never leaves the GPU unless somebody reads it: get_cost / get_parent of a cell other than the two every
// caller wants (the goal cell, and the end of the diagonal that locator.cpp:86 prints), set_cost / set_parent and `mat`
// fetch the band once (pba_align_text_matrix re-runs the pair that align() last saw, whose elements are kept here) and
// then work on that host copy like the reference works on its array.  Cells the last align() did not write read as cost
// 65535, parent 0 (the reference hands back whatever an earlier call left in its 1.25 GB array -- SURVEY B4/B8).
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
    typedef struct { int cost; int parent; } state;                // seq_aligner.h:60-63
    seq_aligner() : R(MAXR), len_a(0), len_b(0), max_dst(0), matlen_a(0), matlen_b(0), edits(this), nedit(this), mat(this),
                    cost_(0), diag_(-1), ops_(MAXN + MAXM), edits_(MAXN + MAXM), nedit_(0), have_edits_(true), eager_(false),
                    a_fwd_(true), b_fwd_(true), slen_a_(0), slen_b_(0), have_mat_(false) {}
    seq_aligner(double r) : R(r), len_a(0), len_b(0), max_dst(0), matlen_a(0), matlen_b(0), edits(this), nedit(this), mat(this),
                            cost_(0), diag_(-1), ops_(MAXN + MAXM), edits_(MAXN + MAXM), nedit_(0), have_edits_(true), eager_(false),
                            a_fwd_(true), b_fwd_(true), slen_a_(0), slen_b_(0), have_mat_(false) {}
    double R;                    // seq_aligner.h:73-80, same names
    int len_a, len_b, max_dst, matlen_a, matlen_b;

    // `edit edits[MAXN+MAXM]; int nedit;` (seq_aligner.h:79-80).  The reference walks the path back after every successful
    // align(); most callers never look (locator.cpp; spaced_seed.cpp with a locked reference), and on the GPU the traced
    // sweep costs 2.6 x the score-only one (a 15 kb pair on its one wavefront: 5.5 ms against 2.1), so here the script is
    // produced when it is first read: both members are thin views that run the traced pass of the pair align() last saw,
    // once.  They read like the reference's (`al.nedit` in int context, `al.edits[k].op / .val`, `al.edits` as an `edit *`);
    // a caller that knows it will read them (the compat ref_seq with an unlocked reference) asks align() for them up
    // front with want_edits(true) and pays one sweep instead of two.
    struct edits_view {
        seq_aligner *al;
        explicit edits_view(seq_aligner *a) : al(a) {}
        edit &operator[](int k) { al->need_edits(); return al->edits_[k]; }
        operator edit *() { al->need_edits(); return al->edits_.data(); }
    } edits;
    struct nedit_view {
        seq_aligner *al;
        explicit nedit_view(seq_aligner *a) : al(a) {}
        operator int() const { al->need_edits(); return al->nedit_; }
        nedit_view &operator=(int v) { al->need_edits(); al->nedit_ = v; return *this; }
    } nedit;
    void want_edits(bool eager) { eager_ = eager; }                // compat extension (not in the reference)

    // `state mat[MAXN][MAXM]` (seq_aligner.h:81): mat[i][c] is cell (i, c + i - max_dst) of the last alignment
    struct mat_row {
        seq_aligner *al; int i;
        state &operator[](int c) { return al->cell(i, c); }
    };
    struct mat_view {
        seq_aligner *al;
        explicit mat_view(seq_aligner *a) : al(a) {}
        mat_row operator[](int i) { mat_row r = {al, i}; return r; }
    } mat;

    // seq_aligner.h:92: -1 on failure, matlen_b on success; results valid until the next call
    int align(seq_accessor *seg_a, seq_accessor *seg_b) {
        pba_result r;
        int32_t ne = 0;
        have_mat_ = false;
        a_fwd_ = seg_a->is_forward(); b_fwd_ = seg_b->is_forward();
        // (the engine clips the accessors like seq_aligner.h:94-102 before it sizes, checks or copies anything: the whole
        // rest of a contig or of the reference may hang off either of them, locator.cpp:80-81, ref_seq.h:282-286)
        int st = eager_ ? pba_align_text_trace(pba_compat_ctx(), seg_a->pt(0), a_fwd_, seg_a->length(), seg_b->pt(0), b_fwd_,
                                               seg_b->length(), R, MAXN, MAXM, &r, ops_.data(), (int32_t)ops_.size(), &ne)
                        : pba_align_text(pba_compat_ctx(), seg_a->pt(0), a_fwd_, seg_a->length(), seg_b->pt(0), b_fwd_,
                                         seg_b->length(), R, MAXN, MAXM, &r);
        if (st != PBA_OK) {
            LOG("pba_align_text: %s\n", pba_ctx_error(pba_compat_ctx()));
            return -1;
        }
        len_a = r.len_a; len_b = r.len_b; max_dst = r.max_dst;
        if (len_a >= (MAXN + MAXM) || max_dst >= MAXM) LOG("segment too long: %d\n", len_a);   // seq_aligner.h:104-107
        matlen_a = r.matlen_a; matlen_b = r.matlen_b; cost_ = r.cost; diag_ = r.diag_cost;
        // the elements the sweep saw, for a later look at the script or the matrix (the accessors point into the caller's
        // buffers); kept in accessor order, so the later calls read them forward
        ea_.resize(len_a); eb_.resize(len_b);
        for (int k = 0; k < len_a; ++k) ea_[k] = seg_a->at(k);
        for (int k = 0; k < len_b; ++k) eb_[k] = seg_b->at(k);
        if (r.rc >= 0) {                                            // seq_aligner.h:115-116, 214-233 (a failed align leaves the last script)
            have_edits_ = eager_;
            if (eager_) fill_edits(ne, eb_);
            else { sa_ = ea_; sb_ = eb_; slen_a_ = len_a; slen_b_ = len_b; }   // the pair whose script is owed
        }
        // the reference leaves the accessors' cursors where its sweep stopped; callers re-reset them before reuse
        return r.rc;
    }
    int final_cost() { return get_cost(matlen_a, matlen_b); }       // seq_aligner.h:130
    int get_cost(int i, int j) {                                    // seq_aligner.h:131
        if (!have_mat_) {
            if (i == matlen_a && j == matlen_b) return cost_;
            if (i == j && i == std::min(len_a, len_b) && diag_ >= 0) return diag_;   // locator.cpp:86
        }
        return cell(i, j - i + max_dst).cost;
    }
    void set_cost(int i, int j, int v) { cell(i, j - i + max_dst).cost = v; }        // seq_aligner.h:132
    int get_parent(int i, int j) { return cell(i, j - i + max_dst).parent; }         // seq_aligner.h:133
    void set_parent(int i, int j, int p) { cell(i, j - i + max_dst).parent = p; }    // seq_aligner.h:134
private:
    // edits[k] from ops_[k]: MATCH / INSERT carry the b element they consume (seq_aligner.h:218,224)
    void fill_edits(int ne, const std::vector<char> &b) {
        nedit_ = ne;
        for (int k = 0, j = 0; k < ne && k < (int)edits_.size(); ++k) {
            edits_[k].op = (OP)ops_[k];
            if (ops_[k] != DELETE && j < (int)b.size()) edits_[k].val = b[j++];
        }
    }
    // the script of the pair the last SUCCESSFUL align() saw (a failed one in between leaves it, like the reference), on first use
    void need_edits() {
        if (have_edits_) return;
        have_edits_ = true;
        pba_result r;
        int32_t ne = 0;
        int st = pba_align_text_trace(pba_compat_ctx(), sa_.data(), 1, slen_a_, sb_.data(), 1, slen_b_, R, MAXN, MAXM, &r, ops_.data(),
                                      (int32_t)ops_.size(), &ne);
        if (st != PBA_OK) { LOG("pba_align_text_trace: %s\n", pba_ctx_error(pba_compat_ctx())); exit(1); }
        fill_edits(r.rc >= 0 ? ne : 0, sb_);
    }
    // cell (i, c) of the stripe, fetched from the GPU on first use after an align()
    state &cell(int i, int c) {
        const int W = 2 * max_dst + 1;
        if (!have_mat_) {
            const size_t cells = (size_t)(len_a + 1) * W;
            std::vector<uint16_t> cost(cells);
            std::vector<uint8_t> par(cells);
            pba_result r;
            int32_t rows = 0;
            int st = pba_align_text_matrix(pba_compat_ctx(), ea_.data(), 1, len_a, eb_.data(), 1, len_b, R, MAXN, MAXM, &r, cost.data(),
                                           par.data(), cells, &rows);
            if (st != PBA_OK) { LOG("pba_align_text_matrix: %s\n", pba_ctx_error(pba_compat_ctx())); exit(1); }
            host_mat_.resize(cells);
            for (size_t k = 0; k < cells; ++k) { host_mat_[k].cost = cost[k]; host_mat_[k].parent = par[k]; }
            have_mat_ = true;
        }
        if (i < 0 || i > len_a || c < 0 || c >= W) { scratch_.cost = 65535; scratch_.parent = 0; return scratch_; }   // outside what align() touched
        return host_mat_[(size_t)i * W + c];
    }
    int cost_, diag_;
    std::vector<uint8_t> ops_;
    std::vector<edit> edits_;
    int nedit_;
    bool have_edits_, eager_, a_fwd_, b_fwd_;
    std::vector<char> ea_, eb_;      // elements of the last call (the matrix on request)
    std::vector<char> sa_, sb_;      // ... of the last successful call whose script has not been asked for yet
    int slen_a_, slen_b_;
    std::vector<state> host_mat_;
    state scratch_;
    bool have_mat_;
};

typedef seq_aligner<MAX_READ_LEN + MAX_DIFF_LEN, MAX_DIFF_LEN> t_aligner;   // seq_aligner.h:260

#endif
