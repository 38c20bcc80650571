// compat_test.cpp -- the reference's own unit-test expectations (test/dna_test.cpp, test/aligner_test.cpp,
// test/ref_test.cpp: all twelve cases), restated against include/compat/*.h, i.e. running on the MI355X through
// libpba.so (OVERLAP_MIN = 16 as SURVEY B9 records: at the reference's 64 its own 43-base cases cannot pass).
// Own code: only the known-answer values are the reference's.  Exit code 0 = all passed.
#define OVERLAP_MIN 16
#define PBA_COMPAT_QUIET
#include <fstream>
#include <string>

#include "dna_seq.h"
#include "ref_seq.h"
#include "seq_aligner.h"

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                                   \
    do {                                                                              \
        ++g_checks;                                                                   \
        if (!(cond)) { ++g_fail; fprintf(stderr, "FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

static void dna_binary() {             // test/dna_test.cpp:18-29
    char dna_str[] = "ACGTGTCATCGGATCAACCGGTT";
    unsigned char bin_buf[64] = {0};
    char txt_buf[41];
    CHECK(10 == dna_seq::text2bin(dna_str, bin_buf, 14));
    CHECK(23 == dna_seq::bin2text(bin_buf, txt_buf, 41));
    CHECK(std::string(dna_str) == txt_buf);
    CHECK(0x34DAB41B == dna_seq::seed_at(bin_buf, 0));
    CHECK(0xD068D36E == dna_seq::seed_at(bin_buf, 1));
    CHECK(0x41A34DBB == dna_seq::seed_at(bin_buf, 2));
    CHECK(0xAF058D36 == dna_seq::seed_at(bin_buf, 7));
}

static void accessor() {               // test/dna_test.cpp:32-60
    char dna_str[] = "ACGTGTCATCGGATCAACCGGTT";
    seq_accessor da(dna_str, true, 4);
    CHECK(4 == da.length());
    CHECK('A' == da.next()); CHECK('C' == da.next()); CHECK('G' == da.next()); CHECK('T' == da.next());
    CHECK('G' == da.at(2)); CHECK(!da.has_more());
    da.reset(2);
    CHECK(da.has_more()); CHECK('G' == da.next()); CHECK('T' == da.next()); CHECK(!da.has_more());
    seq_accessor db(dna_str + 4, false, 3);
    CHECK(3 == db.length());
    CHECK('G' == db.next()); CHECK('T' == db.next()); CHECK(db.has_more()); CHECK('G' == db.next());
    CHECK(!db.has_more()); CHECK('T' == db.at(1));
    db.reset(1);
    CHECK('T' == db.next()); CHECK('G' == db.next()); CHECK(!db.has_more());
}

// test/aligner_test.cpp:29-41: replaying edits[] against the reference string must reproduce it
template <class AL>
static void edit_tester(seq_accessor *pref, AL *pal) {
    int j = 0;
    for (int i = 0; i < pal->nedit; ++i) {
        char op = pal->edits[i].op;
        if (op == MATCH || op == INSERT) { CHECK(pref->at(j) == pal->edits[i].val); ++j; }
    }
}

static void aligner(const char *real_align_path) {   // test/aligner_test.cpp:44-117 (score expectations)
    char dna_ref[] = "ACGTAACCGGTT", dna_seg1[] = "CGTAAGC", dna_seg2[] = "GTAACGGGTTAA", dna_seg3[] = "TCGTAAC";
    t_aligner *pal = new t_aligner();
    {
        seq_accessor ref1(dna_ref, true, 7), seg1(dna_seg1, true, 6);
        int rc = pal->align(&seg1, &ref1);
        CHECK(6 <= rc && rc <= 7); CHECK(2 == pal->final_cost()); edit_tester(&ref1, pal);
        seq_accessor ref2(dna_ref, true, 8), seg2(dna_seg1, true, 7);
        CHECK(7 == pal->align(&seg2, &ref2)); CHECK(2 == pal->final_cost()); edit_tester(&ref2, pal);
        seq_accessor ref3(dna_ref, true, 8), seg3(dna_seg3, true, 7);
        CHECK(7 == pal->align(&seg3, &ref3)); CHECK(1 == pal->final_cost()); edit_tester(&ref3, pal);
    }
    {   // backward
        seq_accessor ref(dna_ref + 7, false, 7), seg(dna_seg1 + 6, false, 7);
        CHECK(7 == pal->align(&seg, &ref)); CHECK(1 == pal->final_cost()); edit_tester(&ref, pal);
    }
    {   // overlay
        seq_accessor ref(dna_ref + 2, true, 10), seg(dna_seg2, true, 12);
        CHECK(10 == pal->align(&seg, &ref)); CHECK(1 == pal->final_cost()); edit_tester(&ref, pal);
    }
    {   // remove: test/aligner_test.cpp:82-98
        seq_accessor ref(dna_ref, true, 10), seg(dna_ref + 1, true, 9);
        CHECK(10 == pal->align(&seg, &ref)); CHECK(10 == pal->nedit); CHECK(INSERT == pal->edits[0].op);
        CHECK(1 == pal->final_cost()); edit_tester(&ref, pal);
        ref.reset(0); seg.reset(0);
        CHECK(9 == pal->align(&ref, &seg)); CHECK(10 == pal->nedit); CHECK(DELETE == pal->edits[0].op);
        CHECK(1 == pal->final_cost()); edit_tester(&seg, pal);
    }
    {   // sample: real reads
        std::ifstream fin(real_align_path);
        std::string ref_str, seg_str;
        fin >> ref_str >> seg_str;
        CHECK(ref_str.size() == 736);
        seq_accessor ref((char *)ref_str.c_str() + ref_str.length() - 1, false, ref_str.length());
        seq_accessor seg((char *)seg_str.c_str() + seg_str.length() - 1, false, seg_str.length());
        CHECK(0 < pal->align(&seg, &ref)); edit_tester(&ref, pal);
        CHECK(736 == pal->matlen_a && 736 == pal->matlen_b && 7 == pal->final_cost() && 739 == pal->nedit);
        fin >> ref_str >> seg_str;
        seq_accessor ref2((char *)ref_str.c_str(), true, ref_str.length());
        seq_accessor seg2((char *)seg_str.c_str(), true, seg_str.length());
        CHECK(-1 == pal->align(&seg2, &ref2));
        // edits[] / nedit are produced on first use here: a failed align() in between leaves the last success's script, like
        // the reference's array (seq_aligner.h:115 resets nedit only on the way to find_path) -- even one nobody had read yet
        seq_accessor ref3((char *)"ACGTAACCGGTT", true, 10), seg3((char *)"CGTAACCGG", true, 9);
        CHECK(10 == pal->align(&seg3, &ref3));                  // success, script not looked at
        seg2.reset(0); ref2.reset(0);
        CHECK(-1 == pal->align(&seg2, &ref2));                  // failure
        CHECK(10 == pal->nedit && INSERT == pal->edits[0].op);  // the success's script
        edit *raw = pal->edits;                                 // the array as a pointer (ref_seq.h:267 passes it on like this)
        CHECK(raw[1].op == MATCH && raw[1].val == 'C');
    }
    {   // the matrix behind the alignment: get_cost / get_parent / set_cost / set_parent / mat (seq_aligner.h:81,131-134)
        // a = "ACGTAACC" vs b = "ACGAACCGG": D(i,j) by hand; max_dst = 1 + (int)(8 * 0.3) = 3
        char a[] = "ACGTAACC", b[] = "ACGAACCGG";
        seq_accessor sa(a, true, 8), sb(b, true, 9);
        int rc = pal->align(&sa, &sb);
        CHECK(rc > 0 && pal->len_a == 8 && pal->len_b == 9 && pal->max_dst == 3);
        CHECK(pal->final_cost() == pal->get_cost(pal->matlen_a, pal->matlen_b));
        CHECK(pal->get_cost(8, 8) == 2);                 // the end of the diagonal, what locator.cpp:86 prints (ACGTAACC vs ACGAACCG)
        CHECK(pal->get_cost(0, 0) == 0 && pal->get_cost(0, 3) == 3 && pal->get_cost(2, 0) == 2);   // init_cell
        CHECK(pal->get_parent(0, 2) == INSERT && pal->get_parent(3, 0) == DELETE && pal->get_parent(0, 0) == 0);
        CHECK(pal->get_cost(3, 3) == 0 && pal->get_parent(3, 3) == MATCH);      // ACG vs ACG
        CHECK(pal->get_cost(4, 3) == 1 && pal->get_parent(4, 3) == DELETE);     // ACGT vs ACG
        CHECK(pal->get_cost(4, 4) == 1 && pal->get_parent(4, 4) == MATCH);      // T / A: a substitution is a MATCH op with cost
        CHECK(pal->get_cost(5, 4) == 1 && pal->get_parent(5, 4) == MATCH);      // ACGTA vs ACGA
        CHECK(pal->mat[5][4 - 5 + pal->max_dst].cost == 1);                     // the public array, stripe coordinates
        pal->set_cost(5, 4, 41); pal->set_parent(5, 4, INSERT);
        CHECK(pal->get_cost(5, 4) == 41 && pal->get_parent(5, 4) == INSERT && pal->mat[5][2].cost == 41);
        // walking parents from the goal reproduces edits[]
        int i = pal->matlen_a, j = pal->matlen_b, n = 0;
        pal->set_cost(5, 4, 1); pal->set_parent(5, 4, MATCH);
        while (i > 0 || j > 0) {
            int p = pal->get_parent(i, j);
            CHECK(p == (int)pal->edits[pal->nedit - 1 - n].op);
            if (p == MATCH) { --i; --j; } else if (p == INSERT) --j; else if (p == DELETE) --i; else break;
            ++n;
        }
        CHECK(n == pal->nedit);
        // a new align() starts a new matrix
        seq_accessor sc(a, true, 8), sd(a, true, 8);
        CHECK(8 == pal->align(&sc, &sd) && pal->get_cost(8, 8) == 0 && pal->get_cost(5, 4) == 1 && pal->get_parent(5, 4) == DELETE);
    }
    {   // size guard, seq_aligner.h:104-107
        seq_aligner<100, 30> small(0.3);
        std::string a(120, 'A');
        seq_accessor x((char *)a.c_str(), true, 120), y((char *)a.c_str(), true, 120);
        CHECK(-1 == small.align(&x, &y));
    }
    delete pal;
}

static void ref_basic() {              // test/ref_test.cpp:100-128
    char dna_txt[] = "ACGTAACCGGTTAAACCCGGGTTTTGCAAAAAAAAAAAAAAAA";
    const int sz = (int)strlen(dna_txt);
    unsigned char bseg[24];
    dna_seq::text2bin(dna_txt, bseg, 24);
    ref_seq *pref = new ref_seq(bseg);
    CHECK((unsigned)sz == pref->length());
    CHECK(!pref->contained(-1)); CHECK(pref->contained(0)); CHECK(pref->contained(sz - 1)); CHECK(!pref->contained(sz));
    seq_accessor f = pref->get_accessor(0, true);
    for (int i = 0; i < sz; ++i) CHECK(dna_txt[i] == f.next());
    seq_accessor b = pref->get_accessor(sz - 1, false);
    for (int i = sz - 1; i >= 0; --i) CHECK(dna_txt[i] == b.next());
    hash_table seedmap;
    pref->get_seedmap(seedmap, 0xFFFFFFFF);
    CHECK((size_t)(sz - 15 - 1) == seedmap.size());
    for (int i = 0; i < sz - 16; ++i) CHECK(seedmap.find(dna_seq::encode(dna_txt + i)) != seedmap.end());
    CHECK(seedmap.find(dna_seq::encode(dna_txt + sz - 15)) == seedmap.end());
    // locked try_align against the reference text itself (ref_seq.h:259-265)
    t_aligner al;
    char seg[] = "ACGTAACCGGTTAAACCCGGGTTTTGC";
    seq_accessor ac(seg, true, 27);
    CHECK(pref->try_align(&al, 0, &ac));
    CHECK(al.matlen_a >= OVERLAP_MIN && al.final_cost() == 0);
    delete pref;
}

// test/ref_test.cpp:131-254 -- the unlocked reference: votes, growth, evolve.  The known-answer strings are the
// reference's (test/ref_test.cpp:71-80); each case starts from a fresh ref_seq of dna_txt like its SetUp does.
static char r_txt[]  = "ACGTAACCGGTTAAACCCGGGTTTTGCAAAAAAAAAAAAAAAA";
static char r_txt1[] = "ACGTAACCGGTTAAACCCGGGTGTTGCAAAAAAAAAAAAAAAA";
static char r_txt2[] = "ACGTAACCGGTTAAACCCGGGTTGTTGCAAAAAAAAAAAAAAAA";
static char r_txt3[] = "ACGTAACCGGTTAAACCCGGGTTGGTTGCAAAAAAAAAAAAAAAA";
static char r_txt4[] = "ACGTAACCGGTTAAACCCGGGTTGTTGCAAAAAAAAAAAAAAAAGGCCTTAA";
static char r_txt5[] = "ACGTAACCGGTTAAACCCGGGTTGTTGCAAAAAAAAAAAAAAAAGGCCTTAAC";
static char r_txt6[] = "TTTTACGTAACCGGTTAAACCCGGGTTGTTGCAAAAAAAAAAAAAAAA";
static char r_txt7[] = "TTTTTACGTAACCGGTTAAACCCGGGTTGTTGCAAAAAAAAAAAAAAAA";

struct RefCase {
    ref_seq *pref;
    t_aligner *pal;
    int sz;
    RefCase() {
        unsigned char bseg[24];
        sz = (int)strlen(r_txt);
        dna_seq::text2bin(r_txt, bseg, 24);
        pref = new ref_seq(bseg);
        pal = new t_aligner();
    }
    ~RefCase() { delete pref; delete pal; }
    bool reads_forward(const char *want, int n) {          // the reference from position 0 reads `want`
        seq_accessor a = pref->get_accessor(0, true);
        for (int i = 0; i < n; ++i) if (want[i] != a.next()) return false;
        return true;
    }
    bool reads_backward(const char *want, int from) {      // ... and backwards from position `from`
        seq_accessor a = pref->get_accessor(from, false);
        for (int i = from; i >= 0; --i) if (want[i] != a.next()) return false;
        return true;
    }
};

static void ref_unlocked() {
    {   // grow, :131-142
        RefCase c;
        char post[] = "CGT", pre[] = "TGC";
        c.pref->append(post, 3);
        CHECK(c.pref->contained(c.sz + 2)); CHECK(!c.pref->contained(c.sz + 3));
        c.pref->prepend(pre, 3);
        CHECK(c.pref->contained(-3)); CHECK(!c.pref->contained(-4));
        CHECK((unsigned)c.sz == c.pref->length());
    }
    {   // change, :144-154: two votes for a substitution beat the reference's own
        RefCase c;
        seq_accessor seg(r_txt1, true, (int)strlen(r_txt1));
        CHECK(c.pref->try_align(c.pal, 0, &seg));
        seg.reset(0);
        CHECK(c.pref->try_align(c.pal, 0, &seg));
        c.pref->evolve();
        CHECK(c.reads_forward(r_txt1, (int)strlen(r_txt1)));
    }
    {   // remove, :156-166: the first base is voted out
        RefCase c;
        seq_accessor seg(r_txt + 1, true, c.sz - 1);
        CHECK(c.pref->try_align(c.pal, 0, &seg));
        seg.reset(0);
        CHECK(c.pref->try_align(c.pal, 0, &seg));
        c.pref->evolve();
        CHECK((unsigned)(c.sz - 1) == c.pref->length());
        CHECK(c.reads_forward(r_txt + 1, c.sz - 1));
    }
    {   // insert, :168-180
        RefCase c;
        const int n = (int)strlen(r_txt2);
        seq_accessor seg(r_txt2, true, n);
        CHECK(c.pref->try_align(c.pal, 0, &seg));
        CHECK(n == c.pal->nedit);
        seg.reset(0);
        CHECK(c.pref->try_align(c.pal, 0, &seg));
        c.pref->evolve();
        CHECK((unsigned)n == c.pref->length());
        CHECK(c.reads_forward(r_txt2, n));
    }
    {   // insert2, :182-191: one vote for two inserted bases: only one of them gets a box of its own
        RefCase c;
        seq_accessor seg(r_txt3, true, (int)strlen(r_txt3));
        CHECK(c.pref->try_align(c.pal, 0, &seg));
        c.pref->evolve();
        CHECK((unsigned)(c.sz + 1) == c.pref->length());
        CHECK(c.reads_forward(r_txt2, c.sz + 1));
    }
    {   // back_insert, :193-209
        RefCase c;
        const int n = (int)strlen(r_txt2);
        seq_accessor seg(r_txt2 + n - 1, false, n);
        CHECK(c.pref->try_align(c.pal, c.sz - 1, &seg));
        CHECK(n == c.pal->nedit);
        seg.reset(0);
        CHECK(c.pref->try_align(c.pal, c.sz - 1, &seg));
        c.pref->evolve();
        CHECK((unsigned)n == c.pref->length());
        CHECK(c.reads_backward(r_txt2, n - 1));
    }
    {   // back_insert2, :211-221
        RefCase c;
        const int n = (int)strlen(r_txt3);
        seq_accessor seg(r_txt3 + n - 1, false, n);
        CHECK(c.pref->try_align(c.pal, c.sz - 1, &seg));
        CHECK(n == c.pal->nedit);
        c.pref->evolve();
        CHECK((unsigned)(c.sz + 1) == c.pref->length());
        CHECK(c.reads_backward(r_txt2, c.sz));
    }
    {   // append, :223-236: a read running off the end grows the reference
        RefCase c;
        seq_accessor s4(r_txt4, true, (int)strlen(r_txt4));
        CHECK(c.pref->try_align(c.pal, 0, &s4));
        CHECK(c.pref->contained(c.sz + 1));
        const int n5 = (int)strlen(r_txt5);
        seq_accessor s5(r_txt5, true, n5);
        CHECK(c.pref->try_align(c.pal, 0, &s5));
        c.pref->evolve();
        CHECK((unsigned)n5 == c.pref->length());
        CHECK(c.reads_forward(r_txt5, n5));
    }
    {   // prepend, :238-254
        RefCase c;
        const int n6 = (int)strlen(r_txt6), n7 = (int)strlen(r_txt7);
        seq_accessor s6(r_txt6 + n6 - 1, false, n6);
        CHECK(c.pref->try_align(c.pal, c.sz - 1, &s6));
        CHECK(c.pref->contained(-1));
        seq_accessor s7(r_txt7 + n7 - 1, false, n7);
        CHECK(c.pref->try_align(c.pal, c.sz - 1, &s7));
        c.pref->evolve();
        CHECK((unsigned)n7 == c.pref->length());
        CHECK(c.reads_backward(r_txt7, n7 - 1));
    }
}

int main(int argc, char **argv) {
    dna_binary();
    accessor();
    aligner(argc > 1 ? argv[1] : "tests/golden/real_align.txt");
    ref_basic();
    ref_unlocked();
    printf("%d checks, %d failed\n", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
