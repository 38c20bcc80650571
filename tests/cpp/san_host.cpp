// san_host.cpp -- host code of the boundary and the CPU oracle under AddressSanitizer / UBSan (sanitizers on
// host code; GPU ASan is not available).  Built by tests/test_sanitizers.py with
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=all  san_host.cpp pba_codec.cpp pba_synth.cpp  +  pba_oracle.c
// and run on (1) the golden vectors of tests/golden/codec.json and align_kat.json, exported to a flat text file by the
// test, and (2) seeded random inputs in EXACT-SIZE heap buffers, cross-checking the product's host codec against the
// oracle's (so an out-of-bounds read or a shift / overflow UB in either aborts the run).
// Test infrastructure: links oracle/ on purpose.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "pba.h"
#include "pba_oracle.h"

static int g_fail = 0;
#define CHECK(cond, ...)                                     \
    do {                                                     \
        if (!(cond)) {                                       \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                    \
            fprintf(stderr, "\n");                           \
            ++g_fail;                                        \
        }                                                    \
    } while (0)

static std::string unhex(const char *h) {
    std::string s;
    if (h[0] == '-' && !h[1]) return s;
    for (size_t i = 0; h[i] && h[i + 1]; i += 2) {
        unsigned v;
        sscanf(h + i, "%2x", &v);
        s.push_back((char)v);
    }
    return s;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}
static std::string rand_dna(size_t n, bool with_other) {
    std::string s(n, 'A');
    for (size_t i = 0; i < n; ++i) {
        const uint32_t r = rnd();
        s[i] = (with_other && (r >> 8) % 37 == 0) ? "NnacgtX\n"[(r >> 16) & 7] : "ACGT"[r & 3];
    }
    return s;
}
// error-laden copy: substitutions, insertions, deletions at rate ~e
static std::string mutate(const std::string &s, double e) {
    std::string o;
    for (size_t i = 0; i < s.size();) {
        const double u = (rnd() & 0xFFFFFF) / (double)0x1000000;
        if (u < e / 3) { o.push_back("ACGT"[rnd() & 3]); continue; }         // insertion
        if (u >= 2 * e / 3) o.push_back(u < e ? "ACGT"[rnd() & 3] : s[i]);   // (else: deletion)
        ++i;
    }
    return o;
}

// ---- golden vectors from the flat file -------------------------------------------------------------------------
static void run_vectors(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    std::vector<char> line(1 << 20);
    std::vector<uint8_t> seedrec;          // record of SEEDTEXT + 64 zero bytes (the goldens probe seed_at's byte-offset form
                                           // past a short record, SURVEY B1: the caller owns that slack)
    size_t n_enc = 0, n_aln = 0, n_seed = 0, n_t2b = 0, n_mask = 0;
    orc_aligner *al = orc_aligner_new(0, 0);
    while (fgets(line.data(), (int)line.size(), f)) {
        char kind[16], a[1 << 16], b[1 << 16], c[1 << 16];
        if (sscanf(line.data(), "%15s", kind) != 1) continue;
        if (!strcmp(kind, "ENC")) {
            unsigned code;
            sscanf(line.data(), "%*s %s %u", a, &code);
            const std::string w = unhex(a);
            CHECK(w.size() == 16, "ENC size");
            std::vector<char> exact(w.begin(), w.end());        // no NUL, no slack: encode reads exactly 16 chars
            CHECK(pba_encode16(exact.data()) == code && orc_encode(exact.data()) == code, "ENC %s", a);
            char back[16];
            pba_decode16(code, back);
            char back2[16];
            orc_decode(code, back2);
            CHECK(!memcmp(back, back2, 16), "DEC %s", a);
            ++n_enc;
        } else if (!strcmp(kind, "C2I")) {
            int ch, v;
            sscanf(line.data(), "%*s %d %d", &ch, &v);
            CHECK(orc_c2i(ch) == v, "C2I %d", ch);
        } else if (!strcmp(kind, "SEEDTEXT")) {
            sscanf(line.data(), "%*s %s", a);
            const std::string t = unhex(a);
            const size_t need = 4 + (t.size() + 3) / 4;
            seedrec.assign(need + 64, 0);
            CHECK(pba_text2bin(t.data(), t.size(), seedrec.data(), need) == need, "SEEDTEXT text2bin");
        } else if (!strcmp(kind, "SEEDAT")) {
            int pos; unsigned want;
            sscanf(line.data(), "%*s %d %u", &pos, &want);
            CHECK(pba_seed_at(seedrec.data(), pos) == want && orc_seed_at(seedrec.data(), pos) == want, "SEEDAT %d", pos);
            CHECK(pba_seed_at_fixed(seedrec.data(), pos) == orc_seed_at_fixed(seedrec.data(), pos), "SEEDAT fixed %d", pos);
            ++n_seed;
        } else if (!strcmp(kind, "MASK")) {
            unsigned m;
            sscanf(line.data(), "%*s %s %u", a, &m);
            const std::string p = unhex(a);
            CHECK(pba_mask_from_pattern(p.c_str()) == m && orc_mask_from_pattern(p.c_str()) == m, "MASK %s", p.c_str());
            ++n_mask;
        } else if (!strcmp(kind, "T2B")) {
            sscanf(line.data(), "%*s %s %s %s", a, b, c);
            const std::string t = unhex(a), rec = unhex(b), back = unhex(c);
            std::vector<uint8_t> out(rec.size()), out2(rec.size());                 // exact size
            CHECK(pba_text2bin(t.data(), t.size(), out.data(), out.size()) == rec.size(), "T2B size");
            CHECK(orc_text2bin(t.data(), t.size(), out2.data(), out2.size()) == rec.size(), "T2B size (oracle)");
            CHECK(!memcmp(out.data(), rec.data(), rec.size()) && !memcmp(out2.data(), rec.data(), rec.size()), "T2B bytes");
            std::vector<char> txt(back.size() + 1), txt2(back.size() + 1);          // exact size
            CHECK(pba_bin2text(out.data(), txt.data(), txt.size()) == back.size() && !memcmp(txt.data(), back.data(), back.size()), "B2T");
            CHECK(orc_bin2text(out.data(), txt2.data(), txt2.size()) == back.size() && !memcmp(txt2.data(), back.data(), back.size()), "B2T (oracle)");
            if (out.size() > 4) CHECK(pba_bin2text(out.data(), txt.data(), txt.size() - 1) == 0, "B2T must refuse a short buffer");
            ++n_t2b;
        } else if (!strcmp(kind, "ALN")) {
            double R; int af, bf, e[9];
            sscanf(line.data(), "%*s %s %s %lf %d %d %d %d %d %d %d %d %d %d %d", a, b, &R, &af, &bf, &e[0], &e[1], &e[2], &e[3], &e[4],
                   &e[5], &e[6], &e[7], &e[8]);
            const std::string sa = unhex(a), sb = unhex(b);
            std::vector<char> va(sa.begin(), sa.end()), vb(sb.begin(), sb.end());   // exact size, no NUL
            std::vector<uint8_t> ops(sa.size() + sb.size() + 1);
            orc_result r;
            const char *pa = va.empty() ? nullptr : (af ? va.data() : va.data() + va.size() - 1);
            const char *pb = vb.empty() ? nullptr : (bf ? vb.data() : vb.data() + vb.size() - 1);
            char dummy = 0;
            orc_align(al, pa ? pa : &dummy, af, (int)sa.size(), pb ? pb : &dummy, bf, (int)sb.size(), R, &r, ops.data());
            CHECK(r.rc == e[0] && r.len_a == e[4] && r.len_b == e[5] && r.max_dst == e[6], "ALN %zu rc/params", n_aln);
            if (e[0] >= 0)
                CHECK(r.cost == e[1] && r.matlen_a == e[2] && r.matlen_b == e[3] && r.nedit == e[7] && (r.nedit ? ops[0] : 0) == e[8],
                      "ALN %zu result", n_aln);
            ++n_aln;
        }
    }
    fclose(f);
    orc_aligner_free(al);
    printf("vectors: %zu ENC, %zu SEEDAT, %zu MASK, %zu T2B, %zu ALN\n", n_enc, n_seed, n_mask, n_t2b, n_aln);
    CHECK(n_enc >= 5 && n_seed > 10 && n_mask >= 8 && n_t2b > 3 && n_aln > 300, "vector file incomplete");
}

// ---- seeded random cross-checks in exact-size buffers --------------------------------------------------------------
static void run_random() {
    // codec: product vs oracle, every in-bounds window
    for (int it = 0; it < 300; ++it) {
        const size_t L = 16 + rnd() % 400;
        const std::string t = rand_dna(L, it % 3 == 0);
        const size_t need = 4 + (L + 3) / 4;
        std::vector<uint8_t> r1(need), r2(need);
        CHECK(pba_text2bin(t.data(), L, r1.data(), need) == need && orc_text2bin(t.data(), L, r2.data(), need) == need, "rand t2b");
        CHECK(r1 == r2, "rand t2b bytes");
        CHECK(pba_text2bin(t.data(), L, r1.data(), need - 1) == 0, "t2b must refuse a short buffer");
        for (size_t pos = 0; pos + 16 <= L; ++pos) {
            std::vector<char> w(t.begin() + pos, t.begin() + pos + 16);
            const uint32_t e = pba_encode16(w.data());
            CHECK(e == orc_encode(w.data()), "rand encode");
            CHECK(pba_seed_at_fixed(r1.data(), (int)pos) == e && orc_seed_at_fixed(r1.data(), (int)pos) == e, "rand seed_at_fixed %zu/%zu", pos, L);
            if ((pos & 3) || pos + 4 <= (L + 3) / 4)          // the byte-offset form (dna_seq.h:64) stays inside the payload
                CHECK(pba_seed_at(r1.data(), (int)pos) == orc_seed_at(r1.data(), (int)pos), "rand seed_at");
        }
        for (int k = 0; k < 256; ++k) CHECK(pba_value_at((uint8_t)k, it & 3) == "ACGT"[(k >> (6 - 2 * (it & 3))) & 3], "value_at");
    }
    // record walk over a file image with ragged records, the last one cut short
    {
        std::vector<uint8_t> img;
        std::vector<size_t> lens;
        for (int k = 0; k < 40; ++k) {
            const size_t L = rnd() % 1500;
            const std::string t = rand_dna(L, false);
            const size_t need = 4 + (L + 3) / 4, at = img.size();
            img.resize(at + need);
            pba_text2bin(t.data(), L, img.data() + at, need);
            lens.push_back(L);
        }
        for (size_t cut : {img.size(), img.size() - 1, img.size() - 5, (size_t)3, (size_t)0}) {
            std::vector<uint8_t> exact(img.begin(), img.begin() + cut);
            std::vector<uint64_t> o1(64), o2(64);
            size_t t1 = 0, t2 = 0;
            const size_t k1 = pba_open_binary(exact.data(), cut, 500, 20000, o1.data(), 64, &t1);
            const size_t k2 = orc_open_binary(exact.data(), cut, 500, 20000, o2.data(), 64, &t2);
            CHECK(k1 == k2 && t1 == t2 && o1 == o2, "open_binary cut %zu", cut);
            CHECK(pba_open_binary(exact.data(), cut, 500, 20000, nullptr, 0, nullptr) == k1, "open_binary count-only");
        }
    }
    // generator: thread-count independence, exact-size outputs
    {
        std::vector<char> g(20000);
        pba_synth_genome(5, g.data(), g.size());
        std::vector<char> r1(50 * 300), r2(50 * 300);
        std::vector<uint32_t> s1(50), s2(50);
        CHECK(pba_synth_reads(9, g.data(), g.size(), 50, 300, 0.05, 0.05, 0.05, r1.data(), s1.data(), 1) == 0, "synth_reads");
        CHECK(pba_synth_reads(9, g.data(), g.size(), 50, 300, 0.05, 0.05, 0.05, r2.data(), nullptr, 3) == 0, "synth_reads mt");
        CHECK(r1 == r2, "synth_reads is thread-count independent");
        std::vector<char> tiny(g.begin(), g.begin() + 100);      // a genome shorter than a read wraps around, inside its 100 bytes
        CHECK(pba_synth_reads(9, tiny.data(), tiny.size(), 50, 300, 0.05, 0.05, 0.05, r2.data(), nullptr, 3) == 0, "synth_reads on a tiny genome");
        CHECK(pba_synth_reads(9, tiny.data(), tiny.size(), 50, 300, 0.5, 0.5, 0.05, r2.data(), nullptr, 3) != 0, "synth_reads must refuse rates >= 1");
    }
    // oracle: index (both orders), find / dump, aligner with traceback, locator driver on threads, locked round, consensus
    {
        const int L = 6000;
        std::vector<char> g(L);
        pba_synth_genome(77, g.data(), L);
        const uint32_t mask = pba_mask_from_pattern("111*11*11*1*1111");
        orc_seedmap *sm = orc_seedmap_new(1 << 10);
        const size_t n_all = orc_index_all(sm, g.data(), L, mask);
        CHECK(n_all == orc_seedmap_entries(sm) && n_all <= (size_t)L, "index_all");
        std::vector<uint32_t> keys(n_all);
        std::vector<int32_t> pos(n_all);
        CHECK(orc_seedmap_dump(sm, keys.data(), pos.data(), n_all) == n_all, "dump");
        std::vector<int32_t> hits(4);
        for (size_t k = 0; k < n_all; k += 97) CHECK(orc_seedmap_find(sm, keys[k], hits.data(), 4) >= 1, "find");
        CHECK(orc_seedmap_find(sm, 0u, hits.data(), 4) == 0, "key 0 is never inserted");
        orc_seedmap_clear(sm);
        for (int len : {10, 16, 17, 43, 5000}) {
            orc_seedmap_clear(sm);
            const unsigned rv = orc_index_head_tail(sm, g.data(), len, 0xFFFFFFFFu);
            CHECK((int)rv == (len > 16 ? len - 16 : len - 16) || len <= 16, "head_tail rv %d", len);
        }
        orc_seedmap_free(sm);

        std::vector<char> reads(40 * 700);
        std::vector<uint64_t> offs(41);
        CHECK(pba_synth_reads(78, g.data(), L, 40, 700, 0.05, 0.05, 0.05, reads.data(), nullptr, 2) == 0, "reads");
        for (int k = 0; k <= 40; ++k) offs[k] = (uint64_t)k * 700;
        std::vector<orc_loc_row> rows(40), rows1(40);
        orc_loc_stats st, st1;
        CHECK(orc_locator_run(g.data(), L, mask, 0.30, 50, 500, 0, 0, reads.data(), offs.data(), 40, 3, rows.data(), &st) == 0, "locator mt");
        CHECK(orc_locator_run(g.data(), L, mask, 0.30, 50, 500, 0, 0, reads.data(), offs.data(), 40, 1, rows1.data(), &st1) == 0, "locator st");
        CHECK(!memcmp(rows.data(), rows1.data(), sizeof(orc_loc_row) * 40) && st.n_pairs == st1.n_pairs && st.n_located > 20, "locator rows");
        orc_pool_release();

        // locked round over a binary read file image (exact size + the slack the byte-offset seed_at form needs)
        std::vector<uint8_t> img;
        std::vector<uint64_t> roff;
        for (int k = 0; k < 40; ++k) {
            const size_t need = 4 + (700 + 3) / 4, at = img.size();
            img.resize(at + need);
            pba_text2bin(reads.data() + k * 700, 700, img.data() + at, need);
            roff.push_back(at);
        }
        img.resize(img.size() + 64, 0);
        std::vector<orc_ss_row> ss(40);
        for (int buggy = 0; buggy < 2; ++buggy)
            CHECK(orc_spaced_round(g.data(), L, mask, 0.30, 32, 64, buggy, img.data(), roff.data(), 40, 2, ss.data()) >= 0, "spaced_round");
        orc_pool_release();

        // aligner + traceback on related and unrelated pairs, both directions, exact-size buffers
        orc_aligner *al = orc_aligner_new(0, 0);
        for (int it = 0; it < 60; ++it) {
            const std::string a = rand_dna(20 + rnd() % 900, false);
            const std::string b = it % 4 == 3 ? rand_dna(20 + rnd() % 900, false) : mutate(a, 0.02 * (it % 10)) + rand_dna(rnd() % 200, false);
            if (b.empty()) continue;
            std::vector<char> va(a.begin(), a.end()), vb(b.begin(), b.end());
            std::vector<uint8_t> ops(a.size() + b.size());
            const int af = it & 1, bf = (it >> 1) & 1;
            orc_result r;
            orc_align(al, af ? va.data() : va.data() + va.size() - 1, af, (int)va.size(), bf ? vb.data() : vb.data() + vb.size() - 1, bf,
                      (int)vb.size(), 0.3, &r, ops.data());
            if (r.rc >= 0) {
                int na = 0, nb = 0;
                for (int k = 0; k < r.nedit; ++k) { na += ops[k] != 2; nb += ops[k] != 3; }
                CHECK(na == r.matlen_a && nb == r.matlen_b, "script consumes matlen_a / matlen_b");
            }
        }
        // consensus: create / try (votes + growth) / evolve
        {
            const std::string ref(g.begin() + 1000, g.begin() + 3000);
            orc_cons *c = orc_cons_new(ref.data(), (int)ref.size(), 1, 20000);
            int32_t out[8];
            for (int k = 0; k < 20; ++k) {
                const int start = 900 + (int)(rnd() % 1800);
                const std::string seg = mutate(std::string(g.begin() + start, g.begin() + start + 600), 0.1);
                std::vector<char> vs(seg.begin(), seg.end());
                const int pos = start - 1000;
                if (pos < 0 || pos >= (int)ref.size()) continue;
                orc_cons_try(c, al, pos, vs.data(), (int)vs.size(), 1, 0.3, 16, out);
            }
            std::vector<uint16_t> sel(4 * 20000), sup(4 * 20000);
            std::vector<int32_t> tot(20000);
            int32_t ext[3];
            const int nb = orc_cons_dump(c, sel.data(), sup.data(), tot.data(), 20000, ext);
            CHECK(nb == ext[1] - ext[0], "cons dump");
            orc_cons_evolve(c);
            std::vector<char> txt(20000);
            CHECK(orc_cons_text(c, txt.data(), 20000) > 1000, "cons text");
            orc_cons_free(c);
        }
        orc_aligner_free(al);
    }
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: san_host vectors.txt\n"); return 2; }
    run_vectors(argv[1]);
    run_random();
    if (g_fail) { fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
    printf("san_host ok\n");
    return 0;
}
