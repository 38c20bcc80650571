// spaced_seed_gpu.cpp -- the reference's `spaced_seed` (src/spaced_seed.cpp) on the MI355X through the C ABI: same command
// line, same stdout (the consensus after every round, spaced_seed.cpp:452-453), the same progress lines on stderr.
//
//   spaced_seed_gpu [-f ref_file] [-r ratio] [-d dumpfile] [-m rounds] [-t trials] [-l] [-s srand_seed] seq_file seed_file
//
// seq_file: binary reads ([u32 length][2-bit packed bases])*, as binary_test writes them; seed_file: one pattern per
// line ('1' = care).  Without -l every read that aligns votes and may grow the reference inside the round
// (pba_cons_round), with -l the reference never changes (pba_spaced_round).  The reference draws the initial segment and
// the seed of a round with rand() after srand(time(0)); -s fixes that seed (glibc's rand(): the draws are then the ones
// the reference would make with the same srand).
// Differences, on purpose: a ref_file line keeps no trailing newline (the reference keeps the '\n' fgets returns as a
// base of the text, spaced_seed.cpp:197-202); -d lines are written after the round rather than during it (same lines,
// same order).
//
//   g++ -O2 -I include -o spaced_seed_gpu examples/spaced_seed_gpu.cpp -L pacbioassembly_amd/lib -lpba -Wl,-rpath,$PWD/pacbioassembly_amd/lib
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "pba.h"

#define SEQ_THRESHOLD 500          // spaced_seed.cpp:36
#define MAX_READ_LEN 20000         // common.h:33
#define MAX_SEQ_LEN 800000         // common.h:31
#define OVERLAP_MIN 64             // common.h:39
#define LOG(...) fprintf(stderr, __VA_ARGS__)

static pba_ctx *ctx = NULL;
static void check(int st, const char *what) {
    if (st == PBA_OK) return;
    fprintf(stderr, "%s: %s (%s)\n", what, pba_strerror(st), ctx ? pba_ctx_error(ctx) : "");
    exit(EXIT_FAILURE);
}

static const char *usage_str = "usage: %s [-f ref_file] [-r ratio] [-d dumpfile] [-m max_round] [-t max_trial] [-l] [-s srand_seed] seq_file seed_file\n";

int main(int argc, char *argv[]) {
    double ratio = 0.3;                                               // MAXR, common.h:37
    bool locked = false;
    int max_round = INT_MAX, max_trial = 32, opt;
    unsigned srand_seed = (unsigned)time(0);                          // spaced_seed.cpp:194
    FILE *fpref = NULL, *fpdump = NULL;
    if (argc < 3) { fprintf(stderr, usage_str, argv[0]); return EXIT_FAILURE; }
    while ((opt = getopt(argc, argv, "f:r:d:m:t:s:lh")) != -1) {      // spaced_seed.cpp:367-396
        switch (opt) {
            case 'h': fprintf(stdout, usage_str, argv[0]); return EXIT_SUCCESS;
            case 'f': if (!(fpref = fopen(optarg, "r"))) { perror("failed to read ref_file"); return EXIT_FAILURE; } break;
            case 'd': if (!(fpdump = fopen(optarg, "w"))) { perror("failed to create dump file"); return EXIT_FAILURE; } break;
            case 'r': ratio = atof(optarg); break;
            case 'l': locked = true; break;
            case 'm': max_round = atoi(optarg); break;
            case 't': max_trial = atoi(optarg); break;
            case 's': srand_seed = (unsigned)strtoul(optarg, NULL, 10); break;
            default: fprintf(stderr, usage_str, argv[0]); return EXIT_FAILURE;
        }
    }
    if (optind + 2 > argc) { fprintf(stderr, usage_str, argv[0]); return EXIT_FAILURE; }

    // open_binary, spaced_seed.cpp:310-345: the whole file, records of SEQ_THRESHOLD < len < MAX_READ_LEN kept
    FILE *fp = fopen(argv[optind], "rb");
    if (!fp) { perror("open"); return EXIT_FAILURE; }
    std::vector<uint8_t> file;
    for (uint8_t tmp[1 << 16];;) {
        const size_t got = fread(tmp, 1, sizeof tmp, fp);
        if (!got) break;
        file.insert(file.end(), tmp, tmp + got);
    }
    fclose(fp);
    check(pba_ctx_create(0, &ctx), "pba_ctx_create");
    pba_seqs *reads = NULL;
    check(pba_seqs_from_records(ctx, file.data(), file.size(), SEQ_THRESHOLD, MAX_READ_LEN, &reads), "seq_file");
    const uint32_t n = pba_seqs_count(reads);
    LOG("indices: size %u\n", n);
    LOG("number of seeding trial: %d\n", max_trial);
    std::vector<uint32_t> lens(n ? n : 1);
    check(pba_seqs_lengths(reads, lens.data(), n), "lengths");

    // init, spaced_seed.cpp:186-230
    srand(srand_seed);
    std::string ref;
    int weight = 1;
    if (fpref) {
        for (int ch; (ch = fgetc(fpref)) != EOF && ch != '\n';) ref.push_back((char)ch);
        if (fscanf(fpref, "%d", &weight) != 1) weight = 1;
        LOG("reference weight: %d\n", weight);
        fclose(fpref);
    } else {
        if (!n) { fprintf(stderr, "no reads\n"); return EXIT_FAILURE; }
        const uint32_t pick = (uint32_t)rand() % n;                   // spaced_seed.cpp:205-207
        ref.resize(lens[pick] + 1);
        check(pba_seqs_get_text(ctx, reads, pick, &ref[0], ref.size()), "initial reference");
        ref.resize(lens[pick]);
        LOG("%u selected as the initial reference.\n", pick);
    }
    LOG("ref_len: %d\n", (int)ref.size());
    std::vector<uint32_t> seeds;
    if (!(fp = fopen(argv[optind + 1], "r"))) { perror("failed to open seedfile"); return EXIT_FAILURE; }
    for (char line[1024]; fgets(line, sizeof line, fp);) {
        line[strlen(line) - 1] = '\0';                                // spaced_seed.cpp:225: the last character goes, newline or not
        seeds.push_back(pba_mask_from_pattern(line));
        LOG("seed %s: %08x\n", line, seeds.back());
    }
    fclose(fp);
    if (seeds.empty()) { fprintf(stderr, "no seeds\n"); return EXIT_FAILURE; }

    pba_cons *cons = NULL;                                            // unlocked: vote boxes and text in HBM
    pba_seqs *lref = NULL;                                            // locked: the text, packed once
    if (locked) {
        const uint64_t off[2] = {0, ref.size()};
        check(pba_seqs_from_text(ctx, ref.data(), off, 1, 1, &lref), "reference");
    } else check(pba_cons_create(ctx, ref.data(), (int)ref.size(), weight, MAX_SEQ_LEN, &cons), "reference");

    std::vector<pba_ss_row> rows(n ? n : 1);
    std::vector<char> text((size_t)3 * MAX_SEQ_LEN + 1), rtext(MAX_READ_LEN + 1);
    // spaced_seed.cpp:287-293: the matched stretch of the reference, then of the read; t[p] = reference position p
    auto dump_pair = [&](const pba_ss_row &w, uint32_t r, const char *t) {
        const bool fwd = w.dir == 1;
        const int r_off = fwd ? w.ref_pos : w.ref_pos + 15;
        for (int i = 0; i < w.matlen_a; ++i) fputc(t[fwd ? r_off + i : r_off - i], fpdump);
        fputc('\n', fpdump);
        check(pba_seqs_get_text(ctx, reads, r, rtext.data(), rtext.size()), "read text");
        const int s_off = fwd ? w.j : (int)lens[r] - w.j - 1;
        for (int i = 0; i < w.matlen_b; ++i) fputc(rtext[fwd ? s_off + i : s_off - i], fpdump);
        fputc('\n', fpdump);
        fflush(fpdump);
    };

    if (locked) {
        // the reference never changes: the whole loop in one call, the draws of rand() handed over in order
        const int cap = max_round < 65536 ? (max_round > 0 ? max_round : 1) : 65536;
        std::vector<uint32_t> picks((size_t)cap);
        for (int k = 0; k < cap; ++k) picks[(size_t)k] = (uint32_t)rand();
        std::vector<int32_t> found_round(n ? n : 1);
        std::vector<pba_ss_round_log> log((size_t)cap);
        int n_rounds = 0;
        check(pba_spaced_multi(ctx, lref, 0, reads, ratio, max_trial, OVERLAP_MIN, 1, PBA_KERNEL_AUTO, seeds.data(), (int)seeds.size(),
                               picks.data(), cap, max_round, rows.data(), found_round.data(), log.data(), cap, &n_rounds), "rounds");
        int nfailure = 0;
        for (int k = 0; k < n_rounds && k < cap; ++k) {
            LOG("--------------- round %d ---------\n", log[(size_t)k].round);
            LOG("seed: %08x\n", log[(size_t)k].mask);
            LOG("reference length: %d\n", (int)ref.size());
            for (uint32_t r = 0; r < n; ++r) {
                if (found_round[r] != log[(size_t)k].round) continue;
                LOG("found %u at cost %d:\tref_ml=%d,\tseg_ml=%d\n", r, rows[r].cost, rows[r].matlen_a, rows[r].matlen_b);
                if (fpdump) dump_pair(rows[r], r, ref.data());
            }
            LOG("#matches: %d\n", log[(size_t)k].n_found);
            if (log[(size_t)k].n_found) nfailure = 0;
            else if (++nfailure == (int)seeds.size()) break;          // spaced_seed.cpp:450: no consensus line for this round
            fwrite(ref.data(), 1, ref.size(), stdout);                // evolve of a locked ref_seq changes nothing
            fputc('\n', stdout);
        }
    } else {
        std::vector<uint32_t> pool(n);
        for (uint32_t r = 0; r < n; ++r) pool[r] = r;
        int nfailure = 0;
        for (int nround = 1; nround <= max_round; ++nround) {         // spaced_seed.cpp:410-453
            const uint32_t seed = nfailure == 0 ? seeds[(size_t)rand() % seeds.size()] : seeds[(size_t)nfailure - 1];
            LOG("--------------- round %d ---------\n", nround);
            LOG("seed: %08x\n", seed);
            int32_t ext[3];
            pba_cons_round_stats st;
            check(pba_cons_extent(cons, ext), "extent");
            LOG("reference length: %d\n", ext[2]);
            check(pba_cons_round(ctx, cons, reads, pool.data(), (uint32_t)pool.size(), seed, ratio, max_trial, OVERLAP_MIN, 1,
                                 PBA_KERNEL_AUTO, 26000, 6000, rows.data(), &st), "round");
            LOG("seedmap size: %u\n", st.n_index);
            int32_t tl = 0;
            if (fpdump) {
                check(pba_cons_extent(cons, ext), "extent");
                check(pba_cons_text(ctx, cons, text.data(), (int)text.size(), &tl), "text");
            }
            std::vector<uint32_t> rest;
            for (size_t k = 0; k < pool.size(); ++k) {
                const uint32_t r = pool[k];
                if (!rows[r].found) { rest.push_back(r); continue; }
                LOG("found %u at cost %d:\tref_ml=%d,\tseg_ml=%d\n", r, rows[r].cost, rows[r].matlen_a, rows[r].matlen_b);
                if (fpdump) dump_pair(rows[r], r, text.data() - ext[0]);
            }
            pool.swap(rest);
            LOG("#matches: %d\n", st.n_found);
            if (st.n_found != 0) nfailure = 0;                        // spaced_seed.cpp:446-449
            else if (++nfailure == (int)seeds.size()) break;
            int32_t new_len = 0;
            check(pba_cons_evolve(ctx, cons, text.data(), (int)text.size(), &new_len), "evolve");
            fwrite(text.data(), 1, (size_t)new_len, stdout);          // dump_seq(stdout, ...), spaced_seed.cpp:452-453
            fputc('\n', stdout);
            fflush(stdout);
        }
    }
    if (fpdump) fclose(fpdump);
    pba_cons_destroy(cons);
    pba_seqs_destroy(lref);
    pba_seqs_destroy(reads);
    pba_ctx_destroy(ctx);
    return EXIT_SUCCESS;
}
