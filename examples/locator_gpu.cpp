// locator_gpu.cpp -- the reference's `locator` (src/locator.cpp) on the MI355X through the C ABI: same command line
// (`locator_gpu contig_file seed [R] < seq_file`), same TSV on stdout: running id among the reads of >= 500 bases,
// contig position, cost, len - j, and get_cost(len - j, len - j) -- the cost at the end of the diagonal, a cell of the
// reference's DP matrix that its sweep writes when the contig remainder is at least as long as the read remainder
// (the reference prints stale memory otherwise, this prints -1).  R defaults to the reference's 0.15
// (locator.cpp:68).
//
//   g++ -O2 -I include -o locator_gpu examples/locator_gpu.cpp -L pacbioassembly_amd/lib -lpba -Wl,-rpath,$PWD/pacbioassembly_amd/lib
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "pba.h"

static void die(pba_ctx *ctx, const char *what, int st) {
    fprintf(stderr, "%s: %s (%s)\n", what, pba_strerror(st), ctx ? pba_ctx_error(ctx) : "");
    exit(EXIT_FAILURE);
}

int main(int argc, char *argv[]) {
    if (argc <= 2) {
        fprintf(stderr, "usage: locator_gpu contig_file seed [R] < seq_file\n");
        return EXIT_FAILURE;
    }
    const double R = argc > 3 ? atof(argv[3]) : 0.15;
    FILE *fp = fopen(argv[1], "r");
    if (!fp) { perror(argv[1]); return EXIT_FAILURE; }
    std::string contig;
    for (int ch; (ch = fgetc(fp)) != EOF && ch != '\n' && ch != ' ' && ch != '\t' && ch != '\r';) contig.push_back((char)ch);   // fscanf("%s"), locator.cpp:49
    fclose(fp);
    if (!contig.empty() && contig[0] == 'N') contig[0] = 'A';        // locator.cpp:57-60 only ever looks at the first base

    std::string text;                                                 // every whitespace-separated token of stdin is a read (locator.cpp:70)
    std::vector<uint64_t> offs(1, 0);
    std::string tok;
    for (int ch; (ch = getchar()) != EOF;) {
        if (ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r') {
            if (!tok.empty()) { text += tok; offs.push_back(text.size()); tok.clear(); }
        } else tok.push_back((char)ch);
    }
    if (!tok.empty()) { text += tok; offs.push_back(text.size()); }
    const uint32_t nreads = (uint32_t)offs.size() - 1;

    pba_ctx *ctx = NULL;
    int st = pba_ctx_create(0, &ctx);
    if (st != PBA_OK) die(NULL, "pba_ctx_create", st);
    pba_seqs *T = NULL, *Rd = NULL;
    pba_index *ix = NULL;
    const uint64_t toff[2] = {0, contig.size()};
    if ((st = pba_seqs_from_text(ctx, contig.data(), toff, 1, 0, &T)) != PBA_OK) die(ctx, "contig", st);
    if ((st = pba_seqs_from_text(ctx, text.data(), offs.data(), nreads, 0, &Rd)) != PBA_OK) die(ctx, "reads", st);
    if ((st = pba_index_build(ctx, T, 0, pba_mask_from_pattern(argv[2]), PBA_INDEX_ALL, &ix)) != PBA_OK) die(ctx, "index", st);   // locator.cpp:51-66
    std::vector<pba_loc_row> rows(nreads ? nreads : 1);
    pba_loc_stats stats;
    // locator.cpp:68-92: 50 probe offsets, reads of >= 500 bases, seq_aligner<40000, 6000>
    if ((st = pba_locate(ctx, ix, T, 0, Rd, R, 50, 500, 40000, 6000, PBA_KERNEL_AUTO, rows.data(), &stats)) != PBA_OK) die(ctx, "locate", st);
    for (uint32_t i = 0; i < nreads; ++i)
        if (rows[i].found)                                                                                           // locator.cpp:84-86
            printf("%d\t%d\t%d\t%d\t%d\n", rows[i].nseq, rows[i].pos, rows[i].cost, rows[i].seglen,
                   (long long)contig.size() - rows[i].pos >= rows[i].seglen ? rows[i].diag_cost : -1);
    fprintf(stderr, "totally %lld sequences processed\n", (long long)stats.n_reads_kept);
    pba_index_destroy(ix);
    pba_seqs_destroy(Rd);
    pba_seqs_destroy(T);
    pba_ctx_destroy(ctx);
    return EXIT_SUCCESS;
}
