// cnf2freq_main.cpp -- drop-in command line for the PlantImpute invocation of the reference
// (demo.sh:37):
//   cnF2freq --mapfile F --pedfile F --genfile F --output F --count N [--limit n] [--capmarker n] [--tmppath d]
// Flag names and semantics follow main() (cnF2freq.cpp:7954-7972, 8119-8195): --count N runs N-1
// sweeps, iteration 0 only dumps; the last iteration writes to --output, earlier ones to stdout.
// Per chromosome and analysed individual the output is "name:chrom", one row per marker of the
// allele-2 dosage posterior ("%.5lf" tab separated, genotypereporter, cnF2freq.cpp:3499-3538)
// and a blank line (cnF2freq.cpp:6183-6188); then the per-individual dump (cnF2freq.cpp:8157-8192).
//
// Everything numeric goes through the C ABI of include/cnf2hip.h; this program has no compute path
// of its own and fails if no GPU is present.  Out of scope: postmarkerdata,
// the toulbar2 bridge and the per-iteration parameter updates, so iterations do not change the
// haplotype weights (documented in INTEGRATION.md).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "cnf2_readers.h"
#include "cnf2hip.h"

using namespace cnf2host;

struct Options {
    std::string mapfile, pedfile, genfile, output, tmppath = ".";
    int         count = 3;          // cnF2freq.cpp:7961
    int         limit = 1000000;    // INDCOUNT (settings.h:9)
    int         capmarker = 0;
    bool        quiet = false;
    bool        merge_modes = true;   // CNF2_MERGE_MODES: exact, faster for F2-type pedigrees (include/cnf2hip.h)
    bool        normalise = false;    // rows as the reporter leaves them: raw class sums (cnF2freq.cpp:3523 has the
                                      // division by probsum commented out); --normalise divides each row by its sum
    bool        parse_only = false;   // print the parsed tables and stop (no GPU needed; used by tests)
};

static bool parse(int argc, char** argv, Options& o)
{
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i], v;
        bool        has = false;
        size_t      eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) {
            v   = a.substr(eq + 1);
            a   = a.substr(0, eq);
            has = true;
        }
        auto val = [&]() -> std::string {
            if (has) return v;
            if (i + 1 >= argc) {
                fprintf(stderr, "missing value for %s\n", a.c_str());
                exit(2);
            }
            return argv[++i];
        };
        if (a == "--mapfile") o.mapfile = val();
        else if (a == "--pedfile") o.pedfile = val();
        else if (a == "--genfile") o.genfile = val();
        else if (a == "--output") o.output = val();
        else if (a == "--tmppath") o.tmppath = val();
        else if (a == "--count") o.count = atoi(val().c_str());
        else if (a == "--limit") o.limit = atoi(val().c_str());
        else if (a == "--capmarker") o.capmarker = atoi(val().c_str());
        else if (a == "--quiet") o.quiet = true;
        else if (a == "--no-merge-modes") o.merge_modes = false;
        else if (a == "--normalise") o.normalise = true;
        else if (a == "--parse-only") o.parse_only = true;
        else {
            fprintf(stderr, "unsupported option %s (this build covers the PlantImpute path only)\n", a.c_str());
            return false;
        }
    }
    return true;
}

#define CHECK(ctx, call)                                                        \
    do {                                                                        \
        int rc_ = (call);                                                       \
        if (rc_ != CNF2_OK) {                                                   \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, cnf2_last_error(ctx)); \
            abort(); /* the reference aborts on any failure (cnF2freq.cpp:21-25) */ \
        }                                                                       \
    } while (0)

int main(int argc, char** argv)
{
    Options opt;
    if (!parse(argc, argv, opt)) return 2;
    Pedigree P;
    if (!opt.mapfile.empty()) {
        FILE* f = fopen(opt.mapfile.c_str(), "rt");
        fprintf(stderr, "Reading map file %s\n", opt.mapfile.c_str());
        if (!read_alpha_map(f, P)) { fprintf(stderr, "cannot read map\n"); abort(); }
        fclose(f);
    }
    if (!opt.pedfile.empty()) {
        FILE* f = fopen(opt.pedfile.c_str(), "rt");
        fprintf(stderr, "Reading pedigree file %s\n", opt.pedfile.c_str());
        if (!read_alpha_ped(f, P)) { fprintf(stderr, "cannot read pedigree\n"); abort(); }
        fclose(f);
    }
    if (!opt.genfile.empty()) {
        FILE* f = fopen(opt.genfile.c_str(), "rt");
        fprintf(stderr, "Reading genotype file %s\n", opt.genfile.c_str());
        if (!read_alpha_gen(f, P)) { fprintf(stderr, "cannot read genotypes\n"); abort(); }
        fclose(f);
    }
    // after ALL files are read: the genotype reader is token based, so capping earlier would misalign
    // it (the reference's own notifier runs before the map exists, cnF2freq.cpp:7965-7969)
    if (opt.capmarker > 0) cap_markers(P, opt.capmarker);
    if (!opt.quiet)
        for (auto& l : P.log) printf("%s\n", l.c_str());
    if ((int)P.dous.size() > opt.limit) P.dous.resize(opt.limit);     // cnF2freq.cpp:8124

    Tables T;
    build_tables(P, T);
    const int M = P.n_markers(), C = (int)P.chromstarts.size() - 1, N = (int)T.dous.size();

    if (opt.parse_only) {
        printf("markers %d chromstarts", M);
        for (int v : P.chromstarts) printf(" %d", v);
        printf("\nrows %d\n", T.n_rows);
        for (size_t r = 0; r < P.inds.size(); r++) {
            const Individual& I = P.inds[r];
            printf("ind %d %s gen %d empty %d pars %d %d row %d analysed %d :", I.n, I.name.c_str(), I.gen, (int)I.empty,
                   I.pars[0] < 0 ? 0 : P.inds[I.pars[0]].n, I.pars[1] < 0 ? 0 : P.inds[I.pars[1]].n, T.row_of[r],
                   (int)(std::find(T.dous.begin(), T.dous.end(), (int)r) != T.dous.end()));
            for (int m = 0; m < M; m++) printf(" %d%d/%.6g/%.6g", I.allele[m * 2], I.allele[m * 2 + 1], I.sure[m * 2], I.sure[m * 2 + 1]);
            printf("\n");
        }
        return 0;
    }

    cnf2_ctx* ctx = nullptr;
    if (cnf2_ctx_create(0, &ctx) != CNF2_OK) {
        fprintf(stderr, "cnf2_ctx_create: %s\n", cnf2_last_error(nullptr));
        abort();
    }
    CHECK(ctx, cnf2_upload_map(ctx, P.pos.data(), M, P.chromstarts.data(), C, nullptr));
    CHECK(ctx, cnf2_upload_rows(ctx, T.n_rows, T.allele.data(), T.sure.data(), T.hw.data()));
    CHECK(ctx, cnf2_upload_pedigree(ctx, (int)P.inds.size(), T.par.data(), T.empty.data(), T.gen.data(), T.row_of.data(),
                                    T.dous.data(), N));

    FILE* out = stdout;
    if (!opt.output.empty()) out = fopen(opt.output.c_str(), "w");
    if (!out) { fprintf(stderr, "cannot open output\n"); abort(); }

    std::vector<double> factors((size_t)N * C * 8), loglik((size_t)N * C), dosage((size_t)N * M * 3);
    for (int it = 0; it < opt.count; it++) {
        const bool early = it < 1;                       // cnF2freq.cpp:8131
        FILE* dst = (it == opt.count - 1) ? out : stdout;
        if (!early && N > 0) {
            CHECK(ctx, cnf2_sweep(ctx, 0, N, factors.data(), loglik.data(), dosage.data(),
                                  (opt.merge_modes ? CNF2_MERGE_MODES : 0) | (opt.normalise ? 0 : CNF2_RAW_DOSAGE)));
            for (int c = 0; c < C; c++) {
                if (!opt.quiet)
                    for (int j = 0; j < N; j++) {
                        int32_t w[17];
                        CHECK(ctx, cnf2_window_info(ctx, j, w));
                        double mx = -1e15;               // the two printf of cnF2freq.cpp:5399-5401
                        for (int s = 0; s < 8; s++) mx = factors[((size_t)j * C + c) * 8 + s] > mx ? factors[((size_t)j * C + c) * 8 + s] : mx;
                        printf("%d,%03d,%03d: %lf\t%lf %d\n", P.inds[T.dous[j]].n, w[1], w[0], mx, loglik[(size_t)j * C + c],
                               P.inds[T.dous[j]].gen < 2 ? 2 : 8);
                    }
                for (int j = 0; j < N; j++) {            // cnF2freq.cpp:6183-6188
                    fprintf(dst, "%s:%d\n", P.inds[T.dous[j]].name.c_str(), c + 1);
                    const double ll = loglik[(size_t)j * C + c];
                    const bool skipped = (ll != ll) || ll < (double)CNF2_MINFACTOR;     // cnF2freq.cpp:5403
                    if (!skipped)
                        for (int m = P.chromstarts[c]; m < P.chromstarts[c + 1]; m++) {
                            const double* d = &dosage[((size_t)j * M + m) * 3];
                            fprintf(dst, "%.5lf\t%.5lf\t%.5lf\n", d[0], d[1], d[2]);
                        }
                    fprintf(dst, "\n");
                }
            }
        }
        fflush(stdout);
        fflush(dst);
        // per-iteration dump of every individual: always to the --output file, whatever the iteration
        // (fprintf(out, ...), cnF2freq.cpp:8157-8192); only the rows above switch between out and stdout
        for (size_t r = 0; r < P.inds.size(); r++) {
            const Individual& I = P.inds[r];
            if (I.n > opt.limit) continue;
            fprintf(out, "%d %s\n", I.n, I.name.c_str());
            for (int m = 0; m < M; m++) {
                if (I.has_prior)
                    fprintf(out, "%f\t%d\t%d\t\t%f\t%lf %lf %lf\t%d\t%d\t%lf\t%lf\n", I.hw[m], I.allele[m * 2], I.allele[m * 2 + 1], 0.0,
                            I.sure[m * 2], I.sure[m * 2 + 1], 0.5, I.prior_allele[m * 2], I.prior_allele[m * 2 + 1],
                            I.prior_sure[m * 2], I.prior_sure[m * 2 + 1]);
                else
                    fprintf(out, "%f\t%d\t%d\t\t%f\t%lf %lf %lf\n", I.hw[m], I.allele[m * 2], I.allele[m * 2 + 1], 0.0, I.sure[m * 2],
                            I.sure[m * 2 + 1], 0.5);
            }
        }
        fflush(stdout);
        fflush(out);
    }
    if (out != stdout) fclose(out);
    cnf2_ctx_destroy(ctx);
    return 0;
}
