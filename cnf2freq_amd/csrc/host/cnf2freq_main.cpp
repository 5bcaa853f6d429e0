// cnf2freq_main.cpp -- drop-in command line for the PlantImpute invocation of the reference
// (demo.sh:37):
//   cnF2freq --mapfile F --pedfile F --genfile F --output F --count N [--limit n] [--capmarker n] [--tmppath d]
//            [--deserialize F] [--gpus N]
// Flag names and semantics follow main() (cnF2freq.cpp:7954-7972, 8083-8195): postmarkerdata, an optional
// --deserialize of an earlier dump, then --count rounds of which the first only dumps and every later one runs a
// haplotyping sweep (doit) before its dump.  Rows of the last round go to --output, earlier ones to stdout; every
// round's dump, and the FIRST PASS / SKEWNESS PASS lines of the last one, go to --output (cnF2freq.cpp:8166-8186,
// 6259, 6351).  Per chromosome and analysed individual the output is "name:chrom", one row per marker of the
// allele-2 dosage posterior ("%.5lf" tab separated, genotypereporter, cnF2freq.cpp:3499-3538) and a blank line
// (cnF2freq.cpp:6183-6188).
//
// --gpus N (not a flag of the reference, whose MPI code is dead: cnF2freq.cpp:5297-5299, 6245-6254): the process reads the
// files, forks N ranks -- before anything has touched a GPU --, rank r takes GPU r, its block of the analysed individuals and
// the records it owns (cnf2_partition.h), the ranks exchange through shared memory (cnf2_shm_transport.h), and rank 0
// writes ONE output in the order of a single-GPU run: the other ranks' rows reach it through files in --tmppath.
//
// Everything numeric goes through the C ABI of include/cnf2hip.h (host bookkeeping in cnf2_engine.cpp); this program
// has no compute path of its own and fails if no GPU is present.  Out of scope: the toulbar2
// bridge and the haplotype inversions it decides, all non-PlantImpute readers.
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dirent.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "cnf2_engine.h"
#include "cnf2_readers.h"
#include "cnf2_rccl_transport.h"
#include "cnf2_shm_transport.h"
#include "cnf2hip.h"

using namespace cnf2host;

struct Options {
    std::string mapfile, pedfile, genfile, output, deserialize, tmppath = ".";
    int         count = 3;          // cnF2freq.cpp:7961
    int         limit = 1000000;    // INDCOUNT (settings.h:9)
    int         capmarker = 0;
    bool        quiet = false;
    bool        merge_modes = true;   // CNF2_MERGE_MODES: exact, faster for F2-type pedigrees (include/cnf2hip.h)
    bool        normalise = false;    // rows as the reporter leaves them: raw class sums (cnF2freq.cpp:3523 has the
                                      // division by probsum commented out); --normalise divides each row by its sum
    bool        preprocess = true;    // --no-preprocess: skip postmarkerdata      } parity aids, not reference modes:
    bool        update = true;        // --no-update: sweeps without the updates    } every round repeats the same sweep
    bool        dump_all = true;      // --dump-last-only: large runs
    bool        rows_all = true;      // --rows-last-only: large runs (rows of non-final rounds are not formatted)
    bool        parse_only = false;   // print the parsed tables and stop (no GPU needed; used by tests)
    int         gpus = 1;             // --gpus N: N ranks, one GPU each
    bool        single_device = false;   // --single-device: every rank on GPU 0 (rehearsal of --gpus N on a one-GPU box)
    std::string transport;               // --transport rccl|shm: what the ranks exchange through (default: rccl when every rank has
                                         // a GPU of its own, shm -- staging through a shared region on the host -- with --single-device)
    bool        rccl_selftest = false;   // --rccl-selftest: the RCCL transport's collectives with a world of one on GPU 0, then stop
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
        else if (a == "--deserialize") o.deserialize = val();
        else if (a == "--tmppath") o.tmppath = val();
        else if (a == "--count") o.count = atoi(val().c_str());
        else if (a == "--limit") o.limit = atoi(val().c_str());
        else if (a == "--capmarker") o.capmarker = atoi(val().c_str());
        else if (a == "--quiet") o.quiet = true;
        else if (a == "--no-merge-modes") o.merge_modes = false;
        else if (a == "--normalise") o.normalise = true;
        else if (a == "--no-preprocess") o.preprocess = false;
        else if (a == "--no-update") o.update = false;
        else if (a == "--dump-last-only") o.dump_all = false;
        else if (a == "--rows-last-only") o.rows_all = false;
        else if (a == "--parse-only") o.parse_only = true;
        else if (a == "--gpus") o.gpus = atoi(val().c_str());
        else if (a == "--single-device") o.single_device = true;
        else if (a == "--transport") o.transport = val();
        else if (a == "--rccl-selftest") o.rccl_selftest = true;
        else {
            fprintf(stderr, "unsupported option %s (this build covers the PlantImpute path only)\n", a.c_str());
            return false;
        }
    }
    return true;
}

// One rank of a run: GPU `rank` (or 0), the whole pedigree, its block of the analysed individuals.  rank 0 writes the output.
static int run_rank(const Options& opt, Pedigree& P, int rank, int world, ShmRegion* region)
{
    if (rank > 0) {
        // what is identical on every rank (progress lines, pass lines, dumps) is written by rank 0 only
        if (!freopen("/dev/null", "w", stdout)) return 3;
    }
    cnf2_ctx* ctx = nullptr;
    const int device = (world > 1 && !opt.single_device) ? rank : 0;
    if (cnf2_ctx_create(device, &ctx) != CNF2_OK) {
        fprintf(stderr, "cnf2_ctx_create(device %d): %s%s\n", device, cnf2_last_error(nullptr),
                world > 1 ? " (--gpus N needs N GPUs; --single-device rehearses it on one)" : "");
        if (world > 1) return 3;
        abort();
    }
    EngineOptions eo;
    eo.quiet = opt.quiet;
    eo.merge_modes = opt.merge_modes;
    eo.normalise = opt.normalise;
    eo.update = opt.update;
    eo.dump_all = opt.dump_all;
    if (world > 1) {
        eo.spool_dir = opt.tmppath;
        eo.spool_tag = "run" + std::to_string((long)getppid());       // every rank is a child of the process that read the files
    }
    ShmTransport  T;
    RcclTransport TR;
    const bool    use_rccl = world > 1 && (opt.transport.empty() ? !opt.single_device : opt.transport == "rccl");
    // any failure below the C ABI ends the run the way the reference ends on every failure (cnF2freq.cpp:21-25)
    try {
    Engine E(P, ctx, eo);
    E.upload();
    if (world > 1) {
        T.R = region;
        T.ctx = ctx;
        T.rank = rank;
        if (use_rccl) {
            if (TR.init(region, ctx, rank, world) != 0) return 4;
            E.set_partition(rank, world, RcclTransport::call, &TR);
        } else
            E.set_partition(rank, world, ShmTransport::call, &T);
        E.set_root_threads(host_threads() * world);          // rank 0 infers the genotypes for all while the others wait
    }
    if (opt.preprocess) E.postmarkerdata(opt.limit);                 // cnF2freq.cpp:8083-8085
    if (!opt.deserialize.empty() && !E.deserialize(opt.deserialize.c_str())) {
        fprintf(stderr, "cannot open %s\n", opt.deserialize.c_str());
        abort();
    }
    if (world > 1) {
        const Partition& Q = E.partition();
        if (rank == 0) fprintf(stderr, "transport: %s\n", use_rccl ? "RCCL in place on the exchange buffer" : "shared memory on the host");
        if (rank == 0) {
            size_t xb[4];
            E.exchange_bytes(xb);
            fprintf(stderr, "%d ranks: blocks", world);
            for (int r = 0; r < world; r++) fprintf(stderr, " [%d, %d)", Q.bounds[r], Q.bounds[r + 1]);
            fprintf(stderr, "; %zu shared records, %zu bytes exchanged per iteration\n", Q.n_shared, xb[3]);
        }
    }

    FILE* out = stdout;
    if (rank > 0) out = fopen("/dev/null", "w");
    else if (!opt.output.empty()) out = fopen(opt.output.c_str(), "w");
    if (!out) { fprintf(stderr, "cannot open output\n"); abort(); }

    for (int it = 0; it < opt.count; it++) {
        const bool early = it < 1;                       // cnF2freq.cpp:8131
        if (!early) {
            E.set_print_rows(opt.rows_all || it == opt.count - 1);
            E.iteration((it == opt.count - 1) ? out : stdout);
        }
        fflush(stdout);
        fflush(out);
        if (opt.dump_all || it == opt.count - 1) {
            const auto t0 = std::chrono::steady_clock::now();
            E.dump(out, opt.limit);                                              // (gathers the ranks' rows: every rank calls it)
            if (getenv("CNF2_TIMING") && rank == 0)
                fprintf(stderr, "  [write] dump of round %-3d                  %.3f s\n", it,
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        fflush(stdout);
        fflush(out);
    }
    if (out != stdout) fclose(out);
    } catch (const EngineError& e) {
        fprintf(stderr, "%s\n", e.what());
        if (world > 1) return 4;                          // the parent stops the other ranks and aborts
        abort();
    }
    cnf2_ctx_destroy(ctx);
    return 0;
}

// --rccl-selftest: the RCCL transport with a world of one on GPU 0 -- the communicator's set-up through the shared region, then
// reduce-scatter, all-gather, the hit-counter sum, a barrier and the host broadcast on the context's exchange buffer, through
// the same entry the engine calls.  (Two ranks need two GPUs: RCCL refuses two ranks on one device.)
static int rccl_selftest()
{
    ShmRegion* R = shm_region_create(1, (size_t)1 << 20);
    cnf2_ctx*  ctx = nullptr;
    if (!R || cnf2_ctx_create(0, &ctx) != CNF2_OK) {
        fprintf(stderr, "rccl selftest: no region / no device: %s\n", cnf2_last_error(nullptr));
        return 3;
    }
    RcclTransport T;
    if (T.init(R, ctx, 0, 1) != 0) return 4;
    const size_t n = 100000;
    void*        buf = nullptr;
    if (cnf2_exchange_buffer(ctx, n * sizeof(double), &buf) != CNF2_OK) return 5;
    std::vector<double> v(n), w(n, 0.0);
    for (size_t i = 0; i < n; i++) v[i] = 0.5 + (double)i * 1e-3;
    int bad = 0;
    bad |= cnf2_exchange_write(ctx, 0, v.data(), n * sizeof(double)) != CNF2_OK;
    bad |= RcclTransport::call(&T, X_SUM_SEGMENTS, buf, n, n) != 0;              // one segment: the sum of one rank's values
    bad |= RcclTransport::call(&T, X_GATHER_SEGMENTS, buf, n * sizeof(double), n * sizeof(double)) != 0;
    bad |= cnf2_exchange_read(ctx, 0, w.data(), n * sizeof(double)) != CNF2_OK;
    for (size_t i = 0; i < n; i++) bad |= w[i] != v[i];
    int32_t h[2] = {7, 11};
    bad |= RcclTransport::call(&T, X_SUM_HITS, h, 2, 2) != 0 || h[0] != 7 || h[1] != 11;
    bad |= RcclTransport::call(&T, X_BARRIER, nullptr, 0, 0) != 0;
    unsigned char hb[300];
    for (int i = 0; i < 300; i++) hb[i] = (unsigned char)(i * 7);
    bad |= RcclTransport::call(&T, X_BCAST_HOST, hb, 300, 0) != 0 || hb[299] != (unsigned char)(299 * 7);
    T.finish();
    cnf2_ctx_destroy(ctx);
    printf("rccl selftest: %s (world 1, %zu doubles through ncclReduceScatter and ncclAllGather in place)\n", bad ? "FAILED" : "ok", n);
    return bad ? 1 : 0;
}

int main(int argc, char** argv)
{
    Options opt;
    if (!parse(argc, argv, opt)) return 2;
    if (opt.rccl_selftest) return rccl_selftest();
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
        const auto t0 = std::chrono::steady_clock::now();
        if (!read_alpha_gen(f, P)) { fprintf(stderr, "cannot read genotypes\n"); abort(); }
        fclose(f);
        if (getenv("CNF2_TIMING"))
            fprintf(stderr, "  [read] genotype file                     %.3f s\n",
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    // after ALL files are read: the genotype reader is token based, so capping earlier would misalign
    // it (the reference's own notifier runs before the map exists, cnF2freq.cpp:7965-7969)
    if (opt.capmarker > 0) cap_markers(P, opt.capmarker);
    if (!opt.quiet)
        for (auto& l : P.log) printf("%s\n", l.c_str());

    if (opt.parse_only) {
        Tables T;
        build_tables(P, T);
        const int M = P.n_markers();
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

    if (opt.transport != "" && opt.transport != "rccl" && opt.transport != "shm") {
        fprintf(stderr, "--transport must be rccl or shm\n");
        return 2;
    }
    if (opt.gpus < 1 || opt.gpus > 64) {
        fprintf(stderr, "--gpus must be between 1 and 64\n");
        return 2;
    }
    // main() trims dous only after postmarkerdata (cnF2freq.cpp:8083, 8124); the analysed list is fixed at upload here,
    // and postmarkerdata does not read it
    if ((int)P.dous.size() > opt.limit) P.dous.resize(opt.limit);
    if (opt.gpus == 1) return run_rank(opt, P, 0, 1, nullptr);

    // N ranks: the region and the fork come before any HIP call of this process
    ShmRegion* region = shm_region_create(opt.gpus, (size_t)64 << 20);
    if (!region) {
        fprintf(stderr, "cannot map the shared region of %d ranks\n", opt.gpus);
        abort();
    }
    // the ranks share this process's CPUs (affinity mask and cgroup quota): each gets its N-th for its host loops
    set_host_threads(std::max(1, host_threads() / opt.gpus));
    fflush(stdout);
    fflush(stderr);
    static std::vector<pid_t> kids;             // static: the signal handler below ends them
    const pid_t parent = getpid();
    auto stop_ranks = [](int sig) {
        for (pid_t k : kids) kill(k, SIGKILL);
        _exit(128 + sig);
    };
    signal(SIGINT, stop_ranks);
    signal(SIGTERM, stop_ranks);
    for (int r = 0; r < opt.gpus; r++) {
        const pid_t pid = fork();
        if (pid < 0) {
            perror("fork");
            for (pid_t k : kids) kill(k, SIGKILL);
            abort();
        }
        if (pid == 0) {
            // a rank must not outlive the run: if the parent is killed the ranks would wait at a barrier for ever, holding
            // their GPU memory
            prctl(PR_SET_PDEATHSIG, SIGKILL);
            if (getppid() != parent) _exit(5);             // the parent died between fork and prctl
            signal(SIGINT, SIG_DFL);
            signal(SIGTERM, SIG_DFL);
            const int rc = run_rank(opt, P, r, opt.gpus, region);
            fflush(stdout);
            fflush(stderr);
            _exit(rc);
        }
        kids.push_back(pid);
    }
    // a rank that fails would leave the others waiting at a barrier: the first failure ends them all
    int failed = 0;
    for (size_t left = kids.size(); left > 0; left--) {
        int         st = 0;
        const pid_t pid = wait(&st);
        if (pid < 0) break;
        if (!(WIFEXITED(st) && WEXITSTATUS(st) == 0) && !failed) {
            failed = 1;
            fprintf(stderr, "a rank ended abnormally: stopping the others\n");
            for (pid_t k : kids)
                if (k != pid) kill(k, SIGKILL);
        }
    }
    if (failed) {
        // text the ranks had spooled for rank 0 (cnf2_<what>_run<pid>_...): nobody will collect it now
        const std::string tag = "_run" + std::to_string((long)parent) + "_";
        if (DIR* d = opendir(opt.tmppath.c_str())) {
            while (dirent* e = readdir(d)) {
                const std::string name = e->d_name;
                if (name.rfind("cnf2_", 0) == 0 && name.find(tag) != std::string::npos) remove((opt.tmppath + "/" + name).c_str());
            }
            closedir(d);
        }
        abort();                                          // the reference ends every failure this way (cnF2freq.cpp:21-25)
    }
    return 0;
}
