// cnf2_engine.h -- host side of a drop-in `cnF2freq` run above the C ABI (include/cnf2hip.h): the parts of main()
// that surround the sweep, in the reference's own order (cnF2freq.cpp:8083-8192):
//   postmarkerdata   genotype inference from relatives (fixkid, fixparents), descendant counts, per-marker variances,
//                    haplotype locking                                   cnF2freq.cpp:3190-3412, 1392-1558, 3045-3097
//   deserialize      re-load a previous run's dump                                        cnF2freq.cpp:7757-7832
//   iteration        doit<false, genotypereporter>: sweep + HOT LOOP 2 accumulators on the GPU, rows, then per
//                    chromosome the parameter updates (on the GPU) and the step-size control   cnF2freq.cpp:5189-6410
//   dump             the per-iteration state of every individual                          cnF2freq.cpp:8157-8192
// Everything numeric runs through libcnf2hip.so; this file holds bookkeeping and I/O only.  Out of scope, as in
// the toulbar2 bridge (DOTOULBAR) and with it every haplotype inversion (negshift stays 0),
// map-distance re-estimation (DOREMAPDISTANCES is off in the reference build).
#ifndef CNF2_ENGINE_H
#define CNF2_ENGINE_H

#include <stdio.h>

#include <string>
#include <vector>

#include "cnf2_readers.h"
#include "cnf2hip.h"

namespace cnf2host {

struct EngineOptions {
    bool quiet = false;
    bool merge_modes = true;    // CNF2_MERGE_MODES for sweeps that need no accumulators
    bool normalise = false;     // rows divided by their sum (older reporter); default: raw class sums (cnF2freq.cpp:3523)
    bool update = true;         // false: iterations only sweep and print (parity aid, not a reference mode)
    bool dump_all = true;       // false: only the last iteration dumps (large runs; the reference always dumps)
    bool print_rows = true;     // false: rows of this iteration are not formatted at all (large runs, non-final iterations)
};

class Engine {
public:
    Engine(Pedigree& ped, cnf2_ctx* ctx, const EngineOptions& opt);

    // tables -> device; remembers the rows as priors (cnF2freq.cpp:6664-6665).  Call once after the readers.
    void upload();
    // cnF2freq.cpp:3190-3412 as main calls it (CORRECTIONINFERENCE set), for individuals numbered below indcount
    void postmarkerdata(int indcount);
    // cnF2freq.cpp:7757-7832; returns false if the file cannot be opened
    bool deserialize(const char* path);
    // one doit<false, genotypereporter>(out, true): rows of every chromosome and analysed individual to `out`
    void iteration(FILE* out);
    // cnF2freq.cpp:8157-8192
    void dump(FILE* out, int limit);

    void   set_update(bool u) { opt.update = u; }
    void   set_print_rows(bool p) { opt.print_rows = p; }
    void   sync_rows() { if (rows_stale_) pull_rows(); }   // host copies of the individuals' rows up to date
    double scalefactor() const { return scalefactor_; }
    int    last_hits() const { return last_hits_; }
    const std::vector<int>& descendants() const { return descendants_; }
    const std::vector<int>& children() const { return children_; }
    const std::vector<double>& variances() const { return variances_; }   // [inds][M]

private:
    void push_rows();            // host individuals -> device rows
    void pull_rows();            // device rows -> host individuals
    void check(int rc, const char* what);

    Pedigree&     P;
    cnf2_ctx*     ctx;
    EngineOptions opt;
    Tables        T;
    int           M = 0, C = 0, N = 0;
    std::vector<int>    descendants_, children_;
    std::vector<double> variances_;
    std::vector<int>    lockstart_;          // [inds][C]
    double scalefactor_ = 0.013, entropyfactor_ = 1.0;   // cnF2freq.cpp:3573-3574
    int    oldhits_ = 0, oldhits2_ = 0, last_hits_ = 0;
    bool   rows_stale_ = false;   // the device rows were updated since they were last copied to the host
};

}  // namespace cnf2host
#endif
