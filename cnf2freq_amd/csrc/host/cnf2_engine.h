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

#include <stddef.h>
#include <stdio.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "cnf2_partition.h"
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
    // multi-process runs: directory through which the ranks' rows reach rank 0, which then writes ONE output in the order of
    // a single-process run (per chromosome the rows of every analysed individual, then the pass lines); empty: every rank
    // writes the rows of its own block to its own output
    std::string spool_dir;
    std::string spool_tag;       // part of the spool files' names that is unique to the run (the executable: its parent's pid)
};

// A call into libcnf2hip.so failed (out of memory, launch error, bad state).  The command line turns it into the
// reference's abort() (cnF2freq.cpp:21-25); libcnf2host.so into an error code and cnf2h_last_error().
struct EngineError : std::runtime_error {
    int code;
    EngineError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// Transport of a multi-process haplotyping run (the reference's reduce calls, cnF2freq.cpp:6245-6254).
// The engine packs what ranks must exchange -- the accumulators and rows of the records their windows SHARE, never the
// [n_rec][M] slabs -- into one device buffer of `world` equal segments and asks the transport for one collective on it:
//   X_SUM_SEGMENTS     buf = device, `count` doubles in segments of `seg`: on return segment `rank` holds the sum over all
//                      ranks of that segment (a reduce-scatter; the other segments are left undefined)
//   X_SUM_HITS         buf = HOST int32[count]: in-place sum over all ranks (the hit counters of an update pass)
//   X_GATHER_SEGMENTS  buf = device, `count` bytes in segments of `seg`: every rank has filled its own segment; on return
//                      all segments are filled on every rank (an all-gather)
//   X_BARRIER          nothing to move: returns when every rank has called it (the rows of the ranks' blocks are spooled to
//                      files and collected by rank 0, EngineOptions::spool_dir)
//   X_BCAST_HOST       buf = HOST, `count` bytes: on return every rank holds rank 0's bytes (the state postmarkerdata leaves,
//                      which rank 0 computes for all)
// Returns 0 on success.
enum { X_SUM_SEGMENTS = 0, X_SUM_HITS = 1, X_GATHER_SEGMENTS = 2, X_BARRIER = 3, X_BCAST_HOST = 4 };
typedef int (*ExchangeFn)(void* user, int op, void* buf, size_t count, size_t seg);

// threads the host loops may use: affinity mask and cgroup CPU quota (cnf2_engine.cpp); set_host_threads(n) fixes the number
// (several ranks of one process group share the CPUs: `cnF2freq --gpus N` gives every rank its N-th), 0 = find out again
int  host_threads();
void set_host_threads(int n);

class Engine {
public:
    Engine(Pedigree& ped, cnf2_ctx* ctx, const EngineOptions& opt);

    // tables -> device; remembers the rows as priors (cnF2freq.cpp:6664-6665).  Call once after the readers.
    void upload();
    // allocates the device buffers of the iterations (of this process's block) now instead of inside the first iteration
    void reserve();
    // wall time of the last iteration by where it went: [0] sweep + accumulators, [1] exchanges, [2] update passes, [3] the rest
    // on the host (bookkeeping, likelihood lines, rows), [4] total; seconds
    const double* last_timing() const { return last_timing_; }
    // cnF2freq.cpp:3190-3412 as main calls it (CORRECTIONINFERENCE set), for individuals numbered below indcount.  In a
    // multi-process run (set_partition called first) rank 0 does it for all -- the host loops of N ranks would share the
    // CPUs of one, i.e. take N times as long each -- and broadcasts the rows, descendant counts and lock positions it
    // leaves (the variances stay on rank 0); set_root_threads(n) lets it use n threads while the others wait
    void postmarkerdata(int indcount);
    void set_root_threads(int n) { root_threads_ = n; }
    // cnF2freq.cpp:7757-7832; returns false if the file cannot be opened
    bool deserialize(const char* path);
    // one doit<false, genotypereporter>(out, true): rows of every chromosome and analysed individual to `out`
    void iteration(FILE* out);
    // cnF2freq.cpp:8157-8192
    void dump(FILE* out, int limit);

    // a sub-range of the analysed individuals [begin, end) of dous for this process's sweeps and rows (single-process use;
    // multi-process runs call set_partition, which sets the rank's block)
    void   set_block(int begin, int end);
    // what the sweep kernels spend on every analysed individual, up to a constant: markers x (1 forward + one backward pass
    // per tie combination).  The shift modes of cnF2freq.cpp:5359, 5378 do not enter: a wavefront carries all 8 modes of
    // its individual whether they are masked or not (SURVEY 8(e)'s sum of M x S_act is the cost model of the CPU path).
    std::vector<double> work_costs();
    // block `rank` of `world` as plan() cuts them (cnF2freq.cpp:5297-5299 deals individuals round-robin; contiguous blocks
    // keep a rank's output rows together)
    void   balanced_block(int rank, int world, int* begin, int* end);
    // blocks, private / shared records and their owners for `world` ranks (the same on every rank: a pure function of the
    // pedigree and the window tables)
    Partition plan(int rank, int world);
    // multi-process runs (cnF2freq.cpp:5297-5299, 6245-6254, 6344-6392): this rank sweeps its block of the plan, `fn` carries
    // the collectives, the rank updates the records it owns.  Rows are printed for the rank's own block.
    void   set_partition(int rank, int world, ExchangeFn fn, void* user);
    const Partition& partition() const { return part_; }
    // bytes of the collectives' buffers per iteration: [0] accumulators (reduce-scatter), [1] rows (all-gather), [2] hits;
    // [3] = payload: what the shared records really occupy in them
    void   exchange_bytes(size_t out[4]) const;
    // update form: CNF2_UPDATE_PLAIN / CNF2_UPDATE_BOTH_FLOWS / CNF2_UPDATE_ONE_SCOUT of cnf2hip.h (0 = the fast form with mirrored
    // certainty flows).  Until this is called the form is 0, or CNF2_UPDATE_BOTH_FLOWS (the bit-exact form) in a deterministic run
    void   set_update_flags(uint32_t f) { update_flags_ = f; update_flags_set_ = true; }
    void   set_deterministic(bool d) { deterministic_ = d; }
    const std::vector<int>& pass_hits() const { return pass_hits_; }       // hitnnn of every chromosome's pass of the last iteration
    void   accumulators(double* haplobase, double* haplocount);            // [inds][M] as the last pass left them
    void   set_update(bool u) { opt.update = u; }
    void   set_print_rows(bool p) { opt.print_rows = p; }
    void   sync_rows();          // host copies of the individuals' rows up to date (multi-process: on every rank, a collective)
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
    long                locked_by_bits_ = 0;  // lockhaplos: (record, chromosome) pairs decided by the reference's own rounding
    long                exact_variances_ = 0; //             entries evaluated for that
    double scalefactor_ = 0.013, entropyfactor_ = 1.0;   // cnF2freq.cpp:3573-3574
    int    oldhits_ = 0, oldhits2_ = 0, last_hits_ = 0;
    bool   rows_stale_ = false;   // the device rows were updated since they were last copied to the host
    int    block_begin_ = 0, block_end_ = -1;     // analysed individuals of this rank (-1: all)
    ExchangeFn exchange_ = nullptr;
    void*      exchange_user_ = nullptr;
    Partition  part_;
    bool       rows_partial_ = false;    // multi-process: other ranks hold newer rows of their private records
    bool       deterministic_ = false;
    uint32_t   update_flags_ = 0;
    bool       update_flags_set_ = false;
    void       exchange(int op, void* buf, size_t count, size_t seg, const char* what);
    void       gather_private_rows();
    void       postmarkerdata_local(int indcount);
    void       broadcast_state();
    int        root_threads_ = 0;
    std::vector<int> pass_hits_;
    int    iteration_no_ = 0;              // iterations run so far (names of the spooled row files)
    double last_timing_[5] = {0, 0, 0, 0, 0};
};

}  // namespace cnf2host
#endif
