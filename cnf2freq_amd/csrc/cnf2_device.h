// cnf2_device.h -- structures shared by the HIP kernels (cnf2_kernels.hip) and the C-ABI
// implementation (cnf2_capi.hip).  Device memory layout is described in DESIGN.md.
#ifndef CNF2_DEVICE_H
#define CNF2_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cnf2_window.h"

namespace cnf2 {

#define CNF2_BLOCK 256
#define CNF2_WAVES_PER_BLOCK (CNF2_BLOCK / 64)
#define CNF2_MINFACTOR_F (-1e15f)  /* settings.h:29 */
#define CNF2_IGNORED_D (-1e30)     /* cnF2freq.cpp:5378 */

enum { KP_NO_DOSAGE = 1, KP_RAW_DOSAGE = 2, KP_NO_TIES = 4,
       KP_ACC_TABLE = 8,    // accumulate: the table-form kernel does every window
       KP_ACC_ATTOP = 16,
       KP_ACC_LANES = 32,
       KP_FLUSH_TINY = 64 };  // general sweep kernel: a state under 1e-300 of its vector is set to 0 before the emission (cnF2freq.cpp:1607-1611) // accumulate: path form with one lane per path (acc_paths_kernel) instead of the tile form // accumulate: the batch holds windows whose root is the top of its lines (table form)

// One unit of sequential work: an analysed individual on one chromosome
// (the body of the loops at cnF2freq.cpp:5283 and 5294).
struct Job {
    int32_t ind;     // index into windows[] / output rows (local to the call)
    int32_t first;   // chromstarts[c]
    int32_t last;    // chromstarts[c+1] - 1
    int32_t chrom;
};

// Four jobs of the same chromosome swept by one wavefront (fb_packed_kernel).
struct PackedJob {
    int32_t ind[4];
    int32_t first, last, chrom;
    int32_t homleaf;   // all four jobs: the grandparents are present and homozygous everywhere too (HOMLEAF)
};

struct KernelParams {
    // inputs, resident in HBM
    const Window*  windows;    // [n_ind]
    const Job*     jobs;       // [n_jobs]
    const uint8_t* allele8;    // [n_rows][n_markers]  a0 | a1 << 4
    const double2* sure;       // [n_rows][n_markers]
    const double*  hw;         // [n_rows][n_markers]
    const double2* rho;        // [n_markers] recombination fraction of gap m -> m+1 for
                               //             genrec[0] (.x) and genrec[1] (.y); 0 when dist <= 0
    const double2* tq;         // [n_markers] r / (1 - r) of the same gaps (fast kernel's scaled butterflies)
    const double*  chrom_logk; // [n_chrom] sum over the chromosome's gaps of 4 log(1-r0) + 2 log(1-r1)
    const PackedJob* pjobs;    // [n_pjobs] (fb_packed_kernel)
    int            n_pjobs;
    int            n_jobs;
    int            n_markers;
    int            n_chrom;
    uint32_t       flags;
    // workspace: alpha-minus spill, one slot per resident wave, [len][8][64] doubles
    double*        spill;
    size_t         spill_stride;   // doubles per wave slot
    // outputs
    double*        factors;    // [n_ind][n_chrom][8]
    double*        loglik;     // [n_ind][n_chrom]
    double*        dosage;     // [n_ind][n_markers][3]
    // debug store (fwbw_store): reference layout for ONE job
    double*        dbg_fwbw;     // [8][len][3][64]
    double*        dbg_factors;  // [8][len][3]
    // accumulate mode (STOREW instantiations): posterior weights wg(s, g) = exp(scales - factor) alphaminus beta of
    // every marker of every job of the launch, [job][wstride markers][4][64 lanes][2] in the sweep's own
    // lane / register layout (lane = chain << 3 | l, registers 2k, 2k+1)
    double*        wbuf;
    size_t         wstride;      // markers per job slot
    // which kernel / producer specialisation swept a job (tests): [n_ind][n_chrom], 0-3 = fast kernel with that `hom`
    // class, 32 | homleaf = packed kernel, 64 = general kernel; NULL = not recorded
    int32_t*       path_log;
    // fast kernel: the likelihoods leave the sweep as mantissa (in factors / loglik) and binary exponent (here); the
    // logarithms are taken by likelihood_logs_kernel right after it, so that no transcendental sits in the sweep kernel.
    // [n_ind][n_chrom][8] and [n_ind][n_chrom]; CNF2_LEXP_* mark chains / jobs without a likelihood
    int32_t*       fexp;
    int32_t*       lexp;
    // [4] or NULL: shader-clock and wall-clock ticks of block 0's first wave (launch_fb_fast: the bench's effective clock)
    unsigned long long* clock_out;
    // fast kernel: the launch's job counter (zeroed before the launch), from which its waves take their jobs one at a
    // time -- whichever waves are resident share the list evenly, whatever the jobs' lengths and whenever their blocks
    // got onto the machine; NULL = wave w sweeps jobs w, w + waves, ...
    int*           job_next;
};
#define CNF2_LEXP_IGNORED (-2147483647 - 1)   /* shift mode not analysed: CNF2_IGNORED_D */
#define CNF2_LEXP_DEAD    (-2147483647)       /* no likelihood left: CNF2_MINFACTOR_F */
enum { PATH_TIED = 16, PATH_PACKED = 32, PATH_GENERAL = 64 };

// Inputs of the batched HOT LOOP 2 kernel (acc_rows_kernel): the weights a STOREW sweep left for `n_jobs` jobs and
// where the per-record accumulators live.  After every locus the reference scales homozyg, then moveinfprobs /
// movehaplos add the thread-private sums to the window members (cnF2freq.cpp:5876-5902, 3577-3616): done here with
// f64 atomics, one wave per (job, marker).
struct AccParams {
    KernelParams   kp;           // windows (offset to ind_begin), jobs (offset to the batch), rows, wbuf, loglik, factors
    int            n_jobs;       // jobs in this batch
    int            max_len;      // longest chromosome of the batch (grid.y)
    uint32_t       flags;        // KP_NO_TIES, KP_ACC_TABLE, KP_ACC_ATTOP
    const int32_t* slot_rec;     // [n_ind][7] record per window slot, -1 none (offset like windows)
    const int32_t* desc;         // [n_rec] individ::descendants
    const uint8_t* rec_empty;    // [n_rec]
    double*        acc_inf;      // [n_rec][n_markers][2][2]
    double*        acc_hb;       // [n_rec][n_markers] haplobase
    double*        acc_hc;       // [n_rec][n_markers] haplocount
    double*        acc_hz;       // [n_ind][n_markers][2] homozyg of the analysed individual (offset like windows)
    double*        part;         // CNF2_DETERMINISTIC: [n_ind][n_markers][7][6] per-job rows (offset like windows), else null
};
void launch_acc_rows(const AccParams& q, hipStream_t stream);
void launch_acc_gather(const AccParams& q, const int32_t* rec_start, const int32_t* list, int n_rec, hipStream_t stream);
void launch_fb_fast_tied(const KernelParams& p, int grid, hipStream_t stream);
void launch_fb_fast_tied_w(const KernelParams& p, int grid, hipStream_t stream);

// Inputs of the per-iteration update kernels (cnf2_update.h): what doit does after the sweep of chromosome `chrom`
// (cnF2freq.cpp:6232-6392): processinfprobs for the markers of that chromosome, updatehaploweights for every marker
// of the chromosomes swept so far in this iteration.
struct UpdateParams {
    int            n_rec, n_markers, n_chrom, chrom, first, last;   // first / last marker of `chrom`; n_rec = records this pass updates
    const int32_t* rec_list;      // device [n_rec] the records this pass updates (ascending), or null = records 0 .. n_rec - 1
    int            chromstarts_host_upto;                           // chromstarts[chrom + 1]
    const int32_t* chromstarts;   // device [n_chrom + 1]
    const int32_t* row_of;        // [n_rec]
    const uint8_t* rec_empty;     // [n_rec]
    const uint8_t* has_prior;     // [n_rec] priormarkerdata exists (the individual was genotyped)
    const int32_t* children;      // [n_rec] analysed children (cnF2freq.cpp:5248-5260)
    const int32_t* descendants;   // [n_rec]
    uint8_t*       allele8;       // rows, updated in place
    double2*       sure;
    double*        hw;
    const uint8_t* prior_allele8; // rows as they were read (cnF2freq.cpp:6664-6665)
    const double2* prior_sure;
    double*        acc_inf;       // [n_rec][n_markers][2][2] cleared as it is consumed
    double*        acc_hb;        // [n_rec][n_markers] rewritten as the reference leaves it
    double*        acc_hc;
    uint8_t*       anyinfo;       // [n_rec][n_chrom] scratch
    double*        fw;            // [n_rec][n_markers][2] scratch
    double*        ratio;         // [n_rec][n_markers] scratch
    double         relhaplo;      // 0.5 on this path (cnF2freq.cpp:2496)
    double         scalefactor, entropyfactor;
    int*           hits;          // device counter
    unsigned long long* flow_next;   // [32]: [0..1] item counters of the persistent flow kernels; null = one thread per element
    unsigned long long* stats;       // [24] diagnostics of the flow kernels (cnf2_update_stats), may be null
    void*          todo;          // flows the scouts set aside for the finish kernels (24 bytes each)
    void*          todo2, *todo3; // two more lists of the same size (the packed lists of the guided kernels)
    unsigned long long* todo_counts;  // [todo_cap / 256 + 2] scratch of the packing
    size_t         todo_cap;      // flows per chunk of a scout = entries of todo
    int            mirror;        // certainties: run one flow per side where both values have evidence, the other is its mirror image
    int            scout_passes;  // 2 = a short first scout pass and a second for the flows still going; 1 = one pass (A/B)
    int            literal_finish; // the flows set aside take one literal bisection step per round (rounds 3 / 4) instead of the guided bisection (A/B, cross-check)
    double*        flow_out;      // [n_rec][markers of chrom][2][2] new probabilities from the certainty flows
};
void launch_update_pass(const UpdateParams& u, hipStream_t stream);
void launch_clock_probe(int n_cu, int iters, double* sink, hipStream_t stream);
void launch_copy_rows_f64(const double* src, size_t src_stride, const int32_t* src_idx, double* dst, size_t dst_stride,
                          const int32_t* dst_idx, int n, size_t elems, hipStream_t stream);
void launch_copy_rows_u8(const uint8_t* src, size_t src_stride, const int32_t* src_idx, uint8_t* dst, size_t dst_stride,
                         const int32_t* dst_idx, int n, size_t elems, hipStream_t stream);
void launch_okvals(const KernelParams& p, int n_windows, uint8_t* out, hipStream_t stream);
void launch_addvariance_batch(const KernelParams& p, int n_windows, double* out, hipStream_t stream);
void launch_variance_exact(const KernelParams& p, const int32_t* markers, int n, double* out, hipStream_t stream);
void launch_variance_closed(const KernelParams& p, int n_windows, double* out, hipStream_t stream);
// rows = false: the instantiation that forms no per-locus rows (p.dosage is not written; windows with tie groups can take
// it too: the posterior weights do not see the tie rule)
void launch_fb_fast_w(const KernelParams& p, int grid, hipStream_t stream, bool rows = true);
void launch_fb_w(const KernelParams& p, int grid, hipStream_t stream);
void launch_fb_fast_ab(const KernelParams& p, int grid, hipStream_t stream);
void launch_fb_ab(const KernelParams& p, int grid, hipStream_t stream);
// Inputs of the batched turn scan (turn_rows_kernel): what a turn-scan sweep (STOREW == 2) left in kp.wbuf
// Row of the turn-scan mode's batch buffer (doubles per job and marker): A = alphaminus e [4][64 lanes][2], B = beta
// likewise, then per shift mode s the scales that make them absolute as mantissa and binary exponent
// [8][4] = (mA, eA, mB, eB): absolute A_s = A * mA * 2^eA, absolute B_s = B * mB * 2^eB.
#define CNF2_TURN_ROW 1056
struct TurnParams {
    KernelParams kp;
    int          n_jobs, max_len;
    int          scaled_transitions;   // the batch came from the fast kernel (log-likelihoods include chrom_logk)
    int          valu_form;            // the dot products on the vector ALU instead of the matrix cores (cross-check, A/B)
    double*      rawervals;   // [n_ind][n_markers][128][8] or NULL
    double*      turn_lse;    // [n_ind][n_markers][128] or NULL
};
void launch_turn_rows(const TurnParams& q, hipStream_t stream);

// Inputs of the stage-2 parity kernels: the reference-layout store of ONE individual x chromosome.
struct Stage2Params {
    KernelParams  kp;          // windows (already offset to the individual), rows, n_markers
    const double* fwbw;        // [8][len][3][64]
    const double* fwbwfactors; // [8][len][3]
    const double* factors;     // [8]
    const double* loglik;      // [1]
    int           first, len;
};
void launch_locked_query(const Stage2Params& q, int marker, double* out, hipStream_t stream);
void launch_turn_scan(const Stage2Params& q, int marker, double* out, hipStream_t stream);
void launch_turn_scan_rows(const Stage2Params& q, double* out, hipStream_t stream);
void launch_state_rows(const Stage2Params& q, uint32_t flags, double* out, hipStream_t stream);
void launch_haplos_rows(const Stage2Params& q, uint32_t flags, double* out, hipStream_t stream);
void launch_infprobs(const Stage2Params& q, int marker, uint32_t flags, double* out, hipStream_t stream);
void launch_infprobs_rows(const Stage2Params& q, uint32_t flags, double* out, hipStream_t stream);
void launch_addvariance(const KernelParams& p, int first, int len, double* out, hipStream_t stream);
void launch_fb(const KernelParams& p, int grid, bool debug_store, hipStream_t stream);
void launch_fb_fast(const KernelParams& p, int grid, bool half_spill, hipStream_t stream);
void launch_fb_fast_xpose(const KernelParams& p, int grid, hipStream_t stream);
void launch_fb_packed(const KernelParams& p, int grid, hipStream_t stream);
void launch_row_flags(const uint8_t* allele8, const double2* sure, int n_rows, int n_markers, uint8_t* flags,
                      hipStream_t stream);
int  fb_fast_blocks_per_cu();
void launch_emission(const KernelParams& p, int ind, int marker, double* out, hipStream_t stream);
void launch_emission_paths(const KernelParams& p, int marker, double* out, hipStream_t stream);
void launch_xor_selftest(double* out, hipStream_t stream);
int  fb_blocks_per_cu();

} // namespace cnf2
#endif
