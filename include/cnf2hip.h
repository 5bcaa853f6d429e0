/*
 * cnf2hip.h -- C ABI of libcnf2hip.so, the MI355X (gfx950) implementation of
 * cnF2freq's per-individual HMM forward-backward sweep.
 *
 * The reference has no FFI; its seam for this path is the in-process call
 *   double individ::doanalyze<Turner,Stop>(tb, turner, startmark, endmark, stopdata,
 *                                          flag2, ruleout, realprobs, minfactor)
 * (cnF2freq.cpp:2122-2131) driven per individual by doit<> (cnF2freq.cpp:5294-5583),
 * reading the global individ graph (cnF2freq.cpp:853-914, 2448-2514), markerposes /
 * chromstarts / genrec (cnF2freq.cpp:233-239) and thread-private alpha/beta stores
 * (cnF2freq.cpp:392-394).  This header is the batch form of that seam: the globals
 * become explicit uploads, the OpenMP loop over `dous` becomes one call.
 *
 * Conventions: plain C types, caller-owned buffers, int status (0 = ok, <0 = error,
 * message via cnf2_last_error).  Impossible data is reported in-band exactly like the
 * reference: a log-likelihood <= CNF2_MINFACTOR (or NaN) means "skip this individual"
 * (cnF2freq.cpp:1659, 5403); no exceptions, no abort.
 * One context per GPU; a context is not thread-safe, distinct contexts are independent.
 * The library fails loudly (CNF2_ERR_NO_DEVICE) when no HIP device is usable; there is
 * no CPU fallback.
 */
#ifndef CNF2HIP_H
#define CNF2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNF2_NUMTYPES   64      /* settings.h:27  inheritance states          */
#define CNF2_NUMSHIFTS  8       /* settings.h:35  phase-shift modes           */
#define CNF2_NUMPATHS   128     /* settings.h:32  allele-assignment paths     */
#define CNF2_MINFACTOR  (-1e15f)/* settings.h:29                              */
#define CNF2_IGNORED    (-1e30) /* factors[] value of a masked shift mode, cnF2freq.cpp:5378 */

enum {
    CNF2_OK            = 0,
    CNF2_ERR_NO_DEVICE = -1,
    CNF2_ERR_ARG       = -2,
    CNF2_ERR_STATE     = -3,   /* call order violated (e.g. sweep before uploads) */
    CNF2_ERR_HIP       = -4,
    CNF2_ERR_NOMEM     = -5
};

/* cnf2_sweep flags */
enum {
    CNF2_OUT_DEVICE   = 1u << 0, /* output pointers are device pointers (no D2H copy, no sync) */
    CNF2_NO_DOSAGE    = 1u << 1, /* HOT LOOP 1 only: factors/loglik, skip the per-locus rows    */
    CNF2_RAW_DOSAGE   = 1u << 2, /* rows un-normalised (sum of val by class, cnF2freq.cpp:3523) */
    CNF2_NO_TIES      = 1u << 3, /* drop ignoreflag2's all-or-none rule (cnF2freq.cpp:3484-3486) */
    CNF2_FULL_SPILL   = 1u << 4, /* store alpha-minus at every marker instead of every second one and
                                    recomputing the others in the backward pass (same results) */
    CNF2_MERGE_MODES  = 1u << 5, /* sweep bit-identical shift modes once: for a window whose two parents are
                                    homozygous with equal sure at every marker (e.g. the private empty F1
                                    parents of an F2, cnF2freq.cpp:6515-6527) the modes that differ in the
                                    parents' shift bits carry the same alpha/beta; four such individuals
                                    share a wavefront.  Same outputs for all 8 modes (rows to rounding).
                                    Ignored together with CNF2_FULL_SPILL. */
    CNF2_ACC_DEVICE   = 1u << 6, /* cnf2_sweep_accumulate: the four accumulator pointers are device pointers owned by the
                                    caller (a multi-GPU driver all-reduces them in place).  They must be ordinary
                                    (coarse-grained) device memory -- hipMalloc or a torch CUDA tensor: the kernels add with
                                    hardware f64 atomics (-munsafe-fp-atomics), which fine-grained or host-mapped memory
                                    silently drops.  The order of the additions is not fixed: accumulators are
                                    reproducible to rounding, not to the bit, unless CNF2_DETERMINISTIC is set */
    CNF2_ACC_KEEP     = 1u << 7, /* cnf2_sweep_accumulate: add to the per-record accumulators instead of zeroing them */
    CNF2_ACC_TABLE    = 1u << 10, /* cnf2_sweep_accumulate: evaluate every window in the table form (one lane per emission-table
                                    entry) instead of the path form (one lane per allele path of a line); same sums, the
                                    slower kernel -- kept as the cross-check of the fast one and for windows whose root
                                    is the top of its own lines, which always take it */
    CNF2_ACC_LANES    = 1u << 11, /* cnf2_sweep_accumulate: path form with one lane per path for every window (the kernel that
                                    windows with tie groups always take) instead of the tile form; A/B and cross-check */
    CNF2_TIES_GENERAL = 1u << 12, /* cnf2_sweep, cnf2_sweep_accumulate, cnf2_sweep_turn_scan: windows with tie groups through the general kernel (one lane per table entry, producer
                                    per marker) instead of the tile-producer kernel's pass per tie combination; cross-check */
    CNF2_UPDATE_PLAIN = 1u << 13, /* cnf2_update_pass: the literal form -- one thread per (record, marker), every bisection step with its
                                    quadrature as the reference's cappedgd runs it -- instead of the scout / finish kernels, which
                                    make the same decisions with a fraction of the gradient evaluations (cnf2_update.h).  The
                                    yardstick of the fast form and an A/B switch: the fast form gives the same results TO THE
                                    BIT only together with CNF2_UPDATE_BOTH_FLOWS; its default (one certainty flow per side, the
                                    other its mirror image) is within rounding of this form, not bit-identical */
    CNF2_UPDATE_BOTH_FLOWS = 1u << 16, /* cnf2_update_pass, fast form: run the certainty flow of BOTH allele values of a side, as
                                    processinfprobs does (cnF2freq.cpp:4222-4290), instead of one flow and its mirror image
                                    (DESIGN.md section 3.11).  The bit-exact form: equal to CNF2_UPDATE_PLAIN to the bit */
    CNF2_UPDATE_ONE_SCOUT = 1u << 17, /* cnf2_update_pass, fast form: the scouts (certainties and, since round 5, weights) in one pass
                                    instead of two (same results to the bit; A/B switch, tools/ab_scout.py) */
    CNF2_UPDATE_LITERAL_FINISH = 1u << 19, /* cnf2_update_pass, fast form: the flows the scouts set aside take one literal bisection
                                    step (midpoint, bound, quadrature) per round, as in rounds 3 / 4, instead of the guided
                                    bisection (cnf2_update.h: the same decisions from 3 - 4 quadratures per flow).  Same
                                    results to the bit; A/B switch and cross-check */
    CNF2_FLUSH_TINY   = 1u << 20, /* cnf2_sweep: every window through the general kernel (one lane per table entry, vectors
                                    normalised at every marker as the reference normalises them) WITH adjustprobs' rule that a
                                    state under 1e-300 of its vector is set to exactly 0 before the emission is applied
                                    (cnF2freq.cpp:1607-1611).  The fast kernels carry such states (no
                                    effect on valid data above 1e-6 -- except where every OTHER state then becomes exactly
                                    impossible, a locked phase contradicting certain genotypes: the reference declares the
                                    shift mode impossible, the fast kernels return the likelihood of the 1e-303 state).
                                    This flag is the reference's behaviour to the letter, at the general kernel's speed */
    CNF2_DETERMINISTIC = 1u << 14, /* cnf2_sweep_accumulate: every analysed individual writes what its window members receive at a
                                    locus into a row of its own (336 B per individual x marker, allocated for the whole
                                    range) and one more kernel adds the rows of every record in ascending order of the
                                    individual, instead of f64 atomics in order of arrival: accumulators -- and with them
                                    whole iterations -- reproduce to the bit from run to run */
    CNF2_TURN_VALU    = 1u << 15, /* cnf2_sweep_turn_scan: the 1 024 dot products per (individual, marker) on the vector ALU (one lane
                                    per pair of shift modes, flips as register renaming) instead of the matrix cores
                                    (v_mfma_f64_16x16x4: a 32 x 32 x 64 product per unit); same sums in another order: cross-check, A/B */
    CNF2_STATIC_JOBS  = 1u << 18, /* cnf2_sweep, cnf2_sweep_accumulate, cnf2_sweep_turn_scan: wavefront w of a launch sweeps jobs w, w + waves, ...
                                     instead of taking its jobs one at a time from the launch's counter (the default: whichever
                                     blocks are resident share the job list evenly, whatever the chromosomes' lengths and
                                     however the kernels of the tied and the untied windows share the machine).  Same results
                                     to the bit: a job's arithmetic does not depend on the wave that runs it. */
    CNF2_LOG_PATHS    = 1u << 9, /* cnf2_sweep records which kernel / producer specialisation swept every job (cnf2_last_paths) */
    CNF2_XPOSE        = 1u << 8  /* sweep kernel variant: the three lane-held state bits of the transition are brought into
                                    registers by a transpose through LDS instead of being exchanged by DPP moves (same
                                    results to rounding; A/B switch while the variant is evaluated) */
};

typedef struct cnf2_ctx cnf2_ctx;

/* Library / device ------------------------------------------------------------ */
int         cnf2_device_count(void);
int         cnf2_ctx_create(int device, cnf2_ctx **out);
void        cnf2_ctx_destroy(cnf2_ctx *ctx);
const char *cnf2_last_error(const cnf2_ctx *ctx);           /* ctx may be NULL: last create error */
const char *cnf2_version(void);

/* Marker map: replaces markerposes / chromstarts / genrec
 * (readalphamap cnF2freq.cpp:6669-6685; main cnF2freq.cpp:7927-7943).
 * pos[n_markers] in cM; chromstarts[n_chrom+1] with chromstarts[n_chrom] == n_markers;
 * genrec[3] (NULL = {-0.02,-0.02,-0.02}). */
int cnf2_upload_map(cnf2_ctx *ctx, const double *pos, int n_markers, const int32_t *chromstarts,
                    int n_chrom, const double *genrec);

/* Genotype rows: replaces individ::markerdata / markersure / haploweight
 * (cnF2freq.cpp:876-887; filled by readalphadata cnF2freq.cpp:6542-6667).
 * Rows are de-duplicated storage; several individuals may share one row (all
 * `empty` individuals normally share a blank row: alleles 0, sure 0, hw 0.5).
 *   allele [n_rows][n_markers][2]  MarkerVal values 0 (unknown), 1, 2, 9
 *   sure   [n_rows][n_markers][2]
 *   hw     [n_rows][n_markers]
 * cnf2_update_rows overwrites rows [row0,row0+n) between sweeps (the per-iteration
 * parameter updates of cnF2freq.cpp:6344-6368 stay on the host). */
int cnf2_upload_rows(cnf2_ctx *ctx, int n_rows, const uint8_t *allele, const double *sure,
                     const double *hw);
int cnf2_update_rows(cnf2_ctx *ctx, int row0, int n, const uint8_t *allele, const double *sure,
                     const double *hw);
/* cnf2_upload_rows with all three pointers NULL allocates n_rows blank rows (alleles 0,
 * sure 0, hw 0.5: an individual without data, getind cnF2freq.cpp:2486-2493).
 * cnf2_update_rows_device takes DEVICE pointers and the packed allele form the kernels use:
 * d_allele8[n][n_markers] = first | second << 4; d_sure[n][n_markers][2]; d_hw[n][n_markers]. */
int cnf2_update_rows_device(cnf2_ctx *ctx, int row0, int n, const uint8_t *d_allele8,
                            const double *d_sure, const double *d_hw);

/* Pedigree graph: replaces individer[] / individ::{pars,empty,gen} and `dous`
 * (readalphaped cnF2freq.cpp:6495-6540).  par[n_rec][2] record index or -1;
 * row_of[n_rec] row index; dous[n_dous] the analysed records in output order.
 * The library derives, per analysed individual, what fixtrees (cnF2freq.cpp:3099-3187)
 * produces: the 7-slot window, shiftignore, flag2ignore, founder flags (for every
 * record, as postmarkerdata does, cnF2freq.cpp:3373-3389) and the groups of slots
 * occupied by one ancestor (relmap). */
int cnf2_upload_pedigree(cnf2_ctx *ctx, int n_rec, const int32_t *par, const uint8_t *empty,
                         const int32_t *gen, const int32_t *row_of, const int32_t *dous,
                         int n_dous);

/* Window topology as derived by the library (parity hook for fixtrees):
 * out[0]=shiftignore out[1]=flag2ignore out[2]=founder out[3..9]=slot records (-1 none)
 * out[10..16]=tie group per slot (-1 = ancestor occupies a single slot). */
int cnf2_window_info(cnf2_ctx *ctx, int ind, int32_t *out17);
/* the same for every analysed individual in one call: out17_all[n_dous][17] */
int cnf2_window_table(cnf2_ctx *ctx, int32_t *out17_all);

/* The sweep: per-individual body of doit<> (cnF2freq.cpp:5294-5403) plus the per-locus
 * allele-2 dosage posterior row that genotypereporter accumulates (cnF2freq.cpp:5406-5553,
 * 3532-3538) for analysed individuals [ind_begin, ind_end), every chromosome.
 *   factors_out [n][n_chrom][8]  log-likelihood per shift mode (CNF2_IGNORED if masked)
 *   loglik_out  [n][n_chrom]     logsumexp over modes; <= CNF2_MINFACTOR or NaN => skipped
 *   dosage_out  [n][n_markers][3] posterior of 0/1/2 copies of allele 2, rows normalised
 *                                (all-zero row for a skipped individual); may be NULL with
 *                                CNF2_NO_DOSAGE
 * With CNF2_OUT_DEVICE the three pointers are device pointers and the call only enqueues
 * work on the context's stream (use cnf2_sync). */
int cnf2_sweep(cnf2_ctx *ctx, int ind_begin, int ind_end, double *factors_out, double *loglik_out,
               double *dosage_out, uint32_t flags);
int cnf2_sync(cnf2_ctx *ctx);

/* Parity/debug view of the alpha/beta store of one individual and chromosome in the
 * reference's layout (cnF2freq.cpp:392-393): fwbw_out[8][mc][3][64] with slot 0 = alpha
 * before emission, 1 = beta, 2 = alpha after emission; fwbwfactors_out[8][mc][3] the
 * cumulative log scales; mc = markers on the chromosome.  Masked modes are left zero. */
int cnf2_fwbw_store(cnf2_ctx *ctx, int ind, int chrom, double *fwbw_out, double *fwbwfactors_out);

/* Stage-2 consumers of the alpha/beta store (parity level, one individual x chromosome per call;
 * not tuned).  They answer in bulk the queries that doit<> issues one by one:
 *  cnf2_locked_query    val_out[8][64][128]: exp(doanalyze(classicstop(q, g), flag2) - factor) for
 *                       every shift mode, state g and path flag2 at `marker` (cnF2freq.cpp:5499-5508;
 *                       0 where the reference would not count the term).  No ignoreflag2 pruning
 *                       is applied (cnF2freq.cpp:5464): the caller masks.
 *  cnf2_turn_scan       rawervals_out[128][8]: doanalyze<aroundturner>(turn, classicstop(q, -1)) - factor
 *                       (cnF2freq.cpp:5686-5724) for every turn and shift mode, unmasked.
 *  cnf2_state_posterior rows_out[mc][64]: what statereporter::addval accumulates (cnF2freq.cpp:3540-3546),
 *                       i.e. sum of val over shift modes and admissible paths by state, per marker. */
int cnf2_locked_query(cnf2_ctx *ctx, int ind, int chrom, int marker, double *val_out);
int cnf2_turn_scan(cnf2_ctx *ctx, int ind, int chrom, int marker, double *rawervals_out);
/*  cnf2_turn_scan_rows  the same for every marker of the chromosome in one launch: rows_out[mc][128][8] */
int cnf2_turn_scan_rows(cnf2_ctx *ctx, int ind, int chrom, double *rows_out);
int cnf2_state_posterior(cnf2_ctx *ctx, int ind, int chrom, double *rows_out, uint32_t flags);
/*  cnf2_haplos          rows_out[mc][7][2]: the HAPLOS accumulators HOT LOOP 2 leaves per window slot
 *                       (slot order of cnf2_window_info) before movehaplos: sum of val by the phase with
 *                       which the slot's individual is used (cnF2freq.cpp:1347-1350, 1561-1575, 5556).
 *                       An individual occupying several slots gets the sum of its slots in the reference. */
int cnf2_haplos(cnf2_ctx *ctx, int ind, int chrom, double *rows_out, uint32_t flags);
/*  cnf2_infprobs        the other accumulators of HOT LOOP 2 at `marker` (DOINFPROBS, cnF2freq.cpp:5513-5577):
 *                       inf_out[7][2][2] = thread-private infprobs[slot][allele index][markerval - 1] before
 *                       moveinfprobs (cnF2freq.cpp:3577-3597; trackpossible<GENOSPROBE> weights, <GENOS>
 *                       updates, cnF2freq.cpp:1351-1354), hz_out[2] = what is added to the individual's
 *                       homozyg[marker] (trackpossible<HOMOZYGOUS>, cnF2freq.cpp:1304-1320).  Brute force over
 *                       (shift mode, state, path) like the reference; sums are accumulated atomically. */
int cnf2_infprobs(cnf2_ctx *ctx, int ind, int chrom, int marker, double *inf_out, double *hz_out, uint32_t flags);
/*  cnf2_infprobs_rows   the same accumulators for every marker of the chromosome, rows_out[mc][30] = infprobs
 *                       [7][2][2] followed by homozyg[2], through their closed form (cnf2_accum.h: the weight of
 *                       a path depends on one line of descent only, the other line enters as its restricted
 *                       total) instead of the 128-path fan-out. */
int cnf2_infprobs_rows(cnf2_ctx *ctx, int ind, int chrom, double *rows_out, uint32_t flags);

/* HOT LOOP 2 with its reductions: for the analysed individuals
 * [ind_begin, ind_end), in that order, the per-locus accumulators of cnF2freq.cpp:5416-5577 are formed on the GPU
 * (closed forms: cnf2_haplos, cnf2_infprobs_rows) and reduced per individual as the reference does after every
 * locus (cnF2freq.cpp:5876-5902): homozyg scaled by 1 / sum of the individual's own allele-index-0 infprobs;
 * moveinfprobs (cnF2freq.cpp:3577-3597) and movehaplos (cnF2freq.cpp:3599-3616) for every window member, with the
 * individual's `descendants` count as weight.  Outputs are zeroed first: infprobs_out[n_rec][M][2][2] (side,
 * markerval 1/2), haplobase_out / haplocount_out[n_rec][M], homozyg_out[ind_end - ind_begin][M][2].
 * cnf2_descendants fills descendants[n_rec] as postmarkerdata computes them (cnF2freq.cpp:3224-3255). */
int cnf2_descendants(cnf2_ctx *ctx, int32_t *desc_out);
int cnf2_accumulate(cnf2_ctx *ctx, int ind_begin, int ind_end, const int32_t *descendants, double *infprobs_out,
                    double *haplobase_out, double *haplocount_out, double *homozyg_out, uint32_t flags);
/* The product form: one call = one haplotyping sweep of doit<> over the individuals (cnF2freq.cpp:5294-5583, 5876-5902):
 * the outputs of cnf2_sweep (may be NULL unless CNF2_OUT_DEVICE) AND the per-record accumulators, all individuals and
 * chromosomes batched on the device.  The sweep kernels run in their accumulate instantiation (posterior weights of
 * every (individual, marker, shift mode, state) into a batch buffer); one more kernel forms every accumulator of a
 * locus through per-line tables (cnf2_acctab.h) and applies homozyg's scale, moveinfprobs and movehaplos with f64
 * atomics.  CNF2_ACC_DEVICE: infprobs / haplobase / haplocount / homozyg are caller-owned device buffers
 * ([n_rec][M][2][2], [n_rec][M], [n_rec][M], [ind_end - ind_begin][M][2]); several GPUs that share ancestors sum
 * the first three with one all-reduce (the reference's reduce calls, cnF2freq.cpp:6245-6254).
 * dosage_out == NULL: the per-locus rows are not formed at all (an iteration that prints none, cnF2freq.cpp:6183: all but
 * the last of a run): the sweep runs in an instantiation without class sums, restricted tables and row epilogue;
 * likelihoods and accumulators are those of a call with rows (to the bit for windows without tie groups, to rounding
 * for the others, whose posterior weights then come from another instantiation of the kernel). */
int cnf2_sweep_accumulate(cnf2_ctx *ctx, int ind_begin, int ind_end, const int32_t *descendants, double *factors_out,
                          double *loglik_out, double *dosage_out, double *infprobs, double *haplobase,
                          double *haplocount, double *homozyg, uint32_t flags);
/* The allocations a later cnf2_sweep_accumulate(ind_begin, ind_end, ..., flags) makes -- accumulators, outputs, spill rows and
 * the batch buffer of posterior weights (sized to half of the free device memory; a first hipMalloc of that size takes
 * seconds) -- without the sweep, so that a run's first iteration (doit, cnF2freq.cpp:5294) costs what the others cost. */
int cnf2_reserve_accumulate(cnf2_ctx *ctx, int ind_begin, int ind_end, uint32_t flags);

/* Batched turn scan (HOT LOOP 3; cnF2freq.cpp:5686-5752 with aroundturner 498-554): for every
 * analysed individual in [ind_begin, ind_end) and every marker,
 *   rawervals_out [n][M][128][8]  doanalyze<aroundturner>(turn, classicstop(q, -1)) - factor for every turn and shift mode,
 *                                 unmasked like cnf2_turn_scan (the caller applies flag2ignore / shiftignore), and / or
 *   turn_lse_out  [n][M][128]     per turn the log-sum-exp of those values over the admissible shift modes: the quantity
 *                                 the clause weights are made of (computew, cnF2freq.cpp:5800-5817: weight(turn) =
 *                                 (lse[turn] - lse[0] * descendants) * descendants).
 * Either pointer may be NULL.  Host buffers are staged through one device buffer of the full size (9 KB per
 * individual x marker); with CNF2_OUT_DEVICE they are device pointers and only the batch buffer is allocated.
 * The sweep kernels run in their turn-scan instantiation; all individuals and chromosomes in batched launches. */
int cnf2_sweep_turn_scan(cnf2_ctx *ctx, int ind_begin, int ind_end, double *rawervals_out, double *turn_lse_out,
                         uint32_t flags);

/* Pre-processing user of the emission (parity level): individ::addvariance
 * (cnF2freq.cpp:1489-1558, called by postmarkerdata for every marker, cnF2freq.cpp:3373-3389) for one analysed
 * individual and chromosome: var_out[mc] = variances[marker], NaN where the reference leaves the entry alone
 * (every term zero).  Brute force over (shift mode 0-1, flag, path) with trackpossible<false, NO_EQUIVALENCE>. */
int cnf2_addvariance(cnf2_ctx *ctx, int ind, int chrom, double *var_out);

/* The same two pre-processing users for ARBITRARY records, batched (postmarkerdata runs them on every individual,
 * cnF2freq.cpp:3257-3279, 3373-3389):
 *  cnf2_fixparents_scan  ok_out[n][M][2]: fixparents' admissibility test (cnF2freq.cpp:1411-1431): is any (state, path of
 *                        parity b) possible at the marker under shift mode 0 with CORRECTIONINFERENCE set, no founder flag
 *                        assigned yet (main() calls postmarkerdata before any fixtrees, cnF2freq.cpp:8083-8085)
 *  cnf2_variances        var_out[n][M] as cnf2_addvariance, through the closed form of cnf2_variance.h (the sums over states and
 *                        paths factorise per line; one thread per record x marker).  ordered bit 0: founder flags as fixtrees
 *                        has assigned them when the records are visited in ascending order (an ancestor's flag counts if its
 *                        record index is not above the record's own), else every flag; bit 1: brute force like
 *                        cnf2_addvariance (cross-check).
 *  cnf2_variances_exact  var_out[n]: the entry of record recs[q] at marker markers[q] with the reference's OWN rounding -- the
 *                        reference's additions in the reference's order (cnF2freq.cpp:1514-1541), bit-equal to its variances[]
 *                        on goldens G10 / G12.  lockhaplos (cnF2freq.cpp:3056) takes the first marker of STRICTLY largest
 *                        variance, and mirror-image configurations tie in exact arithmetic: which of them the reference locks
 *                        is decided by the last bits of its sums, so the host evaluates the markers that can win through this
 *                        entry (a handful per record and chromosome) and compares those. */
int cnf2_fixparents_scan(cnf2_ctx *ctx, const int32_t *recs, int n, uint8_t *ok_out);
int cnf2_variances(cnf2_ctx *ctx, const int32_t *recs, int n, int ordered, double *var_out);
int cnf2_variances_exact(cnf2_ctx *ctx, const int32_t *recs, const int32_t *markers, int n, int ordered, double *var_out);

/* Per-iteration parameter updates on the device (processinfprobs cnF2freq.cpp:4179-4323,
 * updatehaploweights 4533-4734, cappedgd 4040-4177 with an own 15-point Gauss-Legendre rule; toulbar2 and the phase
 * inversions it decides stay out: negshift is never set, no haplotype is inverted between iterations).
 *  cnf2_snapshot_priors  remembers the rows as they are now as priormarkerdata / priormarkersure (what readalphadata
 *                        copies at cnF2freq.cpp:6664-6665); has_prior[n_rec] = the record was genotyped.  Call after
 *                        cnf2_upload_rows / cnf2_upload_pedigree, before any correction is written to the rows.
 *  cnf2_update_pass      what doit does after the sweep of chromosome `chrom` (cnF2freq.cpp:6232-6392): new markerdata /
 *                        markersure from the infprobs of that chromosome's markers (then cleared), new haplotype weights
 *                        for every marker of chromosomes 0..chrom (haplobase / haplocount are rewritten as the reference
 *                        leaves them), written straight into the rows; *hits_out = hitnnn of this pass.  Accumulators:
 *                        NULL = the ones cnf2_sweep_accumulate left in the context; device pointers with CNF2_ACC_DEVICE;
 *                        else host arrays (uploaded, updated, copied back).  children[n_rec]: analysed children per record
 *                        (cnF2freq.cpp:5248-5260), descendants[n_rec].  Non-empty records must not share a row.
 *  cnf2_download_rows    rows [row0, row0 + n) back to the host in the layout of cnf2_upload_rows. */
int cnf2_snapshot_priors(cnf2_ctx *ctx, const uint8_t *has_prior);
int cnf2_update_pass(cnf2_ctx *ctx, int chrom, const int32_t *children, const int32_t *descendants, double *infprobs,
                     double *haplobase, double *haplocount, double scalefactor, double entropyfactor, int *hits_out,
                     uint32_t flags);
int cnf2_download_rows(cnf2_ctx *ctx, int row0, int n, uint8_t *allele, double *sure, double *hw);
/* cnf2_update_pass restricted to the listed records (ascending; n_recs may be 0), on the accumulators the context holds:
 * the form a rank of a multi-process run uses -- every record's update reads only its own accumulators and rows
 * (cnF2freq.cpp:6344-6368 loops over individuals), so ranks update the records they own and *hits_out counts those. */
int cnf2_update_pass_records(cnf2_ctx *ctx, int chrom, const int32_t *recs, int n_recs, const int32_t *children,
                             const int32_t *descendants, double scalefactor, double entropyfactor, int *hits_out, uint32_t flags);

/* Exchange support of multi-process haplotyping runs (the reference's reduce calls,
 * cnF2freq.cpp:6245-6254): what ranks exchange is the records their windows SHARE, packed -- not the [n_rec][M] slabs.
 *  cnf2_exchange_buffer            a device staging buffer of at least `bytes` owned by the context (grows; the pointer is
 *                                  valid until the next call with a larger size)
 *  cnf2_pack_accumulators          d_packed[n][M][6] <- the context's accumulators of the listed records: per record
 *                                  infprobs [M][2][2], then haplobase [M], then haplocount [M]
 *  cnf2_unpack_accumulators        the reverse (overwrites the listed records' accumulators)
 *  cnf2_pack_rows / _unpack_rows   the genotype rows of the listed records, cnf2_packed_row_bytes() per record:
 *                                  sure [M][2] f64, haploweight [M] f64, alleles [M] u8 (a0 | a1 << 4), padded to 8 bytes
 * d_packed are device pointers; the calls return when the copy is done. */
int    cnf2_exchange_buffer(cnf2_ctx *ctx, size_t bytes, void **d_buf);
/* the first `bytes` of the exchange buffer to / from host memory: for transports that move host memory (gloo, MPI without
 * GPU support); a device-aware transport (RCCL) works on the buffer in place */
int    cnf2_exchange_download(cnf2_ctx *ctx, void *host_dst, size_t bytes);
int    cnf2_exchange_upload(cnf2_ctx *ctx, const void *host_src, size_t bytes);
/* ... and any byte range of it (transports that stage the buffer in chunks: the shared-memory transport of `cnF2freq --gpus N`) */
int    cnf2_exchange_read(cnf2_ctx *ctx, size_t offset, void *host_dst, size_t bytes);
int    cnf2_exchange_write(cnf2_ctx *ctx, size_t offset, const void *host_src, size_t bytes);
size_t cnf2_packed_accumulator_doubles(const cnf2_ctx *ctx);
size_t cnf2_packed_row_bytes(const cnf2_ctx *ctx);
int    cnf2_pack_accumulators(cnf2_ctx *ctx, const int32_t *recs, int n, double *d_packed);
int    cnf2_unpack_accumulators(cnf2_ctx *ctx, const int32_t *recs, int n, const double *d_packed);
int    cnf2_pack_rows(cnf2_ctx *ctx, const int32_t *recs, int n, void *d_packed);
int    cnf2_unpack_rows(cnf2_ctx *ctx, const int32_t *recs, int n, const void *d_packed);
/* Diagnostics of the update passes since the last pass of chromosome 0, i.e. of an iteration so far (flow kernels; see cnf2_update_kernels.hip).  out16[0..3] for the genotype
 * certainties, out16[4..7] for the haplotype weights: flows; gradient evaluations the scout spent on them; flows that ended
 * in the scout; flows pinned to their clamp (no evaluation beyond the first).  out16[8..11] / out16[12..15] for the flows the
 * scout set aside: steps taken in the finish kernel; lane slots offered (steps / slots = lane utilisation); quadratures;
 * flows that ended because the tolerance was met.  Collected only when the environment holds CNF2_UPDATE_STATS. */
int cnf2_update_stats(cnf2_ctx *ctx, uint64_t *out16);
/* The same for the lock-step kernels of the guided bisection (cnf2_update.h; the flows the scouts set aside go through
 * them first, out16[8..15] of cnf2_update_stats then describe the persistent kernel that takes what they leave):
 * out8[0..3] certainties, out8[4..7] haplotype weights: literal points evaluated; lane slots offered; gradient evaluations;
 * flows that ended because the tolerance was met. */
int cnf2_update_stats_guided(cnf2_ctx *ctx, uint64_t *out8);
/* The accumulators the context holds (what cnf2_sweep_accumulate left and cnf2_update_pass rewrote when they were called
 * with NULL accumulator pointers): host copies infprobs[n_rec][M][2][2], haplobase / haplocount[n_rec][M]; any pointer may
 * be NULL.  cnf2_upload_accumulators is the reverse (a multi-process driver whose transport moves host memory sums the
 * slabs of its ranks between the two calls). */
int cnf2_download_accumulators(cnf2_ctx *ctx, double *infprobs, double *haplobase, double *haplocount);
/* Device addresses of the same three slabs (valid until the pedigree or the map is replaced): what a multi-GPU driver
 * hands to its all-reduce (RCCL) between cnf2_sweep_accumulate and cnf2_update_pass.  Call cnf2_sync first. */
int cnf2_accumulator_ptrs(cnf2_ctx *ctx, double **infprobs, double **haplobase, double **haplocount);
int cnf2_upload_accumulators(cnf2_ctx *ctx, const double *infprobs, const double *haplobase, const double *haplocount);

/* Emission lookup of one analysed individual and marker, all 8 shift modes (parity hook
 * for trackpossible, cnF2freq.cpp:1075-1359): e_out[8][64] path-free emission e(g). */
int cnf2_emission(cnf2_ctx *ctx, int ind, int marker, double *e_out);
/* The same resolved by allele path: e_out[8][64][128] = trackpossible(..., 2g, flag2, s) for every shift mode,
 * state and path flag2 (calltrackpossible with flag2 >= 0, cnF2freq.cpp:1380-1385, 1141-1146). */
int cnf2_emission_paths(cnf2_ctx *ctx, int ind, int marker, double *e_out);

/* Diagnostic: out384[k*64 + lane] = value 1000+src received by `lane` from the lane-exchange
 * primitive of distance 1<<k (k = 0..5) that the transition butterflies are built on. */
int cnf2_selftest_lane_xor(cnf2_ctx *ctx, double *out384);

/* Measurement support for bench.py: duration in ms of the kernels of the last cnf2_sweep
 * measured with hipEvents on the context's stream (kernel_ms[0] = forward-backward kernel),
 * and workspace bytes currently allocated on the device. */
int    cnf2_last_kernel_ms(cnf2_ctx *ctx, float *kernel_ms, int n);
/* After a cnf2_sweep with CNF2_LOG_PATHS: paths_out[n] (n = individuals of that sweep x chromosomes, [ind][chrom]) = which
 * code swept the job: 0-3 the fast kernel with producer class 0 general / 1 both parents homozygous everywhere / 2 and the
 * grandparents too / 3 complete window (restricted table = unrestricted); 16 the fast kernel's instantiation for windows with
 * tie groups (a backward pass per tie combination); 32 | homleaf the merged-modes kernel; 64 the general kernel (tied windows
 * with CNF2_TIES_GENERAL or CNF2_FULL_SPILL).  Test support: the specialisations are exact shortcuts and must all be exercised. */
int    cnf2_last_paths(cnf2_ctx *ctx, int32_t *paths_out, int n);
size_t cnf2_workspace_bytes(cnf2_ctx *ctx);
/* Shader clock the device runs at under a double-precision vector load, MHz (a loop of independent FMAs on every SIMD:
 * one wave-wide FMA issues per 4 cycles).  Boxes of the same model differ by several per cent; an issue-bound kernel
 * tracks this clock, so bench.py reports it next to the roofline fraction. */
int    cnf2_clock_probe(cnf2_ctx *ctx, double *mhz_out);
/* Shader clock of the last cnf2_sweep's untied fast-kernel launch itself, MHz: its first wave reads the shader-clock
 * counter and the constant-rate wall clock when it starts and when it ends (s_memtime / s_memrealtime); 0 when no such
 * launch has run.  Synchronises the context's stream. */
int    cnf2_sweep_clock(cnf2_ctx *ctx, double *mhz_out);
void  *cnf2_stream(cnf2_ctx *ctx); /* hipStream_t of the context */
/* The sweep kernels are persistent (one resident wave per job in flight) and normally fill every
 * workgroup slot of the GPU.  Leaving `blocks` slots free lets another kernel -- the RCCL gather of
 * the previous sweep's posteriors -- run beside the sweep instead of behind it. */
int    cnf2_set_grid_reserve(cnf2_ctx *ctx, int blocks);
/* The batched consumers (cnf2_sweep_accumulate, cnf2_sweep_turn_scan) run their jobs (individual x chromosome) in batches
 * sized to the memory that is free; `jobs` > 0 caps a batch at that many jobs (0 = no cap).  Results do not depend on the
 * batch size; the knob exists so that the multi-batch path can be exercised at test sizes and memory use bounded by a caller
 * that shares the GPU. */
int    cnf2_set_batch_jobs(cnf2_ctx *ctx, int jobs);

#ifdef __cplusplus
}
#endif
#endif /* CNF2HIP_H */
