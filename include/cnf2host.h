/*
 * cnf2host.h -- C ABI of libcnf2host.so: the host side of a cnF2freq run above libcnf2hip.so (readers' data model,
 * postmarkerdata, the haplotyping iteration with its device-side updates, dump / deserialize), i.e. what main() and
 * doit<> do around the sweep (cnF2freq.cpp:8083-8192, 5189-6410, 3190-3412), driven from arrays instead of files.
 * The `cnF2freq` executable links the same code; this entry exists for callers that already hold the pedigree in
 * memory, for multi-process drivers (one run per GPU, cnf2h_set_block / cnf2h_set_exchange) and for the parity tests.
 * One run owns one cnf2_ctx.  int status: 0 ok, < 0 error (the codes of cnf2hip.h; text in cnf2h_last_error()): a failure
 * below the ABI (out of memory, a launch error) comes back as a status, it does not end the process.
 */
#ifndef CNF2HOST_H
#define CNF2HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cnf2h_run cnf2h_run;

/* Individuals are records 0..n_rec-1 (the reference's number n = record + 1, order of first mention,
 * cnF2freq.cpp:6480-6491).  par[n_rec][2] record or -1; allele[n_rec][M][2], sure[n_rec][M][2], hw[n_rec][M] as
 * readalphadata leaves them (they are also the priors of the records with has_prior set, cnF2freq.cpp:6664-6665);
 * dous[n_dous] the analysed records in output order. */
cnf2h_run *cnf2h_create(int n_rec, const int32_t *par, const uint8_t *empty, const int32_t *gen, const uint8_t *has_prior,
                        const uint8_t *allele, const double *sure, const double *hw, const double *pos, int n_markers,
                        const int32_t *chromstarts, int n_chrom, const int32_t *dous, int n_dous, int quiet);
/* the same on HIP device `device` (one process per GPU: its local rank) */
cnf2h_run *cnf2h_create_on(int device, int n_rec, const int32_t *par, const uint8_t *empty, const int32_t *gen,
                           const uint8_t *has_prior, const uint8_t *allele, const double *sure, const double *hw, const double *pos,
                           int n_markers, const int32_t *chromstarts, int n_chrom, const int32_t *dous, int n_dous, int quiet);
void       cnf2h_destroy(cnf2h_run *run);
const char *cnf2h_last_error(void);

/* postmarkerdata(indcount) as main() calls it (cnF2freq.cpp:8083-8085) */
int cnf2h_postmarkerdata(cnf2h_run *run, int indcount);
/* one doit<false, genotypereporter> (cnF2freq.cpp:8132): rows and pass lines appended to rows_path (NULL: not formatted at all);
 * update = 0 sweeps without the parameter updates */
int cnf2h_iteration(cnf2h_run *run, const char *rows_path, int update);
/* allocates the device buffers of the iterations now (the batch buffer of the accumulate sweep is up to half of the free
 * memory: a first hipMalloc of that size takes seconds) instead of inside the first iteration; optional */
int cnf2h_reserve(cnf2h_run *run);
/* wall time of the last cnf2h_iteration by where it went, seconds: out5[0] sweep + accumulators, [1] exchanges (multi-process),
 * [2] update passes, [3] the rest on the host (bookkeeping, likelihood lines, rows), [4] total */
int cnf2h_get_timing(cnf2h_run *run, double *out5);
/* Multi-process runs (the reference's dead MPI code: partition cnF2freq.cpp:5297-5299, reduce
 * 6245-6254, updates on the reduced values 6344-6392).  Every rank holds the whole pedigree (ancestors' rows replicated)
 * and runs the same postmarkerdata.  cnf2h_set_partition(rank, world, transport) plans the run -- the same plan on every
 * rank, a pure function of the pedigree:
 *   blocks    contiguous blocks of the analysed individuals (positions in dous) balanced by what the sweep kernels spend
 *             on an individual (markers x (1 + tie combinations)), their boundaries moved by up to a quarter of a block to
 *             where the fewest records are touched from both sides: families that fit in a block stay whole;
 *   records   a record only one rank's windows touch is PRIVATE to that rank; a record several ranks touch is SHARED and
 *             owned by one of them (the one with the fewest so far);
 * and sets this rank's block.  An iteration then is: sweep + accumulators of the block; the accumulators of the SHARED
 * records, packed by owner, summed by one reduce-scatter (nothing at all when no family straddles a boundary); per
 * chromosome the update pass of the records the rank owns and one sum of the hit counters (the step-size control,
 * cnF2freq.cpp:6373-6392, stays identical on all ranks); the new rows of the shared records from their owners to
 * everybody by one all-gather.  The rows of private records stay on their rank until the whole state is asked for
 * (cnf2h_get_state, cnf2h_dump, cnf2h_postmarkerdata, cnf2h_deserialize: collectives then -- call them on every rank).
 * Rows are printed for the rank's block.
 *
 * The transport carries the collectives; the engine does all packing on the device.  op:
 *   CNF2H_X_SUM_SEGMENTS     buf = DEVICE pointer, `count` doubles in `world` segments of `seg`: on return segment `rank`
 *                            holds the sum over all ranks of that segment (a reduce-scatter; other segments undefined)
 *   CNF2H_X_SUM_HITS         buf = HOST int32[count]: in-place sum over all ranks
 *   CNF2H_X_GATHER_SEGMENTS  buf = DEVICE pointer, `count` BYTES in `world` segments of `seg`: every rank has filled its
 *                            own segment; on return all segments are filled on every rank (an all-gather)
 *   CNF2H_X_BARRIER          nothing to move: returns when every rank has called it (used by the `cnF2freq --gpus N`
 *                            executable, whose ranks spool their rows to files for rank 0; runs through this C ABI never
 *                            receive it)
 *   CNF2H_X_BCAST_HOST       buf = HOST pointer, `count` bytes: on return every rank holds rank 0's bytes (cnf2h_postmarkerdata
 *                            of a run whose partition is set: rank 0 does the genotype inference for all and broadcasts
 *                            the rows, descendant counts and lock positions it leaves)
 * 0 = ok.  cnf2freq_amd/dist.py holds the torch.distributed transport (RCCL on the device buffer in place; gloo staged
 * through the host); csrc/host/cnf2_shm_transport.h the one of the executable (forked ranks, staged through shared memory). */
enum { CNF2H_X_SUM_SEGMENTS = 0, CNF2H_X_SUM_HITS = 1, CNF2H_X_GATHER_SEGMENTS = 2, CNF2H_X_BARRIER = 3, CNF2H_X_BCAST_HOST = 4 };
typedef int (*cnf2h_exchange_fn)(void *user, int op, void *buf, size_t count, size_t seg);
int cnf2h_set_partition(cnf2h_run *run, int rank, int world, cnf2h_exchange_fn fn, void *user);
/* the plan in numbers: info[0..1] this rank's block [begin, end); [2] records the rank owns; [3] shared records in all;
 * [4] shared records per segment (the largest owner's); bytes per iteration of [5] the reduce-scatter buffer, [6] the
 * all-gather buffer, [7] the hit counters; [8] the payload (what the shared records occupy in them); [9] private records of
 * the rank.  owned (optional) receives the records the rank owns, ascending (info[2] entries). */
int cnf2h_get_partition(cnf2h_run *run, int64_t *info10, int32_t *owned);
/* a sub-range [begin, end) of the analysed individuals for this process's sweeps and rows (single-process use) */
int cnf2h_set_block(cnf2h_run *run, int begin, int end);
/* block `rank` of `world` as cnf2h_set_partition would cut it; does not set it */
int cnf2h_balanced_block(cnf2h_run *run, int rank, int world, int32_t *begin, int32_t *end);
/* form of the update passes: 0 = fast kernels, one certainty flow per side and its mirror image (within rounding of the
 * reference's form), or CNF2_UPDATE_BOTH_FLOWS (fast kernels, bit-exact form), CNF2_UPDATE_PLAIN (literal kernels),
 * CNF2_UPDATE_ONE_SCOUT of cnf2hip.h.  Until this is called a run uses 0, or CNF2_UPDATE_BOTH_FLOWS once
 * cnf2h_set_deterministic is on.  Nothing in the libraries reads the environment for this. */
int cnf2h_set_update_flags(cnf2h_run *run, uint32_t flags);
/* accumulators added in a fixed order (CNF2_DETERMINISTIC of cnf2hip.h) and the update passes in their bit-exact form
 * (CNF2_UPDATE_BOTH_FLOWS): iterations reproduce to the bit */
int cnf2h_set_deterministic(cnf2h_run *run, int on);
/* the cnf2_ctx of the run (for callers that move the accumulators themselves: cnf2_download_accumulators, ...) */
void *cnf2h_context(cnf2h_run *run);
/* after an iteration: hitnnn of every chromosome's update pass (hits[n_chrom]) and haplobase / haplocount [n_rec][M] as
 * the last pass left them; any pointer may be NULL */
int cnf2h_get_passes(cnf2h_run *run, int32_t *hits, double *haplobase, double *haplocount);
/* the dump of cnF2freq.cpp:8157-8192 to a file (append), and deserialize (cnF2freq.cpp:7757-7832) from one */
int cnf2h_dump(cnf2h_run *run, const char *path, int limit);
int cnf2h_deserialize(cnf2h_run *run, const char *path);
/* current state: allele[n_rec][M][2], sure[n_rec][M][2], hw[n_rec][M], descendants / children[n_rec],
 * variances[n_rec][M]; any pointer may be NULL */
int cnf2h_get_state(cnf2h_run *run, uint8_t *allele, double *sure, double *hw, int32_t *descendants, int32_t *children,
                    double *variances, double *scalefactor, int32_t *last_hits);

#ifdef __cplusplus
}
#endif
#endif /* CNF2HOST_H */
