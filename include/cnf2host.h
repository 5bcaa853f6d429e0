/*
 * cnf2host.h -- C ABI of libcnf2host.so: the host side of a cnF2freq run above libcnf2hip.so (readers' data model,
 * postmarkerdata, the haplotyping iteration with its device-side updates, dump / deserialize), i.e. what main() and
 * doit<> do around the sweep (cnF2freq.cpp:8083-8192, 5189-6410, 3190-3412), driven from arrays instead of files.
 * The `cnF2freq` executable links the same code; this entry exists for callers that already hold the pedigree in
 * memory and for the parity tests.  One run owns one cnf2_ctx on device 0.  int status: 0 ok, < 0 error.
 */
#ifndef CNF2HOST_H
#define CNF2HOST_H

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
void       cnf2h_destroy(cnf2h_run *run);
const char *cnf2h_last_error(void);

/* postmarkerdata(indcount) as main() calls it (cnF2freq.cpp:8083-8085) */
int cnf2h_postmarkerdata(cnf2h_run *run, int indcount);
/* one doit<false, genotypereporter> (cnF2freq.cpp:8132): rows and pass lines appended to rows_path (NULL: not formatted at all);
 * update = 0 sweeps without the parameter updates */
int cnf2h_iteration(cnf2h_run *run, const char *rows_path, int update);
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
