/*
 * cnf2_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the cnF2freq forward-backward hot path.
 * This is the *checker* for the HIP product in cnf2freq_amd/csrc; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product path never links or calls anything in oracle/.
 *
 * Parity status: the reference repository ships no tests or golden vectors
 * for this path (demooutput is stale, format only) => "parity unpinned" by
 * the reference's own fixtures.  The restatement is instead cross-checked in
 * the build container against an extract of the reference's own hot-path
 * code compiled into oracle/_ref (see oracle/ref_extract/), and the vectors
 * from that build are committed under tests/golden/.
 *
 * Every function cites the reference lines it follows
 * (cpp: = /root/reference/cnF2freq.cpp, set: = /root/reference/settings.h).
 */
#ifndef CNF2_ORACLE_H
#define CNF2_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* set:19-35 (NUMGEN=3 build) */
enum {
    CNF2O_NUMGEN    = 3,
    CNF2O_TYPEBITS  = 6,
    CNF2O_NUMTYPES  = 64,
    CNF2O_NUMSHIFTS = 8,
    CNF2O_NUMPATHS  = 128,
    CNF2O_TURNBITS  = 7,
    CNF2O_NUMTURNS  = 128
};
#define CNF2O_MINFACTOR (-1e15f) /* set:29 (a float in the reference) */

/*
 * Pedigree + per-marker inputs, one "record" per individual that appears
 * anywhere in a 3-generation window (cpp:853-914 `individ`, only the fields
 * the sweep reads).  Alleles follow MarkerVal (cpp:188-226): 0 unknown,
 * 1/2 alleles, 9 the sex-marker sentinel.
 */
typedef struct cnf2o_ped {
    int           n_rec;
    int           n_markers;      /* markerposes.size() */
    const int32_t *allele;        /* [n_rec][n_markers][2]  markerdata      */
    const double  *sure;          /* [n_rec][n_markers][2]  markersure      */
    const double  *hw;            /* [n_rec][n_markers]     haploweight     */
    const int32_t *par;           /* [n_rec][2] record index or -1          */
    const uint8_t *founder;       /* [n_rec] individ::founder               */
    const uint8_t *empty;         /* [n_rec] individ::empty                 */
    const double  *pos;           /* [n_markers] markerposes (cM)           */
    double        genrec[3];      /* cpp:7927-7943: all -0.02               */
    int           correction_inference; /* set:105, false during sweeps (cpp:8085) */
} cnf2o_ped;

/* cpp:3099-3187.  relmap/relmapshift are returned as parallel arrays over the
 * distinct window members (sorted by record index, like the flat_map over
 * individ* is sorted by pointer; order does not change ignoreflag2's result). */
typedef struct cnf2o_tree {
    int shiftignore;
    int flag2ignore;
    int founder;              /* value fixtrees would assign to ind->founder (only ever set true) */
    int n_rel;                /* entries in rel_* */
    int rel_rec[7];
    int rel_map[7];
    int rel_mapshift[7];
    int ordered[7];           /* reltreeordered: slot -> record or -1 */
} cnf2o_tree;

void cnf2o_fixtrees(const cnf2o_ped *P, int ind, cnf2o_tree *out);

/* Applies fixtrees' founder side effect for every record (what postmarkerdata
 * does for all individuals, cpp:3373-3389).  founder_out[n_rec]. */
void cnf2o_founder_flags(const cnf2o_ped *P, uint8_t *founder_out);

/* cpp:3462-3496 */
int cnf2o_ignoreflag2(const cnf2o_ped *P, const cnf2o_tree *T, int flag2, int g,
                      int shiftflagmode, int marker);

/* cpp:1075-1359 with update == 0.  zeropropagate in {0,1}; gstr may be NULL. */
double cnf2o_trackpossible(const cnf2o_ped *P, int rec, int inmarkerval, double secondval,
                           int marker, unsigned flag, int flag99, int localshift,
                           unsigned genwidth, int zeropropagate, int *gstr);

/* cpp:1380-1385 calltrackpossible<false,false>: emission e(marker, g, flag2, shift). */
double cnf2o_emission(const cnf2o_ped *P, int ind, int marker, int g, int flag2, int shift);

/* mapval of cpp:5511-5512: zero-propagate call, returns the allele-2 dosage class. */
int cnf2o_mapval(const cnf2o_ped *P, int ind, int marker, int g, int flag2, int shift,
                 double *outval);

/*
 * Forward/backward store for one individual over markers [first,last]
 * (cpp:392-394): fwbw[s][m][slot][g], slot 0 = alpha-minus, 1 = beta,
 * 2 = alpha; fwbwfactors[s][m][slot].
 * Layout here: fwbw[((s*n_markers + m)*3 + slot)*64 + g] with m the global
 * marker index; factors[(s*n_markers + m)*3 + slot].
 */
typedef struct cnf2o_fwbw {
    int     n_markers;
    double *fwbw;
    double *factors;
} cnf2o_fwbw;

cnf2o_fwbw *cnf2o_fwbw_new(int n_markers);
void        cnf2o_fwbw_free(cnf2o_fwbw *W);

/* cpp:2074-2120 initfwbw(domask=3) for one shift mode. */
void cnf2o_initfwbw(const cnf2o_ped *P, int ind, int shift, int first, int last, cnf2o_fwbw *W);

/* cpp:2122-2131 + 1936-2032 with NONESTOP, flag2=-1: total log-likelihood of shift mode. */
double cnf2o_total(const cnf2o_ped *P, int ind, int shift, int first, int last,
                   const cnf2o_fwbw *W, double minfactor);

/* Same with classicstop(-1000-marker, g), path flag2, noneturner (HOT LOOP 2 query, cpp:5499). */
double cnf2o_query(const cnf2o_ped *P, int ind, int shift, int first, int last, int marker,
                   int g, int flag2, const cnf2o_fwbw *W, double minfactor);

/* classicstop(q,-1) through aroundturner(turn) (HOT LOOP 3, cpp:5721).  Runs the
 * backward pass of the turned shift mode if needed (W must hold all 8 modes, see
 * cnf2o_sweep_ind). */
double cnf2o_turn_query(const cnf2o_ped *P, int ind, int shift, int first, int last, int marker,
                        int turn, const cnf2o_fwbw *W, double minfactor);

/*
 * doit per-individual body, cpp:5294-5403 (+ HOT LOOP 2 reduced to the
 * genotypereporter row, cpp:5406-5553).
 *   factors_out[8]  per shift mode (-1e30 for ignored ones)
 *   *factor_out     logsumexp
 *   dosage_out      [last-first+1][3], un-normalised sums of val by mapval, or NULL
 *   W               optional caller-provided store that is left filled
 * mode: 0 = full (g,s,flag2) fan-out with ignoreflag2 exactly as the reference,
 *       1 = fan-out with only the flag2ignore mask (stage 1 of ignoreflag2),
 *       2 = closed form from alpha-minus/beta and the class-split rank-2 emission tables,
 *           summed over the tied phase of ancestors that occupy several slots (== mode 0)
 *       3 = closed form without the tie rule (== mode 1)
 * returns 0 if the individual would be skipped (cpp:5403), 1 otherwise.
 */
int cnf2o_sweep_ind(const cnf2o_ped *P, int ind, int gen, int first, int last,
                    double *factors_out, double *factor_out, double *dosage_out,
                    int mode, cnf2o_fwbw *W);

/* Rank-2 emission tables (own formulation, validated against cnf2o_emission):
 * for shift s and marker m fills c[2], A[2][8], B[2][8] with
 *   e(g) = sum_f c[f]*A[f][g&7]*B[f][g>>3]
 * and the class-split companions A1,B1 (part whose top-of-line allele is 2)
 * restricted to paths allowed by flag2ignore (bits set = forced to 0), plus
 * the restricted totals Ar,Br,cr.  Used by mode 2 and by the tests. */
typedef struct cnf2o_emtab {
    double c[2], A[2][8], B[2][8];
    double cr[2], Ar[2][8], Br[2][8], A1[2][8], B1[2][8];
    int    rootclass[2];   /* dosage contributed at root level when pars[0] is missing */
} cnf2o_emtab;
void cnf2o_emission_tables(const cnf2o_ped *P, int ind, int marker, int shift,
                           int flag2ignore, cnf2o_emtab *out);

/* HOT LOOP 2 terms at one marker: out[8][64][128] (see cnf2_oracle.c). */
void cnf2o_val_table(const cnf2o_ped *P, int ind, int gen, int first, int last, int marker,
                     double *out, int *mapval_out);

/* HAPLOS accumulators of HOT LOOP 2 at one marker: out[n_rec][2] (see cnf2_oracle.c). */
void cnf2o_haplos_row(const cnf2o_ped *P, int ind, int gen, int first, int last, int marker,
                      double *out);

/* thread-private infprobs [n_rec][2][2] and the homozyg[2] increments HOT LOOP 2 makes at one marker
 * (cpp:5513-5577); see the .c file */
void cnf2o_infprobs_row(const cnf2o_ped *P, int ind, int gen, int first, int last, int marker,
                        double *inf_out, double *hz_out);

/* individ::descendants (cpp:3224-3255), desc[n_rec] */
void cnf2o_descendants(const cnf2o_ped *P, int32_t *desc);

/* HOT LOOP 2 + moveinfprobs / movehaplos (cpp:5416-5577, 5876-5902, 3577-3616) for a list of individuals; see .c */
void cnf2o_accumulate(const cnf2o_ped *P, const int *inds, const int *gens, int n_ind, int first, int last,
                      const int32_t *desc, double *inf_out, double *hb_out, double *hc_out, double *hz_out);

/* individ::addvariance (cpp:1489-1558); returns 0 (and leaves *out) when all terms are zero */
int cnf2o_addvariance(const cnf2o_ped *P, int rec, int marker, int flag2ignore, double *out);

/* Batch driver used as the CPU baseline: OpenMP over individuals (cpp:5294),
 * per-thread private store.  inds[n_ind] record indices, gens[n_ind].
 * dosage_out [n_ind][last-first+1][3] normalised rows (or NULL).
 * mode as in cnf2o_sweep_ind (2 for timing).  Returns number of threads used. */
int cnf2o_sweep_batch(const cnf2o_ped *P, const int *inds, const int *gens, int n_ind,
                      int first, int last, double *factors_out, double *factor_out,
                      double *dosage_out, int mode, int n_threads);

/* ---- per-iteration parameter updates (cnf2_oracle_iter.c; PARITY UNPINNED, see the header of that file) ---- */
double cnf2o_caplogitchange(double intended, double orig, double epsilon, int *hitnnn, int breakathalf);
int    cnf2o_processinfprobs(const double inf_in[2], const int present[2], int side, int curmarker, double cursure,
                             int has_prior, int priorval, double priorsure, int empty, int children,
                             double scalefactor, double entropyfactor, int *hitnnn, double inf_out[2],
                             int *out_allele, double *out_sure);
void   cnf2o_relskew_ratio(const double *haploweight, const double *relhaplo, int firstmarker, int endmarker,
                           double *ratio);
void   cnf2o_updatehaploweights(int n_chrom, const int *chromstarts, double *haploweight, double *haplobase,
                                double *haplocount, const int32_t *allele, const double *sure, const double *relhaplo,
                                int children, int descendants, double scalefactor, double entropyfactor, int *hitnnn);
double cnf2o_scalefactor_step(double scalefactor, int hitnnn, int *old, int n_dous);
double cnf2o_gauss15_reciprocal_linear(double slope, double icpt, double a, double b);

#ifdef __cplusplus
}
#endif
#endif
