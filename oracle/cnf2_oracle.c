/*
 * cnf2_oracle.c -- TEST INFRASTRUCTURE ONLY (see cnf2_oracle.h).
 *
 * Plain-C restatement of the cnF2freq forward-backward hot path, written to
 * follow the reference's control flow statement by statement so that it can
 * be audited against it.  Citations: cpp: = /root/reference/cnF2freq.cpp,
 * set: = /root/reference/settings.h.  No code is shared with the HIP product.
 */
#include "cnf2_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NUMTYPES 64
#define NUMSHIFTS 8
#define NUMPATHS 128
#define TYPEBITS 6
#define UNKNOWN 0
#define SEXMARKER 9 /* cpp:226 */

static const int TYPESEXES[TYPEBITS] = {0, 0, 1, 1, 0, 1}; /* set:21 */
static const int TYPEGENS[TYPEBITS]  = {1, 0, 0, 1, 0, 0}; /* set:23 */

/* ---------------------------------------------------------------- helpers */

/* cpp:303-316 markermiss<zeropropagate>; *a may be bound to b. */
static int markermiss(int zeropropagate, int *a, int b)
{
    if (zeropropagate == 1) return 0;                          /* ZERO_PROPAGATE only; NO_EQUIVALENCE (-1) matches */
    if (*a == UNKNOWN) {
        if (!zeropropagate) *a = b;                            /* NO_EQUIVALENCE does not bind either */
        return 0;
    }
    if (b == UNKNOWN && *a != SEXMARKER) return 0;
    return *a != b;
}

/* cpp:321-329 */
static int upflagit(int flag, int parnum, unsigned genwidth)
{
    if (flag < 0) return flag;
    flag >>= parnum * (genwidth - 1);
    flag &= ((1 << (genwidth - 1)) - 1);
    return flag;
}

static const int32_t *rec_allele(const cnf2o_ped *P, int rec, int marker)
{
    return P->allele + ((size_t)rec * P->n_markers + marker) * 2;
}
static const double *rec_sure(const cnf2o_ped *P, int rec, int marker)
{
    return P->sure + ((size_t)rec * P->n_markers + marker) * 2;
}

/* ------------------------------------------------------------- emission */

/* Side-effect sink of the update modes (trackpossibleparams + tb.haplos / tb.infprobs, cpp:559-571,
 * 1347-1354).  Modes (cpp:792-795): HAPLOS = 1: haplos[rec][phase] += updateval; GENOS = 2:
 * infprobs[rec][allele index][markerval] += updateval; HOMOZYGOUS = 4 and GENOSPROBE = 8 only change what
 * is evaluated.  The mode travels as an argument because the recursion strips HOMOZYGOUS (cpp:1280,1322). */
typedef struct {
    double  updateval;
    double *haplos;      /* [n_rec][2]            (HAPLOS) */
    double *infprobs;    /* [n_rec][2][2], last index = markerval - 1 for markerval 1, 2 (GENOS) */
} tp_sink;
#define UPD_HAPLOS 1
#define UPD_GENOS 2
#define UPD_HOMOZYGOUS 4
#define UPD_GENOSPROBE 8

static double recurse_tp(const cnf2o_ped *P, int mother, int markerval, double secondval,
                         int marker, int upflag, int upflag2, int upshift, unsigned genwidth,
                         int firstpar, int zeropropagate, int *gstr, int update, const tp_sink *sink);
static double tp_core(const cnf2o_ped *P, int rec, int inmarkerval, double secondval,
                      int marker, unsigned flag, int flag99, int localshift,
                      unsigned genwidth, int zeropropagate, int *gstr, int update, const tp_sink *sink);

/* cpp:1075-1359, update == 0 (no HAPLOS/GENOS/HOMOZYGOUS/GENOSPROBE side effects),
 * SELFING = RELSKEWSTATES = DOIMPOSSIBLE = false (set:11-16). */
double cnf2o_trackpossible(const cnf2o_ped *P, int rec, int inmarkerval, double secondval,
                           int marker, unsigned flag, int flag99, int localshift,
                           unsigned genwidth, int zeropropagate, int *gstr)
{
    return tp_core(P, rec, inmarkerval, secondval, marker, flag, flag99, localshift, genwidth,
                   zeropropagate, gstr, 0, NULL);
}

/* cpp:1075-1359 with update in {0, HAPLOS, GENOS, HOMOZYGOUS, GENOSPROBE}. */
static double tp_core(const cnf2o_ped *P, int rec, int inmarkerval, double secondval,
                      int marker, unsigned flag, int flag99, int localshift,
                      unsigned genwidth, int zeropropagate, int *gstr, int update, const tp_sink *sink)
{
    const int rootgen  = (genwidth == (1u << (CNF2O_NUMGEN - 1)));          /* cpp:1116 */
    const int attopnow = !(update & UPD_HOMOZYGOUS) &&
                         ((genwidth == 1) || P->founder[rec]);               /* cpp:1120 (HAPLOTYPING==1) */
    const int32_t *themarker     = rec_allele(P, rec, marker);               /* cpp:1137 */
    const double  *themarkersure = rec_sure(P, rec, marker);                 /* cpp:1138 */
    const double   haploweight   = P->hw[(size_t)rec * P->n_markers + marker];

    int upflag2 = -1;                                                        /* cpp:1133 */
    const int upflag  = (int)(flag >> 1);
    const int upshift = localshift >> 1;
    int f2s = 0, f2end = 2;

    if (flag99 != -1 && genwidth > 0) {                                      /* cpp:1141-1146 */
        upflag2 = flag99 >> 1;
        f2s     = flag99 & 1;
        f2end   = (flag99 & 1) + 1;
    }

    const int firstpar = flag & 1;                                           /* cpp:1156 */
    double ok = 0;

    for (int flag2 = f2s; flag2 < f2end; flag2++) {                          /* cpp:1166 */
        int f2n = flag2 & 1;
        const int allthesame = themarker[0] == themarker[1];                 /* cpp:1191 */
        int markerval = inmarkerval;
        double baseval;
        const int realf2n = f2n;
        double mainsecondval = 0;

        if (markermiss(zeropropagate, &markerval, themarker[f2n])) {         /* cpp:1198-1202 */
            baseval = themarkersure[f2n];
            if (themarkersure[f2n] && secondval)
                mainsecondval = (1.0 - themarkersure[f2n]) * secondval;
        } else {                                                             /* cpp:1203-1210 */
            double effectivesecondval =
                (inmarkerval == UNKNOWN && markerval != UNKNOWN) ? 1 : secondval;
            baseval = 1.0 - themarkersure[f2n];
            double effectivemarkersure = (themarker[f2n] == UNKNOWN ? 1 : themarkersure[f2n]);
            mainsecondval = effectivemarkersure * effectivesecondval;
        }

        /* cpp:1213: `update & (GENOS || GENOSPROBE)` is `update & 1`, i.e. true in HAPLOS mode */
        if (attopnow || (update & 1)) {                                      /* cpp:1213-1221 */
            baseval += mainsecondval;
            mainsecondval = 0;
        } else {
            if (mainsecondval) mainsecondval /= baseval;
        }

        int doupdatehaplo = 1;                                               /* cpp:1224 */
        f2n ^= ((firstpar ^ localshift) & 1);                                /* cpp:1227 */

        if (zeropropagate || !genwidth) {                                    /* cpp:1229-1233 */
            baseval *= 0.5;
            doupdatehaplo = 0;
        } else if (allthesame &&
                   (P->correction_inference || (themarkersure[0] == themarkersure[1]))) {
            baseval *= (f2n ? 1.0 : 0.0);                                    /* cpp:1235-1239 */
            doupdatehaplo = 0;
        } else {
            baseval *= fabs((f2n ? 1.0 : 0.0) - haploweight);                /* cpp:1245 */
        }

        if (baseval && (attopnow || P->par[rec * 2 + firstpar] < 0)) {       /* cpp:1260-1268 */
            if (zeropropagate && gstr) *gstr += (themarker[realf2n] == 2);
        }

        if (!baseval || attopnow) {                                          /* cpp:1271 */
        } else {
            const int down = update & ~UPD_HOMOZYGOUS;                       /* cpp:1280, 1322 */
            if ((!zeropropagate || rootgen) && !(update & UPD_GENOS)) {      /* cpp:1291 (prelok always true) */
                double secsecondval = 0;
                int secmark = themarker[!realf2n];
                if (!(update & UPD_HOMOZYGOUS)) {
                    if (themarkersure[!realf2n]) {                           /* cpp:1298-1302 */
                        baseval *= (1 - themarkersure[!realf2n]);
                        secsecondval = themarkersure[!realf2n] / (1 - themarkersure[!realf2n]);
                    }
                } else if (markerval != secmark) {                           /* cpp:1304-1313 */
                    if (secmark != UNKNOWN) baseval *= themarkersure[!realf2n];
                    secmark = markerval;
                } else {                                                     /* cpp:1314-1318 */
                    baseval *= (1 - themarkersure[!realf2n]);
                }
                baseval *= recurse_tp(P, rec, secmark, secsecondval, marker, upflag, upflag2,
                                      upshift, genwidth, !firstpar, zeropropagate, gstr, down, sink); /* cpp:1322 */
            }
            if (baseval)                                                     /* cpp:1336-1340 */
                baseval *= recurse_tp(P, rec, markerval, mainsecondval, marker, upflag, upflag2,
                                      upshift, genwidth, firstpar, zeropropagate, gstr, down, sink);
        }

        if (baseval) {                                                       /* cpp:1343-1354 */
            ok += baseval;
            if ((update & UPD_HAPLOS) && doupdatehaplo) sink->haplos[rec * 2 + f2n] += sink->updateval;
            if ((update & UPD_GENOS) && (markerval == 1 || markerval == 2))
                sink->infprobs[(rec * 2 + realf2n) * 2 + (markerval - 1)] += sink->updateval;
        }
    }
    return ok;
}

/* cpp:955-1058 recursetrackpossible: ctor (984-986) + operator double (1035-1057). */
static double recurse_tp(const cnf2o_ped *P, int mother, int markerval, double secondval,
                         int marker, int upflag, int upflag2, int upshift, unsigned genwidth,
                         int firstpar, int zeropropagate, int *gstr, int update, const tp_sink *sink)
{
    int upflagr  = upflagit(upflag, firstpar, genwidth);
    int upflag2r = upflagit(upflag2, firstpar, genwidth);      /* NUMGEN-NUMFLAG2GEN == 0 */
    int upshiftr = upflagit(upshift, firstpar, genwidth >> 1); /* NUMGEN-NUMSHIFTGEN == 1 */
    int par = P->par[mother * 2 + firstpar];
    if (par < 0) return 1 + secondval;                          /* cpp:1043-1046 */
    return tp_core(P, par, markerval, secondval, marker, (unsigned)upflagr, upflag2r,
                   upshiftr, genwidth >> 1, zeropropagate, gstr, update, sink);
}

/* cpp:1380-1385 */
double cnf2o_emission(const cnf2o_ped *P, int ind, int marker, int g, int flag2, int shift)
{
    return cnf2o_trackpossible(P, ind, UNKNOWN, 0, marker, (unsigned)(g * 2), flag2, shift,
                               1u << (CNF2O_NUMGEN - 1), 0, NULL);
}

/* cpp:5511-5512 */
int cnf2o_mapval(const cnf2o_ped *P, int ind, int marker, int g, int flag2, int shift,
                 double *outval)
{
    int mapval = 0;
    double v = cnf2o_trackpossible(P, ind, UNKNOWN, 0, marker, (unsigned)(g * 2), flag2, shift,
                                   1u << (CNF2O_NUMGEN - 1), 1, &mapval);
    if (outval) *outval = v;
    return mapval;
}

/* ------------------------------------------------------------ fixtrees */

static void relmap_or(cnf2o_tree *T, int rec, int mapbits, int shiftbits)
{
    for (int i = 0; i < T->n_rel; i++) {
        if (T->rel_rec[i] == rec) {
            T->rel_map[i] |= mapbits;
            T->rel_mapshift[i] |= shiftbits;
            return;
        }
    }
    T->rel_rec[T->n_rel]      = rec;
    T->rel_map[T->n_rel]      = mapbits;
    T->rel_mapshift[T->n_rel] = shiftbits;
    T->n_rel++;
}

/* cpp:3099-3187 */
void cnf2o_fixtrees(const cnf2o_ped *P, int ind, cnf2o_tree *T)
{
    int flag2ignore = 0, shiftignore = 0;
    memset(T, 0, sizeof(*T));
    for (int i = 0; i < 7; i++) T->ordered[i] = -1;
    T->ordered[0] = ind;
    relmap_or(T, ind, 1, 1);                                   /* cpp:3112-3113 */

    flag2ignore = 1;                                           /* cpp:3117 */
    int anylev1 = 0;
    for (int lev1 = 0; lev1 < 2; lev1++) {
        int lev1i = P->par[ind * 2 + lev1];
        if (lev1i < 0) continue;
        int flag2index = 1 + lev1 * ((1 << (CNF2O_NUMGEN - 1)) - 1); /* cpp:3124 */
        int shiftval   = 2 << lev1;                            /* cpp:3125 */
        if (!P->empty[lev1i]) {                                /* cpp:3127-3133 */
            flag2ignore |= 1 << flag2index;
            relmap_or(T, lev1i, 1 << flag2index, shiftval);
            T->ordered[flag2index] = lev1i;
        }
        int anypars = 0;
        for (int lev2 = 0; lev2 < 2; lev2++) {                 /* cpp:3139-3153 */
            int lev2i = P->par[lev1i * 2 + lev2];
            if (lev2i < 0) continue;
            if (!P->empty[lev2i]) {
                flag2ignore |= 1 << (flag2index + lev2 + 1);
                relmap_or(T, lev2i, 1 << (flag2index + lev2 + 1), 0);
                T->ordered[flag2index + lev2 + 1] = lev2i;
                anypars = 1;
            }
        }
        if (anypars) shiftignore |= shiftval;                  /* cpp:3156-3159 */
        if (anypars || !P->empty[lev1i]) anylev1 = 1;          /* cpp:3165-3168 */
    }
    if (anylev1) shiftignore |= 1;                             /* cpp:3170-3177 */
    else T->founder = 1;
    flag2ignore ^= (NUMPATHS - 1);                             /* cpp:3178-3179 */
    shiftignore ^= (NUMSHIFTS - 1);
    T->flag2ignore = flag2ignore;
    T->shiftignore = shiftignore;
}

void cnf2o_founder_flags(const cnf2o_ped *P, uint8_t *founder_out)
{
    /* postmarkerdata calls fixtrees(ind) for every individual (cpp:3373-3389), whose only
     * persistent effect is ind->founder = true when no parent is informative (cpp:3174-3177).
     * fixtrees does not read `founder`, so the order of evaluation is irrelevant. */
    for (int r = 0; r < P->n_rec; r++) {
        cnf2o_tree T;
        cnf2o_fixtrees(P, r, &T);
        founder_out[r] = (uint8_t)(P->founder ? (P->founder[r] | T.founder) : T.founder);
    }
}

/* cpp:3462-3496 (q <= -1000 form: marker = -q-1000 >= 0) */
int cnf2o_ignoreflag2(const cnf2o_ped *P, const cnf2o_tree *T, int flag2, int g,
                      int shiftflagmode, int marker)
{
    const int flag2filter = (1 << 30) - 1;
    if (flag2 & (T->flag2ignore & flag2filter)) return 1;      /* cpp:3478 */
    for (int i = 0; i < T->n_rel; i++) {
        int currfilter = T->rel_map[i] & flag2filter;
        int filtered   = ((flag2 ^ (g * 2)) & currfilter);
        if (filtered && filtered != currfilter) return 1;      /* cpp:3486 */
        if (marker >= 0) {
            const int32_t *a = rec_allele(P, T->rel_rec[i], marker);
            const double  *s = rec_sure(P, T->rel_rec[i], marker);
            if (a[0] == a[1] && s[0] == s[1] &&
                !((filtered != 0) ^ ((shiftflagmode & T->rel_mapshift[i]) != 0)))
                return 1;                                      /* cpp:3488-3493 */
        }
    }
    return 0;
}

/* ----------------------------------------------------- HMM: adjust, step */

/* cpp:1579-1670 */
static void adjustprobs(const cnf2o_ped *P, int ind, int shift, double *probs, int marker,
                        double *factor, int flag99)
{
    double sum = 0;
    for (int i = 0; i < NUMTYPES; i++) {
        double val = probs[i];
        if (val < 1e-300) {                                    /* cpp:1607-1611 */
            probs[i] = 0;
            continue;
        }
        double realok = cnf2o_emission(P, ind, marker, i, flag99, shift); /* cpp:1615-1618 */
        val *= realok;                                         /* cpp:1622-1625 */
        sum += val;
        probs[i] = val;
    }
    if (sum <= 0) {                                            /* cpp:1656-1660 */
        *factor = CNF2O_MINFACTOR;
    } else {
        for (int i = 0; i < NUMTYPES; i++) probs[i] /= sum;    /* cpp:1664-1668 */
        *factor += log(sum);
    }
}

/* cpp:2270-2367: dense XOR-indexed transition over distance dist (applied only when dist > 0). */
static void transition(const cnf2o_ped *P, double *probs, double dist)
{
    if (!(dist > 0)) return;                                   /* cpp:2273 */
    double probs2[NUMTYPES] = {0};
    double recprob[2][2], other[2][2][2], recombprec[NUMTYPES];
    for (int gen = 0; gen < 2; gen++)
        for (int k = 0; k < 2; k++)
            recprob[gen][k] = 0.5 * (1.0 - exp(P->genrec[gen] * dist)); /* cpp:2286, getactrec cpp:777-780 */
    for (int gen = 0; gen < 2; gen++)
        for (int m = 0; m < 2; m++)
            for (int k = 0; k < 2; k++) {
                double prob = recprob[gen][k];
                if (m) prob = 1.0 - prob;
                other[gen][m][k] = prob;                       /* cpp:2294-2306 */
            }
    for (int index = 0; index < NUMTYPES; index++) recombprec[index] = 1;
    for (int t = 0; t < TYPEBITS; t++) {                       /* cpp:2329-2340 */
        int sex = TYPESEXES[t], gen = TYPEGENS[t];
        for (int index = 0; index < NUMTYPES; index++) {
            int val = !((index >> t) & 1);
            recombprec[index] *= other[gen][val][sex];
        }
    }
    for (int from = 0; from < NUMTYPES; from++) {              /* cpp:2352-2364 */
        double fromval = probs[from];
        if (fromval <= 0) continue;
        for (int to = 0; to < NUMTYPES; to++) probs2[to] += fromval * recombprec[from ^ to];
    }
    memcpy(probs, probs2, sizeof(probs2));
}

/* ------------------------------------------------------------- fw/bw store */

struct fwbw_priv {
    cnf2o_fwbw pub;
    int done[NUMSHIFTS]; /* cpp:394 fwbwdone, here just the domask bits */
};

cnf2o_fwbw *cnf2o_fwbw_new(int n_markers)
{
    struct fwbw_priv *W = (struct fwbw_priv *)calloc(1, sizeof(*W));
    W->pub.n_markers = n_markers;
    W->pub.fwbw    = (double *)calloc((size_t)NUMSHIFTS * n_markers * 3 * NUMTYPES, sizeof(double));
    W->pub.factors = (double *)calloc((size_t)NUMSHIFTS * n_markers * 3, sizeof(double));
    return &W->pub;
}

void cnf2o_fwbw_free(cnf2o_fwbw *W)
{
    if (!W) return;
    free(W->fwbw);
    free(W->factors);
    free(W);
}

static void fwbw_reset(cnf2o_fwbw *W) /* generation++ / resizecaches, cpp:5302-5306 */
{
    memset(((struct fwbw_priv *)W)->done, 0, sizeof(int) * NUMSHIFTS);
}

static double *slot(const cnf2o_fwbw *W, int s, int m, int pad)
{
    return W->fwbw + (((size_t)s * W->n_markers + m) * 3 + pad) * NUMTYPES;
}
static double *fslot(const cnf2o_fwbw *W, int s, int m, int pad)
{
    return W->factors + ((size_t)s * W->n_markers + m) * 3 + pad;
}
static void savefwbw(const cnf2o_fwbw *W, int s, int m, int pad, const double *probs, double factor)
{
    memcpy(slot(W, s, m, pad), probs, sizeof(double) * NUMTYPES); /* cpp:2172-2178 */
    *fslot(W, s, m, pad) = factor;
}

/* cpp:2145-2418 with updateend = STORE|FORWARD|1, NONESTOP, noneturner, flag2 = -1. */
static void forward_store(const cnf2o_ped *P, int ind, int shift, int first, int last,
                          const cnf2o_fwbw *W)
{
    double probs[NUMTYPES], factor = 0;
    for (int i = 0; i < NUMTYPES; i++) probs[i] = 1.0 / NUMTYPES;  /* cpp:2100-2104 EVENGEN */
    for (int j = first + 1; j != last + 1; j++) {                   /* cpp:2193 */
        savefwbw(W, shift, j - 1, 0, probs, factor);                /* cpp:2211-2214 */
        adjustprobs(P, ind, shift, probs, j - 1, &factor, -1);      /* cpp:2238 */
        savefwbw(W, shift, j - 1, 2, probs, factor);                /* cpp:2248-2251 */
        transition(P, probs, P->pos[j] - P->pos[j - 1]);            /* cpp:2270-2367 */
    }
    savefwbw(W, shift, last, 0, probs, factor);                     /* cpp:2403-2406 */
    adjustprobs(P, ind, shift, probs, last, &factor, -1);           /* cpp:2408 */
    savefwbw(W, shift, last, 2, probs, factor);                     /* cpp:2410-2413 */
}

/* Same with updateend = STORE|BACKWARD|1. */
static void backward_store(const cnf2o_ped *P, int ind, int shift, int first, int last,
                           const cnf2o_fwbw *W)
{
    double probs[NUMTYPES], factor = 0;
    for (int i = 0; i < NUMTYPES; i++) probs[i] = 1.0;              /* cpp:2111-2114 */
    savefwbw(W, shift, last, 1, probs, factor);                     /* cpp:2186-2189 */
    for (int j = last - 1; j != first - 1; j--) {                   /* cpp:2193, d = -1 */
        adjustprobs(P, ind, shift, probs, j + 1, &factor, -1);      /* cpp:2238, marker j-d */
        transition(P, probs, P->pos[j + 1] - P->pos[j]);            /* cpp:2199,2270 */
        savefwbw(W, shift, j, 1, probs, factor);                    /* cpp:2393-2396 */
    }
    /* cpp:2408: the closing adjustprobs(first) only affects the unused return value. */
}

static void initfwbw_mask(const cnf2o_ped *P, int ind, int shift, int first, int last,
                          cnf2o_fwbw *W, int domask)
{
    struct fwbw_priv *Wp = (struct fwbw_priv *)W;                   /* cpp:2076-2083 */
    domask &= ~Wp->done[shift];
    if (domask & 1) forward_store(P, ind, shift, first, last, W);
    if (domask & 2) backward_store(P, ind, shift, first, last, W);
    Wp->done[shift] |= domask;
}

void cnf2o_initfwbw(const cnf2o_ped *P, int ind, int shift, int first, int last, cnf2o_fwbw *W)
{
    ((struct fwbw_priv *)W)->done[shift] = 0;
    initfwbw_mask(P, ind, shift, first, last, W, 3);
}

/* ----------------------------------------------------------------- queries */

/* Closing part of quickanalyze, cpp:1994-2018. */
static double finish_query(double *probs, double factor, const cnf2o_fwbw *W, int s, int m)
{
    const double *beta = slot(W, s, m, 1);
    double sum = 0;
    for (int k = 0; k < NUMTYPES; k++) {
        probs[k] *= beta[k];
        sum += probs[k];
    }
    factor += *fslot(W, s, m, 1);
    if (sum <= 0) {
        factor = CNF2O_MINFACTOR;
    } else {
        sum = 1 / sum;
        for (int i = 0; i < NUMTYPES; i++) probs[i] *= sum;
        factor -= log(sum);
    }
    return factor;
}

/* doanalyze<noneturner>(NONESTOP, flag2=-1): cpp:2122-2131 -> 1936-2032.  The marker search
 * (cpp:1945-1959) ends at startmark = endmark; pad = 2; realanalyze<4|2> then performs one
 * emission-free step towards markerposes[endmark+1] (cpp:1984, 2193-2195).  That read is one
 * past the chromosome (past the vector for the last one); the step is sum-preserving and
 * beta_end = 1, so it cannot change the result beyond rounding.  The restatement applies the
 * transition when a next marker exists, exactly as the reference would. */
double cnf2o_total(const cnf2o_ped *P, int ind, int shift, int first, int last,
                   const cnf2o_fwbw *W, double minfactor)
{
    (void)ind; (void)first;
    double probs[NUMTYPES];
    double factor = *fslot(W, shift, last, 2);                      /* cpp:1963 */
    memcpy(probs, slot(W, shift, last, 2), sizeof(probs));          /* cpp:1964 */
    if (factor < minfactor) return factor;                          /* cpp:1967 */
    if (last + 1 < P->n_markers) transition(P, probs, P->pos[last + 1] - P->pos[last]);
    return finish_query(probs, factor, W, shift, last);
}

/* doanalyze<noneturner>(classicstop(-1000-marker, g), flag2): one realanalyze<4> step
 * (cpp:1980): filter to g (cpp:2229-2232), emission for path flag2 (cpp:2238), turner no-op,
 * filter again (cpp:2262-2266). */
double cnf2o_query(const cnf2o_ped *P, int ind, int shift, int first, int last, int marker,
                   int g, int flag2, const cnf2o_fwbw *W, double minfactor)
{
    (void)first; (void)last;
    double probs[NUMTYPES];
    double factor = *fslot(W, shift, marker, 0);                    /* pad = 0, cpp:1960-1964 */
    memcpy(probs, slot(W, shift, marker, 0), sizeof(probs));
    if (factor < minfactor) return factor;
    double f2 = 0;
    for (int i = 0; i < NUMTYPES; i++) probs[i] *= (i == g);
    adjustprobs(P, ind, shift, probs, marker, &f2, flag2);
    for (int i = 0; i < NUMTYPES; i++) probs[i] *= (i == g);
    factor += f2;
    if (factor < minfactor) return factor;                          /* cpp:1992 */
    return finish_query(probs, factor, W, shift, marker);
}

/* doanalyze<aroundturner>(classicstop(q,-1), -1): pad = 2 (alpha incl. emission), no
 * emission (updateend&2), turner permutes states by XOR and flips shift bits
 * (cpp:504-511,530-552), then beta of the *new* shift mode (cpp:1988-2000). */
double cnf2o_turn_query(const cnf2o_ped *P, int ind, int shift, int first, int last, int marker,
                        int turn, const cnf2o_fwbw *W, double minfactor)
{
    double probs[NUMTYPES], probs2[NUMTYPES];
    double factor = *fslot(W, shift, marker, 2);
    memcpy(probs, slot(W, shift, marker, 2), sizeof(probs));
    if (factor < minfactor) return factor;
    int xorturn       = turn & 54;                                  /* cpp:508 */
    int flagmodeshift = (turn >> TYPEBITS) | ((turn & 1) ? 2 : 0) | ((turn & 8) ? 4 : 0);
    for (int i = 0; i < NUMTYPES; i++) probs2[i ^ xorturn] = probs[i];
    int newshift = shift ^ flagmodeshift;                           /* cpp:551 */
    initfwbw_mask(P, ind, newshift, first, last, (cnf2o_fwbw *)W, 2); /* cpp:1988 */
    if (factor < minfactor) return factor;
    return finish_query(probs2, factor, W, newshift, marker);
}

/* ------------------------------------------------- rank-2 emission tables */

/*
 * Own closed form of the emission (not in the reference; verified against
 * cnf2o_emission/cnf2o_mapval in tests):  at the root flag = 2g so firstpar = 0
 * (cpp:1156,1383): allele f goes to pars[0] with the low 3 state bits, the other
 * allele to pars[1] with the high 3 bits, hence
 *     e(g) = sum_f c[f] * A[f][g&7] * B[f][g>>3].
 * path_mask restricts which allele index a window slot may use (bit set in
 * flag2ignore => only index 0, i.e. the flag2 & flag2ignore test of cpp:3478);
 * `cls` selects the part of a line whose top allele equals 2 (mapval, cpp:1260-1268).
 */
typedef struct {
    double tot;  /* sum over allowed paths           */
    double two;  /* part with top-of-line allele == 2 */
} linev;

/* Ancestor at the top of a line (genwidth==1 or founder): value of trackpossible there. */
static linev top_eval(const cnf2o_ped *P, int rec, int marker, int inmv, double secondval,
                      int firstpar, int localshift, int restrict0, int forced)
{
    linev r = {0, 0};
    const int32_t *m = rec_allele(P, rec, marker);
    const double  *s = rec_sure(P, rec, marker);
    double hw = P->hw[(size_t)rec * P->n_markers + marker];
    int allsame = m[0] == m[1];
    for (int f = 0; f < (restrict0 ? 1 : 2); f++) {
        int mv = inmv;
        double baseval, msv = 0;
        /* all-or-none rule of ignoreflag2 (cpp:3484-3486): a tied slot only keeps the
           allele index with (f ^ firstpar) == forced */
        if (forced >= 0 && ((f ^ firstpar) & 1) != forced) continue;
        if (markermiss(0, &mv, m[f])) {
            baseval = s[f];
            if (s[f] && secondval) msv = (1.0 - s[f]) * secondval;
        } else {
            double esv = (inmv == UNKNOWN && mv != UNKNOWN) ? 1 : secondval;
            baseval = 1.0 - s[f];
            msv = (m[f] == UNKNOWN ? 1 : s[f]) * esv;
        }
        baseval += msv;
        int phase = f ^ ((firstpar ^ localshift) & 1);
        if (allsame && (P->correction_inference || s[0] == s[1])) baseval *= phase ? 1.0 : 0.0;
        else baseval *= fabs((phase ? 1.0 : 0.0) - hw);
        r.tot += baseval;
        if (m[f] == 2) r.two += baseval;
    }
    return r;
}

/* Value of recursetrackpossible towards parent slot `par` of `child` with 3-bit (genwidth 2)
 * or 1-bit (genwidth 1) flag `k`.  slotbit = index of this ancestor's bit in flag2ignore. */
static linev line_eval(const cnf2o_ped *P, int rec, int marker, int inmv, double secondval,
                       int k, int localshift, unsigned genwidth, int flag2ignore, int slotbit,
                       const int *force)
{
    linev r = {0, 0};
    if (rec < 0) { /* cpp:1043-1046: no class information from a missing ancestor */
        r.tot = 1 + secondval;
        return r;
    }
    int restrict0 = (flag2ignore >> slotbit) & 1;
    if (genwidth == 1 || P->founder[rec])
        return top_eval(P, rec, marker, inmv, secondval, k & 1, localshift, restrict0,
                        force ? force[slotbit] : -1);

    /* genwidth == 2 interior parent */
    const int32_t *m = rec_allele(P, rec, marker);
    const double  *s = rec_sure(P, rec, marker);
    double hw = P->hw[(size_t)rec * P->n_markers + marker];
    int allsame  = m[0] == m[1];
    int firstpar = k & 1;
    int upflag   = k >> 1;
    for (int f = 0; f < (restrict0 ? 1 : 2); f++) {
        int mv = inmv;
        double baseval, msv = 0;
        if (force && force[slotbit] >= 0 && ((f ^ firstpar) & 1) != force[slotbit]) continue;
        if (markermiss(0, &mv, m[f])) {
            baseval = s[f];
            if (s[f] && secondval) msv = (1.0 - s[f]) * secondval;
        } else {
            double esv = (inmv == UNKNOWN && mv != UNKNOWN) ? 1 : secondval;
            baseval = 1.0 - s[f];
            msv = (m[f] == UNKNOWN ? 1 : s[f]) * esv;
        }
        if (msv) msv /= baseval;
        int phase = f ^ ((firstpar ^ localshift) & 1);
        if (allsame && (P->correction_inference || s[0] == s[1])) baseval *= phase ? 1.0 : 0.0;
        else baseval *= fabs((phase ? 1.0 : 0.0) - hw);
        if (!baseval) continue;
        double ssv = 0;
        int secmark = m[!f];
        if (s[!f]) {
            baseval *= (1 - s[!f]);
            ssv = s[!f] / (1 - s[!f]);
        }
        int gp_tr = P->par[rec * 2 + firstpar];
        int gp_ot = P->par[rec * 2 + !firstpar];
        linev ot = line_eval(P, gp_ot, marker, secmark, ssv, upflagit(upflag, !firstpar, 2), 0, 1,
                             flag2ignore, slotbit + 1 + !firstpar, force);
        linev tr = line_eval(P, gp_tr, marker, mv, msv, upflagit(upflag, firstpar, 2), 0, 1,
                             flag2ignore, slotbit + 1 + firstpar, force);
        double b = baseval * ot.tot;
        r.tot += b * tr.tot;
        if (gp_tr < 0) {
            /* cpp:1260-1268: !pars[firstpar] => this level is the top of the traced line */
            if (m[f] == 2) r.two += b * tr.tot;
        } else {
            r.two += b * tr.two;
        }
    }
    return r;
}

static void emission_tables_forced(const cnf2o_ped *P, int ind, int marker, int shift,
                                   int flag2ignore, const int *force, cnf2o_emtab *T);

void cnf2o_emission_tables(const cnf2o_ped *P, int ind, int marker, int shift,
                           int flag2ignore, cnf2o_emtab *T)
{
    emission_tables_forced(P, ind, marker, shift, flag2ignore, NULL, T);
}

static void emission_tables_forced(const cnf2o_ped *P, int ind, int marker, int shift,
                                   int flag2ignore, const int *force, cnf2o_emtab *T)
{
    memset(T, 0, sizeof(*T));
    const int32_t *m = rec_allele(P, ind, marker);
    const double  *s = rec_sure(P, ind, marker);
    double hw = P->hw[(size_t)ind * P->n_markers + marker];
    int allsame = m[0] == m[1];
    int attop   = P->founder[ind];
    int p0 = P->par[ind * 2 + 0], p1 = P->par[ind * 2 + 1];
    for (int f = 0; f < 2; f++) {
        /* root: inmarkerval unknown => binds to m[f] (cpp:308-311) */
        double baseval = 1.0 - s[f];
        double msv = (m[f] == UNKNOWN ? 1 : s[f]) * ((m[f] != UNKNOWN) ? 1.0 : 0.0);
        if (attop) {
            baseval += msv;
            msv = 0;
        } else if (msv) msv /= baseval;
        int phase = f ^ (shift & 1);
        if (allsame && (P->correction_inference || s[0] == s[1])) baseval *= phase ? 1.0 : 0.0;
        else baseval *= fabs((phase ? 1.0 : 0.0) - hw);
        double ssv = 0;
        if (!attop && baseval && s[!f]) {
            baseval *= (1 - s[!f]);
            ssv = s[!f] / (1 - s[!f]);
        }
        T->c[f] = baseval;
        T->cr[f] = (f == 1 && (flag2ignore & 1)) ? 0.0 : baseval;
        T->rootclass[f] = (m[f] == 2);
        for (int k = 0; k < 8; k++) {
            if (attop) {
                /* root is the top of its single line (cpp:1260-1271): class = root allele f */
                T->A[f][k] = T->B[f][k] = T->Ar[f][k] = T->Br[f][k] = 1.0;
                T->A1[f][k] = T->rootclass[f] ? 1.0 : 0.0;
                T->B1[f][k] = 0.0;
                continue;
            }
            linev a  = line_eval(P, p0, marker, m[f], msv, k, (shift >> 1) & 1, 2, 0, 1, NULL);
            linev b  = line_eval(P, p1, marker, m[!f], ssv, k, (shift >> 2) & 1, 2, 0, 4, NULL);
            linev ar = line_eval(P, p0, marker, m[f], msv, k, (shift >> 1) & 1, 2, flag2ignore, 1, force);
            linev br = line_eval(P, p1, marker, m[!f], ssv, k, (shift >> 2) & 1, 2, flag2ignore, 4, force);
            T->A[f][k]  = a.tot;
            T->B[f][k]  = b.tot;
            T->Ar[f][k] = ar.tot;
            T->Br[f][k] = br.tot;
            /* cpp:1260-1268 at the root: !pars[firstpar=0] makes the root the top of line 0 */
            T->A1[f][k] = (p0 < 0) ? (T->rootclass[f] ? ar.tot : 0.0) : ar.two;
            T->B1[f][k] = br.two;
        }
    }
}

/* ------------------------------------------------------- per-individual body */

static double logsumexp_factors(const double *factors, int shiftend, int shiftignore, double *maxout)
{
    double factor = -1e15;                                          /* cpp:5373 */
    for (int s = 0; s < shiftend; s++) factor = fmax(factor, factors[s]); /* cpp:5381 */
    double realfactor = 0;
    for (int s = 0; s < shiftend; s++) {                            /* cpp:5385-5390 */
        if (s & shiftignore) continue;
        realfactor += exp(factors[s] - factor);
    }
    if (maxout) *maxout = factor;
    return factor + log(realfactor);                                /* cpp:5400 */
}

int cnf2o_sweep_ind(const cnf2o_ped *P, int ind, int gen, int first, int last,
                    double *factors_out, double *factor_out, double *dosage_out,
                    int mode, cnf2o_fwbw *Win)
{
    cnf2o_fwbw *W = Win ? Win : cnf2o_fwbw_new(P->n_markers);
    fwbw_reset(W);
    cnf2o_tree T;
    cnf2o_fixtrees(P, ind, &T);                                     /* cpp:5329 */
    int shiftend = NUMSHIFTS;
    if (gen < 2) shiftend = 2;                                      /* cpp:5359 */

    double factors[NUMSHIFTS];
    double factor = -1e15;
    for (int s = 0; s < NUMSHIFTS; s++) factors[s] = -1e30;
    for (int s = 0; s < shiftend; s++) {                            /* cpp:5375-5382 */
        if (s & T.shiftignore) factors[s] = -1e30;
        else {
            initfwbw_mask(P, ind, s, first, last, W, 3);
            factors[s] = cnf2o_total(P, ind, s, first, last, W, -40000 + factor);
        }
        factor = fmax(factor, factors[s]);
    }
    factor = logsumexp_factors(factors, shiftend, T.shiftignore, NULL);
    if (factors_out) memcpy(factors_out, factors, sizeof(factors));
    if (factor_out) *factor_out = factor;
    int ok = !(isnan(factor) || factor < CNF2O_MINFACTOR);          /* cpp:5403 */

    if (dosage_out) {
        int nm = last - first + 1;
        memset(dosage_out, 0, sizeof(double) * 3 * nm);
        const double unusualstate = -200;                           /* cpp:5414 */
        for (int m = first; ok && m <= last; m++) {                 /* cpp:5406 */
            double *row = dosage_out + (size_t)(m - first) * 3;
            if (mode == 2 || mode == 3) {
                for (int s = 0; s < shiftend; s++) {
                    if (s & T.shiftignore) continue;
                    if (factor - factors[s] > 40) continue;
                    /* ancestors occupying several window slots: sum over their common
                       (f2n ^ firstpar) value, the all-or-none rule of cpp:3484-3486 */
                    int groups[7], ngroups = 0;
                    for (int i = 0; i < T.n_rel; i++)
                        if (mode == 2 && __builtin_popcount(T.rel_map[i]) > 1)
                            groups[ngroups++] = T.rel_map[i];
                    const double *am = slot(W, s, m, 0), *be = slot(W, s, m, 1);
                    double scale = exp(*fslot(W, s, m, 0) + *fslot(W, s, m, 1) - factor);
                    for (int combo = 0; combo < (1 << ngroups); combo++) {
                        int force[7] = {-1, -1, -1, -1, -1, -1, -1};
                        for (int gi = 0; gi < ngroups; gi++)
                            for (int b = 0; b < 7; b++)
                                if (groups[gi] & (1 << b)) force[b] = (combo >> gi) & 1;
                        cnf2o_emtab E;
                        emission_tables_forced(P, ind, m, s, T.flag2ignore, force, &E);
                        for (int g = 0; g < NUMTYPES; g++) {
                            double w = am[g] * be[g] * scale;
                            if (!(w > 0)) continue;
                            for (int f = 0; f < 2; f++) {
                                double a1 = E.A1[f][g & 7], a = E.Ar[f][g & 7];
                                double b1 = E.B1[f][g >> 3], b = E.Br[f][g >> 3];
                                row[2] += w * E.cr[f] * a1 * b1;
                                row[1] += w * E.cr[f] * (a1 * (b - b1) + (a - a1) * b1);
                                row[0] += w * E.cr[f] * (a - a1) * (b - b1);
                            }
                        }
                    }
                }
                continue;
            }
            for (int g = 0; g < NUMTYPES; g++) {                    /* cpp:5416-5553 */
                for (int s = 0; s < shiftend; s++) {
                    if (s & T.shiftignore) continue;
                    if (factor - factors[s] > 40) continue;         /* cpp:5421 */
                    for (int flag2 = 0; flag2 < NUMPATHS; flag2++) {
                        if (mode == 0) {
                            if (cnf2o_ignoreflag2(P, &T, flag2, g, s, m)) continue;
                        } else if (flag2 & T.flag2ignore) continue;
                        double val = cnf2o_query(P, ind, s, first, last, m, g, flag2, W,
                                                 unusualstate + factor) - factor; /* cpp:5499 */
                        if (isfinite(val) && val > unusualstate) {  /* cpp:5502 */
                            val = exp(val);
                            int mapval = cnf2o_mapval(P, ind, m, g, flag2, s, NULL);
                            if (mapval >= 0 && mapval <= 2) row[mapval] += val; /* cpp:3536 */
                        }
                    }
                }
            }
        }
    }
    if (!Win) cnf2o_fwbw_free(W);
    return ok;
}

/* All the terms of HOT LOOP 2 at one marker (cpp:5416-5508): out[s][g][flag2] = val, or -1 where
 * the reference skips the term (masked shift mode, factor gap > 40, ignoreflag2) and 0 where the
 * query is below the -200 threshold.  mapval_out (optional) receives mapval per entry. */
void cnf2o_val_table(const cnf2o_ped *P, int ind, int gen, int first, int last, int marker,
                     double *out, int *mapval_out)
{
    cnf2o_fwbw *W = cnf2o_fwbw_new(P->n_markers);
    double factors[NUMSHIFTS], factor;
    cnf2o_sweep_ind(P, ind, gen, first, last, factors, &factor, NULL, 0, W);
    cnf2o_tree T;
    cnf2o_fixtrees(P, ind, &T);
    int shiftend = gen < 2 ? 2 : NUMSHIFTS;
    for (int s = 0; s < NUMSHIFTS; s++)
        for (int g = 0; g < NUMTYPES; g++)
            for (int flag2 = 0; flag2 < NUMPATHS; flag2++) {
                size_t idx = ((size_t)s * NUMTYPES + g) * NUMPATHS + flag2;
                out[idx] = -1;
                if (mapval_out) mapval_out[idx] = -1;
                if (s >= shiftend || (s & T.shiftignore)) continue;
                if (factor - factors[s] > 40) continue;
                if (cnf2o_ignoreflag2(P, &T, flag2, g, s, marker)) continue;
                double val = cnf2o_query(P, ind, s, first, last, marker, g, flag2, W, -200 + factor) - factor;
                out[idx] = (isfinite(val) && val > -200) ? exp(val) : 0.0;
                if (mapval_out) mapval_out[idx] = cnf2o_mapval(P, ind, marker, g, flag2, s, NULL);
            }
    cnf2o_fwbw_free(W);
}

/* What HOT LOOP 2 leaves in the thread-private `haplos` for one marker before movehaplos
 * (cpp:5416-5556 -> updatehaplo cpp:1561-1575 -> trackpossible<HAPLOS> cpp:1347-1350):
 * out[n_rec][2], out[r][phase] = sum of val over (g, s, path) whose path uses phase `phase`
 * (allele index ^ firstpar ^ localshift) of individual r; individuals that are homozygous with
 * equal sure at the marker never accumulate. */
void cnf2o_haplos_row(const cnf2o_ped *P, int ind, int gen, int first, int last, int marker,
                      double *out)
{
    cnf2o_fwbw *W = cnf2o_fwbw_new(P->n_markers);
    double factors[NUMSHIFTS], factor;
    cnf2o_sweep_ind(P, ind, gen, first, last, factors, &factor, NULL, 0, W);
    cnf2o_tree T;
    cnf2o_fixtrees(P, ind, &T);
    memset(out, 0, sizeof(double) * 2 * P->n_rec);
    int shiftend = gen < 2 ? 2 : NUMSHIFTS;
    int okind = !(isnan(factor) || factor < CNF2O_MINFACTOR);
    for (int g = 0; okind && g < NUMTYPES; g++)
        for (int s = 0; s < shiftend; s++) {
            if (s & T.shiftignore) continue;
            if (factor - factors[s] > 40) continue;
            for (int flag2 = 0; flag2 < NUMPATHS; flag2++) {
                if (cnf2o_ignoreflag2(P, &T, flag2, g, s, marker)) continue;
                double val = cnf2o_query(P, ind, s, first, last, marker, g, flag2, W, -200 + factor) - factor;
                if (!(isfinite(val) && val > -200)) continue;
                val = exp(val);
                /* updatehaplo, cpp:1561-1575 */
                double ok = cnf2o_emission(P, ind, marker, g, flag2, s);
                if (ok) {
                    tp_sink sink = {val, out, NULL};
                    tp_core(P, ind, UNKNOWN, 0, marker, (unsigned)(g * 2), flag2, s,
                            1u << (CNF2O_NUMGEN - 1), 0, NULL, UPD_HAPLOS, &sink);
                }
            }
        }
    cnf2o_fwbw_free(W);
}

/* What HOT LOOP 2 leaves in the thread-private `infprobs` (before moveinfprobs, cpp:3577-3597) and adds to
 * the analysed individual's `homozyg[marker]` for one marker (cpp:5513-5577, DOINFPROBS):
 *   sidevals[side][i-1] = trackpossible<GENOSPROBE>(allele i on the root's side `side`)      cpp:5519-5528
 *   homozyg[i-1]        = trackpossible<HOMOZYGOUS>(allele i)                                cpp:5531-5537
 *   trackpossible<GENOS>(allele i, side) with updateval = val * sidevals / sidevalsums[side]  cpp:5560-5568
 *   hz_out[i-1]        += val * homozyg[i-1] / sidevalsums[0]                                 cpp:5571-5575
 * inf_out[n_rec][2][2]: [individual][allele index][markerval - 1].  Divisions by a zero sidevalsum give
 * the reference's own NaN/inf. */
void cnf2o_infprobs_row(const cnf2o_ped *P, int ind, int gen, int first, int last, int marker,
                        double *inf_out, double *hz_out)
{
    cnf2o_fwbw *W = cnf2o_fwbw_new(P->n_markers);
    double factors[NUMSHIFTS], factor;
    cnf2o_sweep_ind(P, ind, gen, first, last, factors, &factor, NULL, 0, W);
    cnf2o_tree T;
    cnf2o_fixtrees(P, ind, &T);
    memset(inf_out, 0, sizeof(double) * 4 * P->n_rec);
    hz_out[0] = hz_out[1] = 0;
    const unsigned gw = 1u << (CNF2O_NUMGEN - 1);
    int shiftend = gen < 2 ? 2 : NUMSHIFTS;
    int okind = !(isnan(factor) || factor < CNF2O_MINFACTOR);
    for (int g = 0; okind && g < NUMTYPES; g++)
        for (int s = 0; s < shiftend; s++) {
            if (s & T.shiftignore) continue;
            if (factor - factors[s] > 40) continue;
            for (int flag2 = 0; flag2 < NUMPATHS; flag2++) {
                if (cnf2o_ignoreflag2(P, &T, flag2, g, s, marker)) continue;
                double val = cnf2o_query(P, ind, s, first, last, marker, g, flag2, W, -200 + factor) - factor;
                if (!(isfinite(val) && val > -200)) continue;
                val = exp(val);
                double sidevals[2][2] = {{0, 0}, {0, 0}}, sums[2] = {0, 0}, homozyg[2] = {0, 0};
                for (int side = 0; side < 2; side++)
                    for (int i = 1; i <= 2; i++) {
                        double sv = tp_core(P, ind, i, 0, marker, (unsigned)(g * 2 + side), flag2 ^ side, s, gw,
                                            0, NULL, UPD_GENOSPROBE, NULL);
                        sidevals[side][i - 1] += sv;
                        sums[side] += sv;
                    }
                for (int i = 1; i <= 2; i++)
                    homozyg[i - 1] += tp_core(P, ind, i, 0, marker, (unsigned)(g * 2), flag2, s, gw, 0, NULL,
                                              UPD_HOMOZYGOUS, NULL);
                for (int side = 0; side < 2; side++)
                    for (int i = 1; i <= 2; i++) {
                        tp_sink sink = {val * sidevals[side][i - 1] / sums[side], NULL, inf_out};
                        tp_core(P, ind, i, 0, marker, (unsigned)(g * 2 + side), flag2 ^ side, s, gw, 0, NULL,
                                UPD_GENOS, &sink);
                    }
                for (int i = 1; i <= 2; i++) hz_out[i - 1] += val * homozyg[i - 1] / sums[0];
            }
        }
    cnf2o_fwbw_free(W);
}

/* individ::descendants as postmarkerdata leaves it (cpp:3224-3255): every individual sends max(1, own count) to
 * both parents until nothing changes; zeros become 1.  (Restated without a pin: that loop is outside the
 * reference extract.  cnf2o_accumulate takes the counts as an input.) */
void cnf2o_descendants(const cnf2o_ped *P, int32_t *desc)
{
    int32_t *upsent = (int32_t *)calloc((size_t)P->n_rec, sizeof(int32_t));
    memset(desc, 0, sizeof(int32_t) * (size_t)P->n_rec);
    int any;
    do {
        any = 0;
        for (int r = 0; r < P->n_rec; r++) {
            int now = desc[r] ? desc[r] : 1;
            now -= upsent[r];
            if (now > 0) {
                for (int k = 0; k < 2; k++)
                    if (P->par[r * 2 + k] >= 0) desc[P->par[r * 2 + k]] += now;
                upsent[r] += now;
                any = 1;
            }
        }
    } while (any);
    for (int r = 0; r < P->n_rec; r++)
        if (!desc[r]) desc[r] = 1;
    free(upsent);
}

/* HOT LOOP 2 with its reductions (cpp:5416-5577, 5876-5902) for a list of individuals, in list order:
 * per marker the thread-private accumulators (cnf2o_haplos_row / cnf2o_infprobs_row), then
 *   sum = 1 / sum of infprobs[self][allele index 0][.];  homozyg[marker][.] *= sum           cpp:5880-5890
 *   moveinfprobs (cpp:3577-3597): target.infprobs[marker][side][val] += value * sum * 2 / 2^(slots the member
 *       occupies in reltreeordered) * descendants
 *   movehaplos (cpp:3599-3616): if the pair is non-zero and |haploweight - 0.5| < 0.5 - 1e-12:
 *       haplobase += b1 / (b1 + b2) * descendants, haplocount += descendants, b = haplos + e^-400 maxdiff^2 / 2
 * for every member of reltree (all existing window members; reltreeordered holds the non-empty ones only).
 * inf_out[n_rec][nm][2][2], hb_out / hc_out[n_rec][nm], hz_out[n_ind][nm][2]; nm = last - first + 1. */
void cnf2o_accumulate(const cnf2o_ped *P, const int *inds, const int *gens, int n_ind, int first, int last,
                      const int32_t *desc, double *inf_out, double *hb_out, double *hc_out, double *hz_out)
{
    const int   nm      = last - first + 1;
    const float maxdiff = 0.000005f;                                           /* cpp:228 */
    memset(inf_out, 0, sizeof(double) * 4 * (size_t)P->n_rec * nm);
    memset(hb_out, 0, sizeof(double) * (size_t)P->n_rec * nm);
    memset(hc_out, 0, sizeof(double) * (size_t)P->n_rec * nm);
    memset(hz_out, 0, sizeof(double) * 2 * (size_t)n_ind * nm);
    double *inf = (double *)malloc(sizeof(double) * 4 * (size_t)P->n_rec);
    double *hap = (double *)malloc(sizeof(double) * 2 * (size_t)P->n_rec);
    for (int j = 0; j < n_ind; j++) {
        const int ind = inds[j], gen = gens[j];
        double factors[NUMSHIFTS], factor;
        if (!cnf2o_sweep_ind(P, ind, gen, first, last, factors, &factor, NULL, 0, NULL)) continue;
        cnf2o_tree T;
        cnf2o_fixtrees(P, ind, &T);
        const double descf = desc[ind];
        for (int m = first; m <= last; m++) {
            double hz[2];
            cnf2o_infprobs_row(P, ind, gen, first, last, m, inf, hz);
            cnf2o_haplos_row(P, ind, gen, first, last, m, hap);
            double sum = 0;
            for (int v = 0; v < 2; v++) sum += inf[(ind * 2 + 0) * 2 + v];
            sum = 1 / sum;
            for (int v = 0; v < 2; v++) hz_out[((size_t)j * nm + (m - first)) * 2 + v] = hz[v] * sum;
            /* reltree: every existing window member, empty or not, once (cpp:3109,3136,3152,3183-3184) */
            int members[7], n_members = 0;
            members[n_members++] = ind;
            for (int l1 = 0; l1 < 2; l1++) {
                const int p1 = P->par[ind * 2 + l1];
                if (p1 < 0) continue;
                members[n_members++] = p1;
                for (int l2 = 0; l2 < 2; l2++)
                    if (P->par[p1 * 2 + l2] >= 0) members[n_members++] = P->par[p1 * 2 + l2];
            }
            for (int k = 0; k < n_members; k++) {
                const int r = members[k];
                int dup = 0;
                for (int k2 = 0; k2 < k; k2++) dup |= (members[k2] == r);
                if (dup) continue;
                double norm = sum;
                norm *= 2;
                for (int s7 = 0; s7 < 7; s7++)
                    if (T.ordered[s7] == r) norm /= 2;
                norm *= descf;
                const size_t o = (size_t)r * nm + (m - first);
                for (int side = 0; side < 2; side++)
                    for (int v = 0; v < 2; v++)
                        inf_out[(o * 2 + side) * 2 + v] += inf[(r * 2 + side) * 2 + v] * norm;
                if (hap[r * 2] || hap[r * 2 + 1]) {
                    const double hw = P->hw[(size_t)r * P->n_markers + m];
                    if (fabs(hw - 0.5) < 0.5 - 1e-12) {
                        double b1 = hap[r * 2] + exp(-400) * maxdiff * maxdiff * 0.5;
                        double b2 = hap[r * 2 + 1] + exp(-400) * maxdiff * maxdiff * 0.5;
                        hb_out[o] += b1 / (b1 + b2) * descf;
                        hc_out[o] += descf;
                    }
                }
            }
        }
    }
    free(inf);
    free(hap);
}

/* individ::addvariance (cpp:1489-1558), the emission-driven statistic postmarkerdata computes for every
 * individual and marker (cpp:3373-3389) to pick the marker whose phase gets locked: trackpossible with
 * zeropropagate = NO_EQUIVALENCE (-1: alleles are matched as usual, every level weighs 0.5 instead of the
 * phase weight, below the root only the traced line is followed), fed the individual's OWN two alleles
 * (with their sure as error odds) in turn, over shift modes 0-1, all 128 flags i = 2 g + firstpar and all
 * admissible paths; per (shift, i & 1, flag2 & 1) the signed sum over the two alleles is squared.
 * Returns 0 and leaves *out alone when every term is zero (cpp:1550). */
int cnf2o_addvariance(const cnf2o_ped *P, int rec, int marker, int flag2ignore, double *out)
{
    const int32_t *themarker = rec_allele(P, rec, marker);
    const double  *thesure   = rec_sure(P, rec, marker);
    double sum = 0, sqsum = 0;
    for (int shift = 0; shift < 2; shift++)
        for (int majori = 0; majori < 2; majori++)
            for (int majorflag2 = 0; majorflag2 < 2; majorflag2++) {
                double fullok = 0, ok = 0;
                for (int i = majori; i < NUMTYPES * 2; i += 2)
                    for (int flag2 = majorflag2; flag2 < NUMPATHS; flag2 += 2) {
                        if (flag2 & flag2ignore) continue;
                        for (int allele = 0; allele < 2; allele++) {
                            double term = tp_core(P, rec, themarker[allele], thesure[allele], marker, (unsigned)i,
                                                  flag2, shift, 1u << (CNF2O_NUMGEN - 1), -1, NULL, 0, NULL);
                            ok += term * (allele ? 1 : -1);
                            fullok += term;
                        }
                    }
                ok = fabs(ok);
                sum += fullok;
                sqsum += ok * ok;
            }
    if (!sum) return 0;
    *out = sqsum;
    return 1;
}

int cnf2o_sweep_batch(const cnf2o_ped *P, const int *inds, const int *gens, int n_ind,
                      int first, int last, double *factors_out, double *factor_out,
                      double *dosage_out, int mode, int n_threads)
{
    int used = 1;
    int nm = last - first + 1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
        cnf2o_fwbw *W = cnf2o_fwbw_new(P->n_markers);               /* threadprivate store, cpp:407 */
#pragma omp for schedule(dynamic, 1)                                /* cpp:5294 */
        for (int j = 0; j < n_ind; j++) {
#else
    {
        cnf2o_fwbw *W = cnf2o_fwbw_new(P->n_markers);
        for (int j = 0; j < n_ind; j++) {
#endif
            double *d = dosage_out ? dosage_out + (size_t)j * nm * 3 : NULL;
            cnf2o_sweep_ind(P, inds[j], gens ? gens[j] : 2, first, last,
                            factors_out ? factors_out + (size_t)j * NUMSHIFTS : NULL,
                            factor_out ? factor_out + j : NULL, d, mode, W);
            if (d) {
                for (int m = 0; m < nm; m++) {
                    double sum = d[m * 3] + d[m * 3 + 1] + d[m * 3 + 2];
                    if (sum > 0)
                        for (int k = 0; k < 3; k++) d[m * 3 + k] /= sum;
                }
            }
        }
        cnf2o_fwbw_free(W);
    }
    return used;
}
