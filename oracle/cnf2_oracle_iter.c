/*
 * cnf2_oracle_iter.c -- TEST INFRASTRUCTURE ONLY (see cnf2_oracle.h).
 *
 * Plain-C restatement, statement by statement, of the per-iteration parameter updates of cnF2freq
 * (SURVEY.md section 8(f)-4): caplogitchange / cappedgd (cpp:4004-4177), processinfprobs (cpp:4179-4323),
 * relskewhmm (cpp:4325-4466), updatehaploweights (cpp:4533-4734) and the step-size control of doit
 * (cpp:6373-6392).  cpp: = /root/reference/cnF2freq.cpp.  No code is shared with the HIP product, whose own
 * form of the same arithmetic is cnf2freq_amd/csrc/cnf2_update.h.
 *
 * PARITY UNPINNED for this file: these ranges call boost::math::quadrature::gauss<double, 15> (cpp:4150) and
 * Boost is absent from the image, so they cannot be compiled into oracle/_ref (a stand-in for the Boost header
 * would not be the reference).  Dependency: Boost.Math (the reference names boost_1_61_0 only in demo.sh:6;
 * quadrature/gauss.hpp appeared in Boost 1.66).  Its published algorithm is restated in gauss15() below: for an
 * odd point count the centre node first, then the node pairs in ascending abscissa, (f(+x) + f(-x)) * w, on
 * [a, b] through avg + scale * z, result scaled by (b - a) / 2.  The reference holds no fixtures for these
 * functions; they are checked for self-consistency and against the product's independent form.
 */
#include "cnf2_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cpp:228: `const float maxdiff`.  The reference's expressions keep float arithmetic wherever maxdiff meets an int:
 * `maxdiff / (ind->children + 1)` (cpp:4240, 4703) is a FLOAT division and `1 - maxdiff` (cpp:4667) a FLOAT
 * subtraction (0.99999499320983887, not 1 - 5e-6), both promoted to double afterwards.  Found by the bit-level pin
 * against the reference's own text (goldens G14). */
static const float maxdiff = 0.000005f;
static double epsilon_of(int children) { return (double)(maxdiff / (float)(children + 1)); }
static double similarity_cap(void) { return (double)(1.0f - maxdiff); }

/* 15-point Gauss-Legendre rule, non-negative abscissas in ascending order with their weights
 * (Abramowitz & Stegun table 25.4; the values boost/math/quadrature/gauss.hpp tabulates for N = 15) */
static const double gl_x[8] = {0.0000000000000000000000000000000000,
                               0.2011940939974345223006283033945962,
                               0.3941513470775633698972073709810455,
                               0.5709721726085388475372267372539106,
                               0.7244177313601700474161860546139380,
                               0.8482065834104272162006483207742169,
                               0.9372733924007059043077589477102095,
                               0.9879925180204854284895657185866126};
static const double gl_w[8] = {0.2025782419255612728806201999675193,
                               0.1984314853271115764561183264438393,
                               0.1861610000155622110268005618664228,
                               0.1662692058169939335532008604812088,
                               0.1395706779261543144478047945110283,
                               0.1071592204671719350118695466858693,
                               0.0703660474881081247092674164506674,
                               0.0307532419961172683546283935772044};

typedef double (*grad_fn)(double x, const void *ctx);

typedef struct {
    grad_fn     gradient;
    const void *ctx;
    double      epsilon;
} actual_gradient;

/* cpp:4107-4113: actualgradient */
static double actualgradient(const actual_gradient *a, double val)
{
    if (val < a->epsilon) val = a->epsilon;              /* std::clamp(val, epsilon, 1 - epsilon) */
    else if (1 - a->epsilon < val) val = 1 - a->epsilon;
    double toret = a->gradient(val, a->ctx);
    return 1. / (toret + 0 /* randomdrift */);
}

/* boost::math::quadrature::gauss<double, 15>::integrate(f, a, b) */
static double gauss15(const actual_gradient *f, double a, double b)
{
    double avg = (a + b) * 0.5;
    double scale = (b - a) * 0.5;
    double result = actualgradient(f, avg + scale * 0.0) * gl_w[0];
    for (int i = 1; i < 8; i++) {
        double fp = actualgradient(f, avg + scale * gl_x[i]);
        double fm = actualgradient(f, avg + scale * -gl_x[i]);
        result += (fp + fm) * gl_w[i];
    }
    return scale * result;
}

/* cpp:4006-4038 */
double cnf2o_caplogitchange(double intended, double orig, double epsilon, int *hitnnn, int breakathalf)
{
    double nnn = 3;
    if (nnn < 1.0) nnn = 1.0;

    double limn = (nnn - 1.0) * orig * (-1 + orig);

    double limd1 = -1 - (nnn - 1.0) * orig;
    double limd2 = (nnn - 1.0) * orig - nnn;

    intended = (1.0 - epsilon < intended) ? 1.0 - epsilon : intended; /* min */
    intended = (intended < epsilon) ? epsilon : intended;             /* max */
    double diff = intended - orig;

    if (diff > limn / limd1) {
        intended = orig + limn / limd1;
        if (intended < 0.5) (*hitnnn)++;
    }

    if (diff < -limn / limd2) {
        intended = orig - limn / limd2;
        if (intended > 0.5) (*hitnnn)++;
    }

    if (breakathalf && (intended - 0.5) * (orig - 0.5) < 0) intended = 0.5 * (0.5 + orig);

    return intended;
}

/* cpp:4040-4177, the #else branch (binary search on the integral of 1 / gradient) */
static double cappedgd(grad_fn gradient, const void *ctx, double orig, double epsilon, double scalefactor,
                       int *hitnnn, int breakathalf)
{
    int             dumpval = 0;
    actual_gradient ag = {gradient, ctx, epsilon};
    double lolim = cnf2o_caplogitchange(epsilon, orig, epsilon, &dumpval, breakathalf);
    double lo = lolim - epsilon * 0.125;
    double hilim = cnf2o_caplogitchange(1 - epsilon, orig, epsilon, &dumpval, breakathalf);
    double hi = hilim + epsilon * 0.125;

    orig = cnf2o_caplogitchange(orig, orig, epsilon, &dumpval, breakathalf);

    double gradval = actualgradient(&ag, orig);
    if (!isfinite(gradval) || !scalefactor) {
        lo = orig;
        hi = orig;
    }
    int lowside = gradval < 0;
    if (lowside) hi = orig;
    else lo = orig;
    for (int i = 0; i < 51 && scalefactor; i++) {
        if (lo > hilim || hi < lolim) break;

        double mid = (lo + hi) / 2;
        double prel = 0;
        double gv = actualgradient(&ag, mid);
        if (((gv < 0) ^ lowside) || !isfinite(gv)) {
            prel = (scalefactor + 0.1) * 1.1;
        } else {
            double start = orig;
            double end = mid;
            if (start > end) {
                double t = start;
                start = end;
                end = t;
            }
            if (end - start < 1e-10) break;
            prel = gauss15(&ag, start, end);
            if (end != mid) prel = -prel;
            if (!isfinite(prel)) prel = (scalefactor + 0.1) * 1.1;
        }
        if (fabs(prel - scalefactor) < scalefactor * 1e-3) break;

        if ((prel < scalefactor) ^ lowside) lo = mid;
        else hi = mid;
    }

    if (!scalefactor) {
        lo = orig;
        hi = orig;
    }

    return cnf2o_caplogitchange((lo + hi) / 2, orig, epsilon, hitnnn, breakathalf);
}

static double square(double v) { return v * v; }

/* ---------------------------------------------------------------------------- processinfprobs */
typedef struct {
    double curprob, hzygcorred, sum, etf, priord;
} pip_ctx;

/* the gradient lambda of cpp:4270-4287 */
static double pip_gradient(double x, const void *vc)
{
    const pip_ctx *c = (const pip_ctx *)vc;
    const double curprob = c->curprob, hzygcorred = c->hzygcorred, sum = c->sum;
    double d = -(-square(curprob*hzygcorred)*log(x) + square(curprob*hzygcorred)*log(1 - x) + square(curprob)*hzygcorred*sum*log(x) - square(curprob)*hzygcorred*sum*log(1 - x) - square(curprob)*hzygcorred*sum - square(curprob*sum)*x + square(curprob*sum) + curprob*square(hzygcorred)*log(x) - curprob*square(hzygcorred)*log(1 - x) + curprob*square(hzygcorred) + 2*curprob*hzygcorred*sum*x - curprob*hzygcorred*sum*log(x) + curprob*hzygcorred*sum*log(1 - x) - curprob*hzygcorred*sum - square(hzygcorred)*x)/square(curprob*hzygcorred + curprob*sum*x - curprob*sum - hzygcorred*x);
    double et = log(1 / x - 1);
    d += c->etf * et;
    d += c->etf * c->priord;
    return d;
}

/* cpp:4179-4323 for one (individual, marker j, side).  The accumulator map infprobs[j][side] is given as
 * inf[2] (keys MarkerVal 1 and 2; present[v] says whether the key exists); markerdata / markersure / prior
 * values of this side are scalars.  Returns 1 and sets *out_allele, *out_sure when cpp:4303-4313 assigns. */
int cnf2o_processinfprobs(const double inf_in[2], const int present[2], int side, int curmarker, double cursure,
                          int has_prior, int priorval, double priorsure, int empty, int children,
                          double scalefactor, double entropyfactor, int *hitnnn, double inf_out[2],
                          int *out_allele, double *out_sure)
{
    double bestprob = 0;
    int    bestmarker = 0;
    double sum = 0;
    double inf[2] = {inf_in[0], inf_in[1]};

    for (int v = 0; v < 2; v++)
        if (present[v]) sum += inf[v];
    if (!has_prior) priorval = 0;                                          /* cpp:4195-4199 */

    double ef = exp(0 * -0.01 * 1) * entropyfactor;                         /* cpp:4220 */

    for (int v = 0; v < 2; v++) {                                          /* cpp:4222 */
        if (!present[v]) continue;
        const int first = v + 1;
        const double second = inf[v];
        double curprob = 0.5;
        if (curmarker != 0) curprob = fabs((curmarker == first ? 1 : 0) - cursure);   /* cpp:4228-4231 */
        double hzygcorred = second;
        double etf = 1 * ef;
        double epsilon = epsilon_of(children);
        double priord = 0;
        double priorprob = 0.5;
        if (priorval != 0) {                                               /* cpp:4245-4268 */
            priorprob = 1.0 - priorsure;
            if (first != priorval) priorprob = 1.0 - priorprob;
            if (priorprob == 0) priord -= 10000;
            else if (priorprob == 1) priord += 10000;
            else {
                priorprob = priorprob < 1e-14 ? 1e-14 : (1 - 1e-14 < priorprob ? 1 - 1e-14 : priorprob);
                priord += log(priorprob) - log(1 - priorprob);
            }
        }
        pip_ctx c = {curprob, hzygcorred, sum, etf, priord};
        double intended = cappedgd(pip_gradient, &c, curprob, epsilon, scalefactor, hitnnn, 0);
        inf[v] = intended;                                                 /* cpp:4289 */
    }

    for (int v = 0; v < 2; v++) {                                          /* cpp:4292-4300 */
        if (!present[v]) continue;
        if (inf[v] > bestprob - (side ? 1e-30 : 0)) {
            bestmarker = v + 1;
            bestprob = inf[v];
        }
    }
    inf_out[0] = inf[0];
    inf_out[1] = inf[1];
    if (!empty && (bestmarker != 0 || bestprob > 0)) {                     /* cpp:4303-4313 */
        if (has_prior) {
            *out_allele = bestmarker;
            *out_sure = 1.0 - bestprob;
            return 1;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------- relskewhmm */
/* cpp:4338-4441 (realhmm == true): ratio[endmarker - firstmarker] from haploweight and relhaplo */
void cnf2o_relskew_ratio(const double *haploweight, const double *relhaplo, int firstmarker, int endmarker,
                         double *ratio)
{
    const int n = endmarker - firstmarker;
    double *fw = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double s[2] = {0.5, 0.5};
    for (int m = firstmarker; m < endmarker; m++) {                        /* FW, cpp:4408-4415 */
        double w = haploweight[m];                                         /* doemissions */
        for (int k = 0; k < 2; k++) s[k] *= fabs(!k - w);
        fw[(m - firstmarker) * 2 + 0] = s[0];
        fw[(m - firstmarker) * 2 + 1] = s[1];
        if (s[0] + s[1] < 1e-10) {                                         /* renormalizes */
            s[0] *= 1e20;
            s[1] *= 1e20;
        }
        double nn = relhaplo[m], nb = 1 - nn;                              /* dotransitions */
        double nexts[2];
        for (int k = 0; k < 2; k++) nexts[k] = s[k] * nn + s[!k] * nb;
        s[0] = nexts[0];
        s[1] = nexts[1];
    }
    s[0] = s[1] = 0.5;                                                     /* cpp:4418-4440 */
    int m = n - 1;
    ratio[m] = fw[m * 2 + 1] / (fw[m * 2 + 0] + fw[m * 2 + 1]);
    for (m = endmarker - 2; m >= firstmarker; m--) {
        double w = haploweight[m + 1];
        for (int k = 0; k < 2; k++) s[k] *= fabs(!k - w);
        double nn = relhaplo[m], nb = 1 - nn;
        double nexts[2];
        for (int k = 0; k < 2; k++) nexts[k] = s[k] * nn + s[!k] * nb;
        s[0] = nexts[0];
        s[1] = nexts[1];
        if (s[0] + s[1] < 1e-10) {
            s[0] *= 1e20;
            s[1] *= 1e20;
        }
        double ratiofactors[2] = {0, 0};
        for (int k = 0; k < 2; k++) ratiofactors[k] += s[k] * fw[(m - firstmarker) * 2 + k];
        ratio[m - firstmarker] = ratiofactors[1] / (ratiofactors[0] + ratiofactors[1]);
    }
    free(fw);
}

/* ---------------------------------------------------------------------------- updatehaploweights */
typedef struct {
    double haploweight, haplobase, haplocount, similarity, ef, relskewterm, descendants;
} uhw_ctx;

/* the gradient lambda of cpp:4684-4698 */
static double uhw_gradient(double in, const void *vc)
{
    const uhw_ctx *c = (const uhw_ctx *)vc;
    const double hw = c->haploweight, hb = c->haplobase, hc = c->haplocount;
    double x = in;
    double out = -(-square(hw*hb)*log(x) + square(hw*hb)*log(1 - x) + square(hw)*hb*hc*log(x) - square(hw)*hb*hc*log(1 - x) - square(hw)*hb*hc - square(hw*hc)*x + square(hw*hc) + hw*square(hb)*log(x) - hw*square(hb)*log(1 - x) + hw*square(hb) + 2*hw*hb*hc*x - hw*hb*hc*log(x) + hw*hb*hc*log(1 - x) - hw*hb*hc - square(hb)*x)/square(hw*hb + hw*hc*x - hw*hc - hb*x);
    out += ((1 - c->similarity) * 1 * (c->ef * log(1 / in - 1)) +
            (c->relskewterm - in) / (in - in * in) * c->descendants);
    return out;
}

/* cpp:4533-4734 for ONE individual: every marker of every chromosome that has any haplocount.  Arrays are
 * indexed by marker; haploweight, haplobase and haplocount are updated in place exactly as the reference leaves
 * them.  chromstarts[n_chrom + 1].  lastinved[c] == -1 on this path (cpp:5310). */
void cnf2o_updatehaploweights(int n_chrom, const int *chromstarts, double *haploweight, double *haplobase,
                              double *haplocount, const int32_t *allele, const double *sure, const double *relhaplo,
                              int children, int descendants, double scalefactor, double entropyfactor,
                              int *hitnnn)
{
    for (int cno = 0; cno < n_chrom; cno++) {
        const int c0 = chromstarts[cno], c1 = chromstarts[cno + 1];
        int anyinfo = 0;
        for (int k = c0; k < c1 && !anyinfo; k++)
            if (haplocount[k]) anyinfo = 1;                                /* cpp:4559-4563 */
        if (!anyinfo) continue;
        double *ratio = (double *)malloc(sizeof(double) * (size_t)(c1 - c0));
        cnf2o_relskew_ratio(haploweight, relhaplo, c0, c1, ratio);         /* cpp:4564 */
        for (int j = c0; j < c1; j++) {
            if (!(haploweight[j] && haploweight[j] != 1)) continue;        /* cpp:4591 */
            double relskewterm = ratio[j - c0];                            /* cpp:4606-4610 */
            double scorea = 1.0 - sure[j * 2];
            double scoreb = 1.0 - sure[j * 2 + 1];
            if (allele[j * 2] != allele[j * 2 + 1]) scoreb = 1 - scoreb;   /* cpp:4644-4646 */
            double similarity = (scorea * scoreb + (1 - scorea) * (1 - scoreb));   /* cpp:4658 */
            if (!haplocount[j] || similarity == 1.0) {                     /* cpp:4660-4664 */
                haplocount[j] = haplocount[j] < 1.0 ? 1.0 : haplocount[j];
                haplobase[j] = haploweight[j] * haplocount[j];
            } else {                                                       /* cpp:4665-4677 */
                if (similarity >= similarity_cap()) similarity = similarity_cap();
                double count = haplocount[j];
                haplobase[j] -= count * haploweight[j];
                count = count - similarity * count;
                haplobase[j] += count * haploweight[j];
                haplobase[j] *= haplocount[j] / count;
                if (haplobase[j] < 0) haplobase[j] = 0;
                if (haplobase[j] >= haplocount[j]) haplobase[j] = haplocount[j];
            }
            double ef = exp(0 * -0.01 * 1) * entropyfactor;                /* cpp:4679 */
            uhw_ctx c = {haploweight[j], haplobase[j], haplocount[j], similarity, ef, relskewterm, (double)descendants};
            double intended = cappedgd(uhw_gradient, &c, haploweight[j], epsilon_of(children), scalefactor,
                                       hitnnn, 0 /* lastinved[cno] != -1 */);   /* cpp:4703 */
            haploweight[j] = intended;                                     /* cpp:4714 */
        }
        free(ratio);
    }
}

/* cpp:6373-6392 with any == false: returns the new scalefactor, updates old[0] = oldhitnnn, old[1] = oldhitnnn2 */
double cnf2o_scalefactor_step(double scalefactor, int hitnnn, int *old, int n_dous)
{
    int mx = old[0] > old[1] ? old[0] : old[1];
    int mn = old[0] < old[1] ? old[0] : old[1];
    int badhit = hitnnn > mx;
    if (badhit) scalefactor /= 1.1;
    int fl = n_dous / CNF2O_TURNBITS;
    int goodhit = hitnnn < (mn > fl ? mn : fl) * 0.99;
    if (goodhit) scalefactor *= 1.21;
    scalefactor *= 0.997;
    old[1] = old[0];
    old[0] = hitnnn;
    return scalefactor;
}

/* exposed for the quadrature self-check in tests: integral of 1 / (slope x + icpt) through the rule above */
static double lin_grad(double x, const void *vc)
{
    const double *c = (const double *)vc;
    return c[0] * x + c[1];
}
double cnf2o_gauss15_reciprocal_linear(double slope, double icpt, double a, double b)
{
    double c[2] = {slope, icpt};
    actual_gradient ag = {lin_grad, c, 0.0};
    return gauss15(&ag, a, b);
}
