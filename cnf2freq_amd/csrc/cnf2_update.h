// cnf2_update.h -- the per-iteration parameter updates that close a haplotyping iteration of cnF2freq
// (SURVEY.md section 8(f)-4): what processinfprobs (cnF2freq.cpp:4179-4323), updatehaploweights
// (cnF2freq.cpp:4533-4734), relskewhmm (cnF2freq.cpp:4325-4466), cappedgd and caplogitchange
// (cnF2freq.cpp:4004-4177) do to one (individual, marker), restated as pure functions over plain numbers.
// Host + device: the library runs them as kernels over the device-resident accumulators (cnf2_kernels.hip),
// the host build exists so that the arithmetic is unit-tested without a GPU against the oracle's literal
// restatement (tests/shim, tests/test_host_update.py).
//
// Every parameter p in (0, 1) (a genotype certainty or a haplotype weight) is moved along the flow
// dp/dt = G(p) of a gradient G for a fixed "time" (scalefactor): the new value q solves
//     integral_p^q dx / G(x) = scalefactor,
// found by bisection with a 15-point Gauss-Legendre rule for the integral, inside a cap on how far a value may
// move in one iteration.  The reference takes the rule from boost::math::quadrature::gauss<double, 15>
// (cnF2freq.cpp:4150); Boost is not part of this build, the nodes and weights below are the published ones of the
// 15-point rule and the summation order is Boost's (centre node first, then node pairs outwards).
#ifndef CNF2_UPDATE_H
#define CNF2_UPDATE_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CNF2_UHD __host__ __device__ inline
#else
#define CNF2_UHD inline
#endif

namespace cnf2 {

// maxdiff is a float in the reference (cnF2freq.cpp:228) and its expressions stay in float arithmetic where it meets an
// int: the clamp distance maxdiff / (children + 1) (cnF2freq.cpp:4240, 4703) is a float quotient and the similarity cap
// 1 - maxdiff (cnF2freq.cpp:4667) a float difference (0.99999499..., i.e. 1 - similarity >= 5.0068e-6, not 5e-6); both are
// promoted to double afterwards.  (Found by pinning against the reference's own text, goldens G14.)
#define CNF2_MAXDIFF_F 0.000005f
CNF2_UHD double clamp_distance(int children) { return (double)(CNF2_MAXDIFF_F / (float)(children + 1)); }
#define CNF2_SIMILARITY_CAP ((double)(1.0f - CNF2_MAXDIFF_F))

struct StepControl {
    double scalefactor;     // cnF2freq.cpp:3573 (0.013 at start, adapted after every update pass, 6373-6392)
    double entropyfactor;   // cnF2freq.cpp:3574 (1)
};

// a / b inside the gradients: on the device a reciprocal with two Newton steps and a residual correction (the operands
// are probabilities, evidence sums and their products: no denormals to scale for); 0 and non-finite divisors give a
// non-finite quotient, which is all the callers test for.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double upd_div(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
#else
inline double upd_div(double a, double b) { return a / b; }
#endif

// ------------------------------------------------------------------ 15-point Gauss-Legendre
CNF2_UHD double gl15_node(int i)
{
    const double x[8] = {0.0,
                         0.20119409399743452230062830339460,
                         0.39415134707756336989720737098105,
                         0.57097217260853884753722673725391,
                         0.72441773136017004741618605461394,
                         0.84820658341042721620064832077422,
                         0.93727339240070590430775894771021,
                         0.98799251802048542848956571858661};
    return x[i];
}
CNF2_UHD double gl15_weight(int i)
{
    const double w[8] = {0.20257824192556127288062019996752,
                         0.19843148532711157645611832644384,
                         0.18616100001556221102680056186642,
                         0.16626920581699393355320086048121,
                         0.13957067792615431444780479451103,
                         0.10715922046717193501186954668587,
                         0.07036604748810812470926741645067,
                         0.03075324199611726835462839357720};
    return w[i];
}

template <class F>
CNF2_UHD double gauss15(F&& f, double a, double b)
{
    const double mid = (a + b) * 0.5, half = (b - a) * 0.5;
    double acc = f(mid) * gl15_weight(0);
    for (int i = 1; i < 8; i++) {
        const double fp = f(mid + half * gl15_node(i));
        const double fm = f(mid + half * -gl15_node(i));
        acc += (fp + fm) * gl15_weight(i);
    }
    return half * acc;
}

// ------------------------------------------------------------------ cap on one iteration's move
// caplogitchange (cnF2freq.cpp:4006-4038) with nnn = 3: a value may rise by at most 2p(1-p)/(1+2p) and fall by at
// most 2p(1-p)/(3-2p); a capped move that stays on the far side of 1/2 counts as a "hit" (the step-size control of
// cnF2freq.cpp:6373-6392 watches their number).  breakathalf: a move across 1/2 stops half way to it.
CNF2_UHD double cap_step(double intended, double orig, double epsilon, int* hits, bool breakathalf)
{
    const double room = (2.0 * orig) * (1.0 - orig);
    const double up = room / (1.0 + 2.0 * orig), down = room / (3.0 - 2.0 * orig);
    const double top = 1.0 - epsilon;
    intended = (top < intended) ? top : intended;
    intended = (intended < epsilon) ? epsilon : intended;
    const double diff = intended - orig;
    if (diff > up) {
        intended = orig + up;
        if (intended < 0.5) ++*hits;
    }
    if (diff < -down) {
        intended = orig - down;
        if (intended > 0.5) ++*hits;
    }
    if (breakathalf && (intended - 0.5) * (orig - 0.5) < 0) intended = 0.5 * (0.5 + orig);
    return intended;
}

// cappedgd (cnF2freq.cpp:4040-4177, the branch that is compiled in): rgradient(x) -> dt/dp = 1 / (dp/dt) at x (the
// callers form the reciprocal of their gradient as ONE quotient).  The bisection is kept as data (FlowState) with
// begin / advance / end, so that the device kernels can run one step of 64 independent flows per wavefront and hand a
// lane the next flow as soon as its own has ended (flows take between 1 and 51 steps); flow_step() is the plain loop.
struct FlowState {
    double orig, epsilon, lolim, hilim, lo, hi;
    int    it;            // steps taken so far
    bool   falling;       // the gradient at the starting point is negative
    bool   live;          // more steps to take
};
template <class G>
CNF2_UHD double flow_pace(G&& rgradient, double v, double epsilon)      // dt/dp at the clamped position
{
    const double top = 1.0 - epsilon;
    v = (v < epsilon) ? epsilon : ((top < v) ? top : v);
    return rgradient(v);
}
template <class G>
CNF2_UHD void flow_begin(FlowState* f, G&& rgradient, double orig, double epsilon, double scalefactor, bool breakathalf)
{
    const double top = 1.0 - epsilon;
    int          ignored = 0;
    // bisection bracket: slightly wider than the cap, so that the final cap_step is the one that counts the hit
    f->epsilon = epsilon;
    f->lolim = cap_step(epsilon, orig, epsilon, &ignored, breakathalf);
    f->hilim = cap_step(top, orig, epsilon, &ignored, breakathalf);
    f->lo = f->lolim - epsilon * 0.125;
    f->hi = f->hilim + epsilon * 0.125;
    f->orig = cap_step(orig, orig, epsilon, &ignored, breakathalf);
    const double g0 = flow_pace(rgradient, f->orig, epsilon);
    if (!isfinite(g0) || !scalefactor) f->lo = f->hi = f->orig;
    f->falling = g0 < 0;
    if (f->falling) f->hi = f->orig;
    else f->lo = f->orig;
    f->it = 0;
    f->live = scalefactor != 0;
}
// one bisection step; returns whether another one follows
template <class G>
CNF2_UHD bool flow_advance(FlowState* f, G&& rgradient, double scalefactor)
{
    if (!f->live) return false;
    if (f->it >= 51 || f->lo > f->hilim || f->hi < f->lolim) return f->live = false;   // done / outside the true bounds
    f->it++;
    const double mid = (f->lo + f->hi) / 2;
    const double gm = flow_pace(rgradient, mid, f->epsilon);
    double       t;                                          // time the flow needs from orig to mid
    if (((gm < 0) != f->falling) || !isfinite(gm)) {
        t = (scalefactor + 0.1) * 1.1;                        // the gradient turns round before mid: too far
    } else {
        double a = f->orig, b = mid;
        if (a > b) {
            const double s = a;
            a = b;
            b = s;
        }
        if (b - a < 1e-10) return f->live = false;
        const double eps = f->epsilon;
        t = gauss15([&](double v) { return flow_pace(rgradient, v, eps); }, a, b);
        if (b != mid) t = -t;
        if (!isfinite(t)) t = (scalefactor + 0.1) * 1.1;
    }
    if (fabs(t - scalefactor) < scalefactor * 1e-3) return f->live = false;
    if ((t < scalefactor) != f->falling) f->lo = mid;
    else f->hi = mid;
    return true;
}
CNF2_UHD double flow_end(const FlowState& f, double scalefactor, int* hits, bool breakathalf)
{
    double lo = f.lo, hi = f.hi;
    if (!scalefactor) lo = hi = f.orig;
    return cap_step((lo + hi) / 2, f.orig, f.epsilon, hits, breakathalf);
}
template <class G>
CNF2_UHD double flow_step(G&& rgradient, double orig, double epsilon, double scalefactor, int* hits, bool breakathalf)
{
    FlowState f;
    flow_begin(&f, rgradient, orig, epsilon, scalefactor, breakathalf);
    while (flow_advance(&f, rgradient, scalefactor)) {}
    return flow_end(f, scalefactor, hits, breakathalf);
}

// The data term shared by both updates.  With a current value y of the parameter, g the evidence for "1" gathered
// at y and h the total evidence, the reference differentiates
//     val(x) = (H (1-x) log(1-x) + G x log x) / (H (1-x) + G x),   G = g / y,  H = (h - g) / (1 - y)
// and writes the derivative out as one long polynomial in (y, g, h, log x, log(1-x))
// (cnF2freq.cpp:4275, 4684).  Multiplying G and H through by y (1 - y) gives the same derivative in the
// division-free form used here:  a = g (1-y), b = (h-g) y, Q = b (1-x) + a x,
//     val'(x) = (a b logit(x) + (a - b) Q) / Q^2,   logit(x) = log x - log(1-x).
// The entropy terms of both gradients are multiples of log(1/x - 1) = -logit(x), so one logarithm serves a whole
// gradient evaluation (the flow spends its time there: 16 evaluations per bisection step).
struct Evidence {
    double ab, amb, a, b;   // a b, a - b, a, b
};
CNF2_UHD Evidence evidence_terms(double y, double g, double h)
{
    Evidence e;
    e.a = g * (1.0 - y);
    e.b = (h - g) * y;
    e.ab = e.a * e.b;
    e.amb = e.a - e.b;
    return e;
}
// logit(x) = log(x / (1 - x)) for x in [epsilon, 1 - epsilon] (the clamped positions of flow_step: the quotient is a
// positive normal number between 1e-7 and 1e7, so the device version needs none of the library logarithm's special
// cases and double-double arithmetic -- 98 instructions there): split off the exponent, bring the mantissa to
// [sqrt(1/2), sqrt(2)), log m = 2 atanh((m - 1) / (m + 1)) as an odd series (|s| < 0.1716: eleven terms), e * ln 2 in two
// parts.  Good to a few units in the last place; the host build keeps the library call.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double logit(double x)
{
    const double r = upd_div(x, 1.0 - x);
    double       m = __builtin_amdgcn_frexp_mant(r);             // [1/2, 1)
    int          e = __builtin_amdgcn_frexp_exp(r);
    const bool   low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double s = upd_div(m - 1.0, m + 1.0), z = s * s;
    double       p = 1.0 / 21.0;
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    const double s2 = s + s;
    const double lm = fma(s2, z * p, s2);                        // 2 s (1 + z / 3 + z^2 / 5 + ...)
    const double ed = (double)e;
    return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, lm));
}
#else
inline double logit(double x) { return log(x / (1.0 - x)); }
#endif
CNF2_UHD double evidence_slope(const Evidence& e, double x, double lg)
{
    const double q = e.b * (1.0 - x) + e.a * x;
    return (e.ab * lg + e.amb * q) / (q * q);
}
CNF2_UHD double evidence_slope(double y, double g, double h, double x)
{
    return evidence_slope(evidence_terms(y, g, h), x, logit(x));
}

// ------------------------------------------------------------------ genotype certainties (processinfprobs)
// One side (allele index) of one individual at one marker.  inf[v-1] = accumulated evidence for allele value v
// (moveinfprobs, cnF2freq.cpp:3577-3597); an entry takes part if it is > 0 (the reference iterates a map that
// holds an entry for every value that was ever added to; values are sums of positive terms).
struct SideState {
    int    allele;        // markerdata[side]: 0 unknown, 1, 2, 9
    double sure;          // markersure[side]
    int    prior_allele;  // priormarkerdata[side] or 0 when the individual has no prior (not genotyped)
    double prior_sure;
};

// The flow of ONE value v + 1 of a side as data: gradient = data + entropy (log(1/x - 1) = -logit) + prior
// = (N + Q^2 E) / Q^2; its reciprocal is formed as one quotient.
struct CertaintyFlow {
    Evidence ev;
    double   ef, priord;
    double   curprob, epsilon;
};
// false when the value takes no part (no evidence for it)
CNF2_UHD bool certainty_flow_setup(const double inf[2], int v, const SideState& s, int children, const StepControl& sc,
                                   CertaintyFlow* c)
{
    if (!(inf[v] > 0)) return false;
    double sum = 0;
    for (int k = 0; k < 2; k++)
        if (inf[k] > 0) sum += inf[k];
    const int value = v + 1;
    double    curprob = 0.5;
    if (s.allele != 0) curprob = fabs((s.allele == value ? 1 : 0) - s.sure);
    double priord = 0;
    if (s.prior_allele != 0) {                                // cnF2freq.cpp:4245-4268
        double priorprob = 1.0 - s.prior_sure;
        if (value != s.prior_allele) priorprob = 1.0 - priorprob;
        if (priorprob == 0) priord -= 10000;
        else if (priorprob == 1) priord += 10000;
        else {
            priorprob = priorprob < 1e-14 ? 1e-14 : (priorprob > 1 - 1e-14 ? 1 - 1e-14 : priorprob);
            priord += log(priorprob) - log(1 - priorprob);
        }
    }
    c->ev = evidence_terms(curprob, inf[v], sum);
    c->ef = sc.entropyfactor;                                 // exp(0 * -0.01 * iter) * entropyfactor
    c->priord = priord;
    c->curprob = curprob;
    c->epsilon = clamp_distance(children);
    return true;
}
CNF2_UHD double certainty_rgradient(const CertaintyFlow& c, double x)
{
    const double lg = logit(x);
    const double q = c.ev.b * (1.0 - x) + c.ev.a * x, q2 = q * q;
    return upd_div(q2, (c.ev.ab * lg + c.ev.amb * q) + q2 * (c.ef * (c.priord - lg)));
}
// cnF2freq.cpp:4292-4313: the more probable value wins; true when the side's allele / sure are to be overwritten
// (non-empty individuals that have a prior, i.e. that were genotyped).  out[v] = the new probability of value v + 1.
CNF2_UHD bool certainty_pick(const double inf[2], const double out[2], int side, bool empty, bool has_prior,
                             int* new_allele, double* new_sure)
{
    int    best = 0;
    double bestprob = 0;
    for (int v = 0; v < 2; v++) {
        if (!(inf[v] > 0)) continue;
        if (out[v] > bestprob - (side ? 1e-30 : 0)) {          // cnF2freq.cpp:4292-4300
            best = v + 1;
            bestprob = out[v];
        }
    }
    if (!empty && (best != 0 || bestprob > 0) && has_prior) {
        *new_allele = best;
        *new_sure = 1.0 - bestprob;
        return true;
    }
    return false;
}
CNF2_UHD bool update_certainty(const double inf[2], const SideState& s, int side, bool empty, bool has_prior,
                               int children, const StepControl& sc, int* hits, int* new_allele, double* new_sure)
{
    double out[2] = {0, 0};
    for (int v = 0; v < 2; v++) {
        CertaintyFlow c;
        if (!certainty_flow_setup(inf, v, s, children, sc, &c)) continue;
        out[v] = flow_step([&](double x) { return certainty_rgradient(c, x); }, c.curprob, c.epsilon, sc.scalefactor, hits,
                           false);
    }
    return certainty_pick(inf, out, side, empty, has_prior, new_allele, new_sure);
}

// ------------------------------------------------------------------ phase-consistency ratio (relskewhmm)
// The two-state chain over the markers [first, end) of one chromosome: emission (1 - w, w) from the haplotype
// weights, transition "stay" with probability relhaplo[m].  ratio[m - first] = posterior of state 1 as the
// constructor leaves it (cnF2freq.cpp:4340-4441; forward values include the emission at m, the backward pass
// includes the emissions after m).  fw is scratch of 2 (end - first) doubles.
CNF2_UHD void phase_ratio(const double* hw, const double* relhaplo, int first, int end, double* fw, double* ratio)
{
    double s0 = 0.5, s1 = 0.5;
    auto emit = [&](int m) {
        const double w = hw[m];
        s0 *= fabs(1 - w);
        s1 *= fabs(0 - w);
    };
    auto move = [&](int m) {
        const double n = relhaplo[m], nb = 1 - n;
        const double t0 = s0 * n + s1 * nb, t1 = s1 * n + s0 * nb;
        s0 = t0;
        s1 = t1;
    };
    auto rescue = [&]() {
        if (s0 + s1 < 1e-10) {
            s0 *= 1e20;
            s1 *= 1e20;
        }
    };
    for (int m = first; m < end; m++) {
        emit(m);
        fw[(m - first) * 2] = s0;
        fw[(m - first) * 2 + 1] = s1;
        rescue();
        move(m);
    }
    s0 = s1 = 0.5;
    const int last = end - first - 1;
    ratio[last] = fw[last * 2 + 1] / (fw[last * 2] + fw[last * 2 + 1]);
    for (int m = end - 2; m >= first; m--) {
        emit(m + 1);
        move(m);
        rescue();
        const double r0 = s0 * fw[(m - first) * 2], r1 = s1 * fw[(m - first) * 2 + 1];
        ratio[m - first] = r1 / (r0 + r1);
    }
}

// ------------------------------------------------------------------ haplotype weights (updatehaploweights)
// One marker of one individual whose chromosome has any information (some haplocount != 0) and whose weight is
// neither 0 nor 1 (locked).  haplobase / haplocount are read AND rewritten (the reference leaves the adjusted
// values in place, cnF2freq.cpp:4660-4676: they are what a later pass of the same iteration sees).
struct HaploFlow {
    Evidence ev;
    double   ent, phaseratio, descendants;
    double   epsilon;
};
// rewrites haplobase / haplocount (cnF2freq.cpp:4660-4677) and forms the gradient's data
CNF2_UHD void haplo_flow_setup(double hw, double* haplobase, double* haplocount, int a0, int a1, double sure0, double sure1,
                               double phaseratio, int children, int descendants, const StepControl& sc, HaploFlow* h)
{
    const double scorea = 1.0 - sure0;
    double       scoreb = 1.0 - sure1;
    if (a0 != a1) scoreb = 1 - scoreb;
    double similarity = scorea * scoreb + (1 - scorea) * (1 - scoreb);
    if (!*haplocount || similarity == 1.0) {
        *haplocount = (*haplocount < 1.0) ? 1.0 : *haplocount;
        *haplobase  = hw * *haplocount;
    } else {
        if (similarity >= CNF2_SIMILARITY_CAP) similarity = CNF2_SIMILARITY_CAP;
        double count = *haplocount;
        *haplobase -= count * hw;
        count = count - similarity * count;
        *haplobase += count * hw;
        *haplobase *= *haplocount / count;
        if (*haplobase < 0) *haplobase = 0;
        if (*haplobase >= *haplocount) *haplobase = *haplocount;
    }
    h->ev = evidence_terms(hw, *haplobase, *haplocount);
    h->ent = (1 - similarity) * sc.entropyfactor;
    h->phaseratio = phaseratio;
    h->descendants = descendants;
    h->epsilon = clamp_distance(children);
}
// gradient = data + phase consistency + entropy = (N + Q^2 E) / Q^2; its reciprocal as one quotient
CNF2_UHD double haplo_rgradient(const HaploFlow& h, double x)
{
    const double lg = logit(x);
    const double q = h.ev.b * (1.0 - x) + h.ev.a * x, q2 = q * q;
    const double e = upd_div(h.phaseratio - x, x - x * x) * h.descendants - h.ent * lg;
    return upd_div(q2, (h.ev.ab * lg + h.ev.amb * q) + q2 * e);
}
CNF2_UHD double update_haploweight(double hw, double* haplobase, double* haplocount, int a0, int a1, double sure0,
                                   double sure1, double phaseratio, int children, int descendants,
                                   const StepControl& sc, bool breakathalf, int* hits)
{
    HaploFlow h;
    haplo_flow_setup(hw, haplobase, haplocount, a0, a1, sure0, sure1, phaseratio, children, descendants, sc, &h);
    return flow_step([&](double x) { return haplo_rgradient(h, x); }, hw, h.epsilon, sc.scalefactor, hits, breakathalf);
}

// Step-size control after an update pass (cnF2freq.cpp:6373-6392; `any` is false without the inversion machinery).
struct StepHistory {
    int oldhits = 0, oldhits2 = 0;
};
CNF2_UHD void adapt_scalefactor(StepControl* sc, StepHistory* h, int hits, int n_analysed)
{
    const int  mx = h->oldhits > h->oldhits2 ? h->oldhits : h->oldhits2;
    const int  mn = h->oldhits < h->oldhits2 ? h->oldhits : h->oldhits2;
    const int  floorhits = n_analysed / 7;                    // dous.size() / TURNBITS
    const bool bad = hits > mx;
    if (bad) sc->scalefactor /= 1.1;
    const bool good = hits < (mn > floorhits ? mn : floorhits) * 0.99;
    if (good) sc->scalefactor *= 1.21;
    sc->scalefactor *= 0.997;
    h->oldhits2 = h->oldhits;
    h->oldhits  = hits;
}

} // namespace cnf2
#endif
