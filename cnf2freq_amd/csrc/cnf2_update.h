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
// forced: a call inside the flow kernels' loops costs a register save / restore through scratch memory
#define CNF2_UHD __host__ __device__ __forceinline__
#define CNF2_LI __attribute__((always_inline))     /* on a lambda: its call operator is inlined as well */
#else
#define CNF2_LI
#define CNF2_UHD inline
#endif

// Floating-point contraction: the device compiler's default fuses multiply-adds ACROSS statements, differently in every
// context a function is inlined into, so the same gradient evaluated in two kernels could differ in its last bit -- and
// where a gradient is rounding noise (oracle/pyiter.py: "ill-conditioned elements") that decides the result.  Within this
// header a multiply-add is fused where the source writes it as one expression, nowhere else: every kernel that inlines
// these functions computes the same bits (restored to the default at the end of the file).
#if defined(__clang__)
#pragma clang fp contract(on)
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

// the same rule on a constant integrand c, in the same order of operations (see FlowState::pinned): the sum does not
// depend on the interval, so a pinned flow forms it once
CNF2_UHD double gauss15_const_sum(double c)
{
    double acc = c * gl15_weight(0);
    for (int i = 1; i < 8; i++) acc += (c + c) * gl15_weight(i);
    return acc;
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
// pinned: a value that sits on the clamp (epsilon or 1 - epsilon: a genotype that is as certain as it may get, a
// resolved phase) and is pushed further out.  Every position the bisection then asks about -- the midpoints and all 15
// nodes of every quadrature lie between the start and the bracket's outer end -- is clamped back onto the start, so the
// integrand is the constant g0 = 1 / gradient(start): the whole flow needs no gradient evaluation beyond the first, and
// its (up to 51) steps are a dozen instructions each.  In the steady state of a run most flows are of this kind.
struct FlowState {
    double orig, epsilon, lolim, hilim, lo, hi;
    double g0;            // dt/dp at the (clamped) starting point
    double csum;          // pinned: the quadrature rule's sum on the constant g0
    double mid, qa, qb;   // the step in flight: its midpoint, and the interval of its quadrature (flow_try -> flow_quadrature)
    unsigned long long path;   // the decisions so far, oldest in the highest used bit: 1 = the lower end moved up to the midpoint
    int    it;            // steps taken so far
    int    quads;         // steps that needed the quadrature (diagnostics)
    int    spared;        // steps whose quadrature the bound made unnecessary (diagnostics)
    int    why;           // how the bisection ended: 1 tolerance met, 2 interval under 1e-10, 3 steps used up / out of bounds (diagnostics)
    bool   falling;       // the gradient at the starting point is negative
    bool   live;          // more steps to take
    bool   pinned;
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
    f->path = 0;
    f->quads = 0;
    f->spared = 0;
    f->why = 0;
    f->live = scalefactor != 0;
    f->g0 = g0;
    f->pinned = isfinite(g0) && (f->falling ? f->orig <= epsilon : f->orig >= top);
    f->csum = f->pinned ? gauss15_const_sum(g0) : 0.0;
}
// no bound on the quadrature's value: every same-sign step runs it
struct NoTimeBound {
    CNF2_UHD bool operator()(double, double, double, double) const { return false; }
};
// One bisection step in two parts, so that a wavefront can run the cheap part for all its flows and gather the ones
// that need a quadrature until there are enough of them to run it for (the kernels of cnf2_kernels.hip).
//   flow_try        everything up to the quadrature: 0 = the flow has ended, 1 = the step is done and another follows,
//                   2 = the step needs the quadrature over [f->qa, f->qb] (call flow_quadrature).
//                   bound(xa, xb, |1 / G(mid)|, limit): is what the quadrature would report certainly under limit?
//                   (flow_time_under).
//   flow_quadrature the rest of such a step: 0 / 1 as above.
CNF2_UHD int flow_decide(FlowState* f, double t, double scalefactor)
{
    if (fabs(t - scalefactor) < scalefactor * 1e-3) {
        f->why = 1;
        f->live = false;
        return 0;
    }
    const bool up = (t < scalefactor) != f->falling;
    if (up) f->lo = f->mid;
    else f->hi = f->mid;
    f->path = (f->path << 1) | (up ? 1ull : 0ull);
    return 1;
}
// the bracket after `steps` completed steps with the decisions `path` (flow_decide), from a freshly begun flow: how a flow
// that was set aside is taken up again without its state having been stored
CNF2_UHD void flow_replay(FlowState* f, unsigned long long path, int steps)
{
    for (int k = steps - 1; k >= 0; k--) {
        const double mid = (f->lo + f->hi) / 2;
        if ((path >> k) & 1ull) f->lo = mid;
        else f->hi = mid;
    }
    f->it = steps;
    f->path = path;
}
template <class G, class B = NoTimeBound>
CNF2_UHD int flow_try(FlowState* f, G&& rgradient, double scalefactor, B&& bound = B())
{
    if (!f->live) return 0;
    if (f->it >= 51 || f->lo > f->hilim || f->hi < f->lolim) {   // done / outside the true bounds
        f->why = 3;
        f->live = false;
        return 0;
    }
    f->it++;
    const double mid = (f->lo + f->hi) / 2;
    f->mid = mid;
    const double gm = f->pinned ? f->g0 : flow_pace(rgradient, mid, f->epsilon);
    if (((gm < 0) != f->falling) || !isfinite(gm))
        return flow_decide(f, (scalefactor + 0.1) * 1.1, scalefactor);     // the gradient turns round before mid: too far
    double a = f->orig, b = mid;
    if (a > b) {
        const double s = a;
        a = b;
        b = s;
    }
    if (b - a < 1e-10) {
        f->why = 2;
        f->live = false;
        return 0;
    }
    // "not there yet" without the quadrature: 0 < t <= bound < the tolerance band (see flow_time_bound)
    if (!f->pinned && a >= f->epsilon && b <= 1.0 - f->epsilon &&
        bound(a, b, fabs(gm), scalefactor * (1.0 - 1e-3) * (1.0 - 1e-9))) {
        f->spared++;
        return flow_decide(f, 0.0, scalefactor);
    }
    if (f->pinned) {
        f->quads++;
        double t = ((b - a) * 0.5) * f->csum;
        if (b != mid) t = -t;
        if (!isfinite(t)) t = (scalefactor + 0.1) * 1.1;
        return flow_decide(f, t, scalefactor);
    }
    f->qa = a;
    f->qb = b;
    return 2;
}
template <class G>
CNF2_UHD int flow_quadrature(FlowState* f, G&& rgradient, double scalefactor)
{
    f->quads++;
    const double eps = f->epsilon;
    double       t = gauss15([&](double v) CNF2_LI { return flow_pace(rgradient, v, eps); }, f->qa, f->qb);
    if (f->qb != f->mid) t = -t;
    if (!isfinite(t)) t = (scalefactor + 0.1) * 1.1;
    return flow_decide(f, t, scalefactor);
}
// one whole step; returns whether another one follows
template <class G, class B = NoTimeBound>
CNF2_UHD bool flow_advance(FlowState* f, G&& rgradient, double scalefactor, B&& bound = B())
{
    int r = flow_try(f, rgradient, scalefactor, bound);
    if (r == 2) r = flow_quadrature(f, rgradient, scalefactor);
    return r == 1;
}
// The same step sequence as a machine that asks for ONE gradient evaluation at a time: the form the device kernels run.
// A wavefront holds 64 flows at different points of their bisections -- some at a midpoint, some in the middle of a
// quadrature -- and every round each of them wants exactly one value of its gradient: all lanes execute the expensive
// part (the gradient) together, whatever they are going to do with the value.
//   flow_want(r, sf, &v)   the (clamped) position to evaluate next; false: the flow has ended
//   flow_feed(r, value, sf, bound)   the reciprocal gradient there
// Order of operations and results are those of flow_try / flow_quadrature (gauss15: centre, then the node pairs outwards).
struct FlowRun {
    FlowState f;
    int       phase;      // 0: the next value is a step's midpoint; k = 1..15: it is evaluation k of the quadrature in flight
    double    acc, fp;    // quadrature sum so far; value at the +node that waits for its partner
    double    mc, half;   // centre and half width of the quadrature's interval
};
CNF2_UHD bool flow_want(FlowRun* r, double scalefactor, double* v)
{
    FlowState* f = &r->f;
    double     x;
    if (r->phase == 0) {
        if (!f->live) return false;
        if (f->it >= 51 || f->lo > f->hilim || f->hi < f->lolim) {   // done / outside the true bounds
            f->why = 3;
            f->live = false;
            return false;
        }
        f->it++;
        f->mid = (f->lo + f->hi) / 2;
        x = f->mid;
    } else if (r->phase == 1) {
        x = r->mc;
    } else {
        const int    i = r->phase >> 1;
        const double z = gl15_node(i);
        x = r->mc + r->half * ((r->phase & 1) ? -z : z);
    }
    const double top = 1.0 - f->epsilon;
    *v = (x < f->epsilon) ? f->epsilon : ((top < x) ? top : x);
    return true;
}
template <class B = NoTimeBound>
CNF2_UHD void flow_feed(FlowRun* r, double value, double scalefactor, B&& bound = B())
{
    FlowState* f = &r->f;
    if (r->phase == 0) {
        const double gm = value;
        if (((gm < 0) != f->falling) || !isfinite(gm)) {
            flow_decide(f, (scalefactor + 0.1) * 1.1, scalefactor);     // the gradient turns round before mid: too far
            return;
        }
        double a = f->orig, b = f->mid;
        if (a > b) {
            const double s = a;
            a = b;
            b = s;
        }
        if (b - a < 1e-10) {
            f->why = 2;
            f->live = false;
            return;
        }
        if (a >= f->epsilon && b <= 1.0 - f->epsilon &&
            bound(a, b, fabs(gm), scalefactor * (1.0 - 1e-3) * (1.0 - 1e-9))) {
            f->spared++;
            flow_decide(f, 0.0, scalefactor);
            return;
        }
        f->qa = a;
        f->qb = b;
        f->quads++;
        r->mc = (a + b) * 0.5;
        r->half = (b - a) * 0.5;
        r->phase = 1;
        return;
    }
    if (r->phase == 1) {
        r->acc = value * gl15_weight(0);
        r->phase = 2;
        return;
    }
    if (!(r->phase & 1)) {
        r->fp = value;
        r->phase++;
        return;
    }
    r->acc += (r->fp + value) * gl15_weight(r->phase >> 1);
    if (r->phase < 15) {
        r->phase++;
        return;
    }
    double t = r->half * r->acc;
    if (f->qb != f->mid) t = -t;
    if (!isfinite(t)) t = (scalefactor + 0.1) * 1.1;
    r->phase = 0;
    flow_decide(f, t, scalefactor);
}

CNF2_UHD double flow_end(const FlowState& f, double scalefactor, int* hits, bool breakathalf)
{
    double lo = f.lo, hi = f.hi;
    if (!scalefactor) lo = hi = f.orig;
    return cap_step((lo + hi) / 2, f.orig, f.epsilon, hits, breakathalf);
}
template <class G, class B = NoTimeBound>
CNF2_UHD double flow_step(G&& rgradient, double orig, double epsilon, double scalefactor, int* hits, bool breakathalf,
                          B&& bound = B())
{
    FlowState f;
    flow_begin(&f, rgradient, orig, epsilon, scalefactor, breakathalf);
    while (flow_advance(&f, rgradient, scalefactor, bound)) {}
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

// ------------------------------------------------------------------ bounds that spare most quadratures
// Both gradients have the form  G(x) = D(x) - e L(x) + d W(x) + c0,  D the data term above, L = logit,
// W(x) = (pr - x) / (x (1 - x)) the phase-consistency term (d = 0 for the certainties), e, d >= 0.  With u = x (1 - x):
//     -G'(x) = e / u - a b / (u Q^2) + (a - b)^2 / Q^2 + 2 a b (a - b) L / Q^3 + d ((x - pr)^2 + pr (1 - pr)) / u^2 .
// In the steady state of a run most values sit next to a root x* of G that attracts them (near 0 or 1 the entropy term
// e / u is in the hundreds), the time to reach it is infinite, and the bisection of flow_advance() closes in on x* from
// both sides until its 51 steps are used up; on the near side every step costs a 15-point quadrature whose only message
// is "not there yet".  That message can be had from the interval [orig, mid] alone: if s1 > 0 is a lower bound of -G' on
// it (interval arithmetic on the expression above: u is concave, Q linear, L monotone), G is strictly monotone there,
// has no root inside, and |G(x)| >= |G(mid)| + s1 |mid - x| =: l(x).  The rule has positive weights and, for the
// reciprocal of a linear function, never exceeds the integral (its error term has the sign of the 30th derivative), so
//     t = |rule[1 / G]| <= rule[1 / l] <= integral of 1 / l = log(1 + s1 |mid - orig| / |G(mid)|) / s1 ,
// and because the node nearest to mid keeps 0.6 % of the interval's length away from it, even G(mid) = 0 gives
//     t <= rule[1 / (s1 |mid - x|)] = C15 / s1,   C15 = sum of w_i / (1 - z_i) = 6.636...
// When one of the two is under the tolerance band the decision of the step is known -- identical to the literal one,
// not an approximation of it -- for well under a hundred instructions instead of 1 300.  No bound (s1 <= 0, the
// interval touching the clamp, a NaN anywhere): the quadrature runs as before.
// Rounding: a computed gradient value is its true value up to `noise` (1e-14 times the magnitude of its terms, bounded
// over the interval like the slope); a value only counts where it stands clear of that, and the nodes of a rule must.
struct SlopeTerms {
    Evidence ev;
    double   e;        // coefficient of -logit: entropyfactor (certainties), (1 - similarity) entropyfactor (weights)
    double   d, pr;    // descendants and phase ratio of the weights' third term; d = 0: none
    double   c0;       // constant term: entropyfactor x prior (certainties)
};
#define CNF2_GL15_C 6.636457986458086      /* sum of w_i / (1 - z_i) over the 15 nodes */
struct IntervalFacts {
    double s1;         // lower bound of -G' on the interval; <= 0 or NaN: none
    double noise;      // bound of the rounding error of a computed G on the interval
};
// fine: the bound of |logit| is a logarithm (once per flow); else 1 / min(x, 1 - x) (per step: crude, but the terms it
// enters are small where the bounds matter)
CNF2_UHD IntervalFacts flow_interval(const SlopeTerms& s, double xa, double xb, bool fine)
{
    IntervalFacts F;
    F.s1 = 0.0;
    F.noise = HUGE_VAL;
    const double ua = xa * (1.0 - xa), ub = xb * (1.0 - xb);
    const double ulo = ua < ub ? ua : ub;
    const double uhi = (xa <= 0.5 && 0.5 <= xb) ? 0.25 : (ua < ub ? ub : ua);
    const double qa = s.ev.b * (1.0 - xa) + s.ev.a * xa, qb = s.ev.b * (1.0 - xb) + s.ev.a * xb;
    const double qlo = qa < qb ? qa : qb, qhi = qa < qb ? qb : qa;
    if (!(ulo > 0.0) || !(qlo > 0.0) || !(s.e >= 0.0) || !(s.d >= 0.0)) return F;
    const double edge = xa < 1.0 - xb ? xa : 1.0 - xb;
    // |logit(x)| <= -log(min(x, 1 - x)) <= 1 / min(x, 1 - x); and -log(m 2^k) <= -k log 2 for a mantissa m in [1, 2)
    double lmax;
    if (fine) {
        int k;
        (void)frexp(edge, &k);                               // edge = m' 2^k with m' in [1/2, 1): edge >= 2^(k - 1)
        lmax = (double)(1 - k) * 0.69314718055994530942;
    } else {
        lmax = upd_div(1.0, edge);
    }
    const double amb = s.ev.amb < 0 ? -s.ev.amb : s.ev.amb;
    const double r_uhi = upd_div(1.0, uhi), r_ulo = upd_div(1.0, ulo), r_qlo = upd_div(1.0, qlo), r_qhi = upd_div(1.0, qhi);
    // -G' >= e / uhi - (a b / ulo + 2 a b |a - b| lmax / qlo) / qlo^2 + (a - b)^2 / qhi^2 + d num / uhi^2
    double s1 = s.e * r_uhi - s.ev.ab * (r_ulo + 2.0 * amb * r_qlo * lmax) * (r_qlo * r_qlo) + (s.ev.amb * r_qhi) * (s.ev.amb * r_qhi);
    double mag = (s.ev.ab * lmax + amb * qhi) * (r_qlo * r_qlo) + s.e * lmax + fabs(s.c0);
    if (s.d != 0.0) {
        const double dist = s.pr < xa ? xa - s.pr : (s.pr > xb ? s.pr - xb : 0.0);
        const double num = dist * dist + s.pr * (1.0 - s.pr);      // >= 0 for a ratio in [0, 1]
        s1 += s.d * (num >= 0.0 ? num * (r_uhi * r_uhi) : num * (r_ulo * r_ulo));
        const double fa = fabs(s.pr - xa), fb = fabs(s.pr - xb);
        mag += s.d * (fa > fb ? fa : fb) * r_ulo;
    }
    F.s1 = s1 * (1.0 - 1e-9);                               // the bound itself is rounded
    F.noise = 1e-14 * mag;
    return F;
}
// Is what the rule reports over [xa, xb] certainly under `limit`?  One end of the interval is the flow's start, the other
// the midpoint, where the COMPUTED reciprocal gradient is pace (same sign as at the start); xa < xb, both inside the
// clamp.  false = not known.
CNF2_UHD bool flow_under(const IntervalFacts& F, double width, double pace, double limit)
{
    if (!(F.s1 > 0.0) || !(F.noise < HUGE_VAL)) return false;
    // the true root may lie before the midpoint by the noise; the nodes must be clear of it
    if (!(0.006 * width * F.s1 > 8.0 * F.noise)) return false;
    const double ls = limit * F.s1;
    if (ls > 1.02 * CNF2_GL15_C) return true;                // the rule cannot report more than C15 / s1
    const double g = upd_div(1.0, pace);                     // |G(mid)| as computed
    if (!(g > 4.0 * F.noise)) return false;
    const double z = F.s1 * width * upd_div(1.0, g - 2.0 * F.noise);
    if (z < ls) return true;                                 // log(1 + z) <= z
#if defined(__HIP_DEVICE_COMPILE__)
    return false;                                            // the logarithm costs the device more than the quadratures it would spare
#else
    if (!(z < 1e300)) return false;
    return log1p(z) < ls;
#endif
}
CNF2_UHD bool flow_time_under(const SlopeTerms& s, double xa, double xb, double pace, double limit)
{
    return flow_under(flow_interval(s, xa, xb, false), xb - xa, pace, limit);
}
// the bound itself (tests): the smallest limit flow_time_under accepts, +infinity when there is none
CNF2_UHD double flow_time_bound(const SlopeTerms& s, double xa, double xb, double pace)
{
    if (!flow_time_under(s, xa, xb, pace, 1e300)) return HUGE_VAL;
    double lo = 0.0, hi = 1.0;
    while (!flow_time_under(s, xa, xb, pace, hi) && hi < 1e300) hi *= 2.0;
    for (int i = 0; i < 200; i++) {
        const double mid = 0.5 * (lo + hi);
        if (flow_time_under(s, xa, xb, pace, mid)) hi = mid;
        else lo = mid;
    }
    return hi;
}

// ------------------------------------------------------------------ the scout: steps without quadratures, cheaply
// Most flows of a run in its steady state never need a quadrature: they close in on a root of their gradient (or find it
// within 1e-10 of their start), and every step is settled by a sign or by the bounds above.  Settled that way a flow
// still pays one gradient evaluation per step, up to 51 of them, to locate the root by bisection.  Where the gradient has
// been shown strictly monotone on the whole bracket (flow_interval over [lo, hi] and the start), every evaluation is a
// fact about ONE root: a point whose value has the start's sign -- by more than the rounding noise -- has the root
// beyond it, and so has every point before it; a point with the other sign has it before.  The scout keeps the two
// innermost such points and answers a midpoint's sign from them when it can; between them it closes in on the root
// superlinearly (false position with the Illinois modification; each of its evaluations is a fact of the same kind),
// so that the bisection's own midpoints need an evaluation only in the last few steps, inside the band where computed
// signs are noise and only the evaluation itself says what the literal algorithm sees.  The step sequence, and with it
// the result, is the literal one; what changes is how much of it is computed.
// flow_scout runs a begun flow until it ends (returns 0) or a step needs a quadrature (returns 2: the steps completed
// and their decisions are in f->it and f->path; flow_replay + flow_advance take it from there).
// Written for wavefronts that run 64 scouts in lock step (all start at step 0): what costs instructions -- the attempt
// to show the gradient monotone, the closing in on the root -- happens at fixed step numbers, the same for every lane,
// and a same-sign step that the constant bound C15 / s1 does not settle ends the scout (the finish pass has the finer
// bound and the quadrature).
// max_steps: give up after that many steps of this call (returns 3: the flow is neither ended nor at a quadrature; its
// completed steps are in f->it and f->path like for 2) -- the first of two scout passes runs a few steps of every flow,
// the second the rest of the few that are still going, with wavefronts full of them.
template <class G>
CNF2_UHD int flow_scout(FlowState* f, G&& rgradient, const SlopeTerms& st, double scalefactor, int* evaluations, int max_steps = 1 << 30)
{
    const double eps = f->epsilon, top = 1.0 - f->epsilon;
    const double limit = scalefactor * (1.0 - 1e-3) * (1.0 - 1e-9);
    IntervalFacts B;                   // facts about the bracket once the gradient is known to be monotone on it
    B.s1 = 0.0;
    B.noise = HUGE_VAL;
    bool          mono = false, refined = false, settles = false;
    double        zmax = 0.0;                       // exp(limit s1) - 1
    double        near_d = 0.0, far_d = HUGE_VAL;   // distance from the start of the farthest point solidly on its side of the root, of the nearest solidly beyond
    double        near_g = 0.0, far_g = 0.0;        // computed gradients there (near_g at distance 0: the start's, when solid)
    int           evals = 0;
    const double  dir = f->falling ? -1.0 : 1.0;    // the flow moves towards orig + dir d
    auto too_far = [&](double gm) CNF2_LI { return ((gm < 0) != f->falling) || !isfinite(gm); };    // the literal test of a step
    auto note = [&](double d, double gm) CNF2_LI {  // an evaluation at distance d > 0 is a fact about the root, if it is solid
        const double g = 1.0 / gm;
        const bool   solid = isfinite(gm) && gm != 0.0 && fabs(g) > 2.0 * B.noise;
        const bool   beyond = too_far(gm);
        const bool   new_far = solid && beyond && d < far_d, new_near = solid && !beyond && d > near_d;
        far_d = new_far ? d : far_d;
        far_g = new_far ? g : far_g;
        near_d = new_near ? d : near_d;
        near_g = new_near ? g : near_g;
    };
    for (int step = 0;; step++) {
        *evaluations = evals;
        if (!f->live) return 0;
        if (f->it >= 51 || f->lo > f->hilim || f->hi < f->lolim) {
            f->why = 3;
            f->live = false;
            return 0;
        }
        if (step >= max_steps) return 3;
        // is the gradient monotone on the bracket (and on the way from the start)?  Tried while the bracket shrinks.
        if (step == 0 || step == 1 || step == 2 || step == 4 || step == 7 || step == 11 || step == 16) {
            if (!mono && f->lo >= eps && f->hi <= top) {
                const double xa = f->lo < f->orig ? f->lo : f->orig, xb = f->hi > f->orig ? f->hi : f->orig;
                B = flow_interval(st, xa, xb, true);
                if (B.s1 > 0.0 && B.noise < HUGE_VAL) {
                    mono = true;
                    settles = limit * B.s1 > 1.02 * CNF2_GL15_C;       // the rule cannot report more than C15 / s1
                    const double ls = limit * B.s1;
                    zmax = (ls < 700.0 ? exp(ls) : 1e300) - 1.0;
                    const double g_start = 1.0 / f->g0;
                    if (fabs(g_start) > 2.0 * B.noise) near_g = g_start;       // the start itself (distance 0) is a solid point
                }
            }
        }
        // the root is bracketed by two solid facts: close in on it (false position, Illinois), each evaluation one more
        if ((step == 3 || step == 6 || step == 10 || step == 15 || step == 21) && mono && !refined && far_d < HUGE_VAL &&
            near_g != 0.0) {
            refined = true;
            double dn = near_d, df = far_d, fn = near_g, ff = far_g;
            int    side = 0;
            for (int k = 0; k < 16; k++) {
                double dp = (dn * ff - df * fn) / (ff - fn);
                if (!(dp > dn && dp < df)) dp = 0.5 * (dn + df);
                const double xp = f->orig + dir * dp;
                const double dd = (xp - f->orig) * dir;              // the distance as the steps will measure it
                if (!(xp >= eps && xp <= top) || !(dd > dn && dd < df)) break;
                const double gp = flow_pace(rgradient, xp, eps);
                evals++;
                if (!isfinite(gp) || gp == 0.0 || !(fabs(1.0 / gp) > 2.0 * B.noise)) break;    // inside the noise band: as close as facts get
                note(dd, gp);
                if (too_far(gp)) {
                    df = dd;
                    ff = 1.0 / gp;
                    if (side == 1) fn *= 0.5;
                    side = 1;
                } else {
                    dn = dd;
                    fn = 1.0 / gp;
                    if (side == -1) ff *= 0.5;
                    side = -1;
                }
            }
        }
        f->it++;
        const double mid = (f->lo + f->hi) / 2;
        f->mid = mid;
        const double d = (mid - f->orig) * dir;
        bool   far, have_gm = false;
        double gm = 0.0;
        if (mono && d > 0.0 && d >= far_d) far = true;
        else if (mono && d > 0.0 && d <= near_d) far = false;
        else {
            gm = flow_pace(rgradient, mid, eps);
            evals++;
            have_gm = true;
            far = too_far(gm);
            if (mono && d > 0.0) note(d, gm);
        }
        if (far) {
            flow_decide(f, (scalefactor + 0.1) * 1.1, scalefactor);
            continue;
        }
        double a = f->orig, b = mid;
        if (a > b) {
            const double t = a;
            a = b;
            b = t;
        }
        if (b - a < 1e-10) {
            f->why = 2;
            f->live = false;
            *evaluations = evals;
            return 0;
        }
        // "not there yet" from the slope bound of the whole bracket (the nodes of the rule must be clear of the noise):
        // the constant form C15 / s1 where that is under the band; else log(1 + s1 d / |G(mid)|) / s1 < limit, i.e.
        // s1 d / |G(mid)| < exp(limit s1) - 1 =: zmax, with |G(mid)| >= |G| at the farthest same-sign point known
        // (the midpoint lies before it and |G| falls towards the root), or with the midpoint's own value
        bool spared = false;
        if (mono && a >= eps && b <= top && 0.006 * (b - a) * B.s1 > 8.0 * B.noise) {
            spared = settles;
            if (!spared && d <= near_d && fabs(near_g) > 4.0 * B.noise)
                spared = B.s1 * d < zmax * (fabs(near_g) - 2.0 * B.noise);
            if (!spared) {
                if (!have_gm) {
                    gm = flow_pace(rgradient, mid, eps);
                    evals++;
                    have_gm = true;
                    note(d, gm);
                }
                const double g = fabs(1.0 / gm);
                if (!too_far(gm) && g > 4.0 * B.noise) spared = B.s1 * d < zmax * (g - 2.0 * B.noise);
            }
        }
        if (!spared) {
            f->it--;                   // this step is for the finish pass: hand the flow over as it stood before it
            *evaluations = evals;
            return 2;
        }
        f->spared++;
        flow_decide(f, 0.0, scalefactor);
    }
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
CNF2_UHD SlopeTerms certainty_slope(const CertaintyFlow& c)
{
    SlopeTerms s;
    s.ev = c.ev;
    s.e = c.ef;
    s.d = 0.0;
    s.pr = 0.0;
    s.c0 = c.ef * c.priord;
    return s;
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
// literal: every same-sign step runs its quadrature, as the reference's cappedgd does (the yardstick of the other forms)
CNF2_UHD bool update_certainty(const double inf[2], const SideState& s, int side, bool empty, bool has_prior,
                               int children, const StepControl& sc, int* hits, int* new_allele, double* new_sure,
                               bool literal = false)
{
    double out[2] = {0, 0};
    for (int v = 0; v < 2; v++) {
        CertaintyFlow c;
        if (!certainty_flow_setup(inf, v, s, children, sc, &c)) continue;
        const SlopeTerms st = certainty_slope(c);
        out[v] = flow_step([&](double x) CNF2_LI { return certainty_rgradient(c, x); }, c.curprob, c.epsilon, sc.scalefactor, hits,
                           false, [&](double xa, double xb, double pc, double lim) CNF2_LI {
                               return !literal && flow_time_under(st, xa, xb, pc, lim);
                           });
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
// similarity of the two sides of a genotype (cnF2freq.cpp:4643-4658) and the rewrite of haplobase / haplocount with it
// (cnF2freq.cpp:4660-4677); returns the similarity the entropy term uses
CNF2_UHD double haplo_similarity(int a0, int a1, double sure0, double sure1)
{
    const double scorea = 1.0 - sure0;
    double       scoreb = 1.0 - sure1;
    if (a0 != a1) scoreb = 1 - scoreb;
    return scorea * scoreb + (1 - scorea) * (1 - scoreb);
}
CNF2_UHD double haplo_rewrite(double hw, double* haplobase, double* haplocount, double similarity)
{
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
    return similarity;
}
// the gradient's data from the REWRITTEN haplobase / haplocount and the similarity haplo_rewrite returned
CNF2_UHD void haplo_flow_terms(double hw, double haplobase, double haplocount, double similarity, double phaseratio,
                               int children, int descendants, const StepControl& sc, HaploFlow* h)
{
    h->ev = evidence_terms(hw, haplobase, haplocount);
    h->ent = (1 - similarity) * sc.entropyfactor;
    h->phaseratio = phaseratio;
    h->descendants = descendants;
    h->epsilon = clamp_distance(children);
}
CNF2_UHD void haplo_flow_setup(double hw, double* haplobase, double* haplocount, int a0, int a1, double sure0, double sure1,
                               double phaseratio, int children, int descendants, const StepControl& sc, HaploFlow* h)
{
    const double similarity = haplo_rewrite(hw, haplobase, haplocount, haplo_similarity(a0, a1, sure0, sure1));
    haplo_flow_terms(hw, *haplobase, *haplocount, similarity, phaseratio, children, descendants, sc, h);
}
CNF2_UHD SlopeTerms haplo_slope(const HaploFlow& h)
{
    SlopeTerms s;
    s.ev = h.ev;
    s.e = h.ent;
    s.d = h.descendants;
    s.pr = h.phaseratio;
    s.c0 = 0.0;
    return s;
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
                                   const StepControl& sc, bool breakathalf, int* hits, bool literal = false)
{
    HaploFlow h;
    haplo_flow_setup(hw, haplobase, haplocount, a0, a1, sure0, sure1, phaseratio, children, descendants, sc, &h);
    const SlopeTerms st = haplo_slope(h);
    return flow_step([&](double x) CNF2_LI { return haplo_rgradient(h, x); }, hw, h.epsilon, sc.scalefactor, hits, breakathalf,
                     [&](double xa, double xb, double pc, double lim) CNF2_LI {
                         return !literal && flow_time_under(st, xa, xb, pc, lim);
                     });
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
#if defined(__clang__) && defined(__HIPCC__)
#pragma clang fp contract(fast)
#endif
#endif
