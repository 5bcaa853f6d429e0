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
#ifndef CNF2_SCOUT_DEEP
#define CNF2_SCOUT_DEEP (1.0 / 1048576.0)  /* flow_scout, hand_over: Euler step / bracket under which a flow stays with the scout */
#endif
#ifndef CNF2_SCOUT_NEAR_ROOT
#define CNF2_SCOUT_NEAR_ROOT 0.002         /* flow_scout, hand_over: (root - midpoint) / root under which a set-aside flow takes the step-per-round kernels (0.02 -> 0.002: probe update passes -2.4 %, late iterations -0.5 %, config 5 x 100 103.0 -> 101.5 s; 0.1: +5 % early; profiles/r05_zz_ab_hand_over_thresholds.log) */
#endif
#ifndef CNF2_SCOUT_SATURATING
#define CNF2_SCOUT_SATURATING 3.0          /* flow_scout, hand_over: slope towards the bracketed root x step size from which a set-aside flow takes the step-per-round kernels */
#endif
#ifndef CNF2_SCOUT_MONO_AT
#define CNF2_SCOUT_MONO_AT(step) (step == 0 || step == 1 || step == 2 || step == 4 || step == 7 || step == 11 || step == 16)   /* flow_scout: the steps at which the monotonicity proof is tried (again) */
#endif
#ifndef CNF2_SCOUT_KEEP
#define CNF2_SCOUT_KEEP 0.6                /* flow_scout, hand_over: slope x step size from which a flow stays with the scout */
#endif
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
    // |logit(x)| = log(max(x, 1 - x) / m), m = min(x, 1 - x), is at most |1 - 2 x| / m (log(1 + z) <= z) and at most -log m
    // <= -k log 2 for m = m' 2^k, m' in [1/2, 1) (fine: one frexp per end; the callers that ask per step do without);
    // it grows away from 1/2, so its bound on the interval is the larger of the ends'
    auto logit_bound = [&](double x) CNF2_LI {
        const double m = x < 1.0 - x ? x : 1.0 - x;
        double       l = upd_div(fabs(1.0 - 2.0 * x), m);
        if (fine) {
            int k;
            (void)frexp(m, &k);
            const double l2 = (double)(1 - k) * 0.69314718055994530942;
            l = l2 < l ? l2 : l;
        }
        return l;
    };
    const double la = logit_bound(xa), lb = logit_bound(xb);
    const double lmax = la > lb ? la : lb;
    // the term 2 a b (a - b) L / Q^3 of -G' counts only where it is negative: (a - b) > 0 and L < 0 (possible left of 1/2:
    // L >= logit(xa)), or (a - b) < 0 and L > 0 (right of 1/2: L <= logit(xb))
    const double lneg = s.ev.amb > 0 ? (xa < 0.5 ? la : 0.0) : (xb > 0.5 ? lb : 0.0);
    const double amb = s.ev.amb < 0 ? -s.ev.amb : s.ev.amb;
    const double r_uhi = upd_div(1.0, uhi), r_ulo = upd_div(1.0, ulo), r_qlo = upd_div(1.0, qlo), r_qhi = upd_div(1.0, qhi);
    // -G' >= (e - a b / qlo^2) / u - 2 a b |a - b| lneg / qlo^3 + (a - b)^2 / qhi^2 + d num / uhi^2: the two terms in 1 / u are
    // kept together (their coefficient taken at its lowest, then u where that is worst) -- bounded one by one, e / uhi less
    // a b / (ulo qlo^2), the bound fails for every value next to 0 or 1 whose evidence weighs about what the entropy does
    const double cu = s.e - s.ev.ab * (r_qlo * r_qlo);
    double s1 = (cu >= 0.0 ? cu * r_uhi : cu * r_ulo) - s.ev.ab * (2.0 * amb * r_qlo * lneg) * (r_qlo * r_qlo) + (s.ev.amb * r_qhi) * (s.ev.amb * r_qhi);
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

// An enclosure of the gradient itself on an interval inside the clamp: G = L (a b / Q^2 - e) + (a - b) / Q + c0 + d W with
// L = logit in [L(xa), L(xb)], Q between its values at the ends, u = x (1 - x) as above, every term taken at the corner of
// its box that is worst (the terms are treated as independent: wider than the truth, never narrower).  glo <= G <= ghi on
// [xa, xb]; lg_a, lg_b: logit at the ends as computed (widened here by their error).  Used where the flow is known to
// run into its cap: if |G| >= gmin > 0 on the whole bracket, the rule over [start, m] cannot report more than
// |m - start| / gmin, whatever the gradient does in between.
struct GradientRange {
    double glo, ghi;
};
CNF2_UHD GradientRange flow_gradient_range(const SlopeTerms& s, double xa, double xb, double lg_a, double lg_b)
{
    GradientRange R;
    R.glo = -HUGE_VAL;
    R.ghi = HUGE_VAL;
    const double ua = xa * (1.0 - xa), ub = xb * (1.0 - xb);
    const double ulo = ua < ub ? ua : ub;
    const double uhi = (xa <= 0.5 && 0.5 <= xb) ? 0.25 : (ua < ub ? ub : ua);
    const double qa = s.ev.b * (1.0 - xa) + s.ev.a * xa, qb = s.ev.b * (1.0 - xb) + s.ev.a * xb;
    const double qlo = qa < qb ? qa : qb, qhi = qa < qb ? qb : qa;
    if (!(ulo > 0.0) || !(qlo > 0.0) || !(xa < xb) || !isfinite(lg_a) || !isfinite(lg_b)) return R;
    const double wl = 1e-12 * (1.0 + fabs(lg_a)), wh = 1e-12 * (1.0 + fabs(lg_b));
    const double llo = lg_a - wl, lhi = lg_b + wh;
    const double r_qlo = upd_div(1.0, qlo), r_qhi = upd_div(1.0, qhi);
    const double klo = s.ev.ab * (r_qhi * r_qhi) - s.e, khi = s.ev.ab * (r_qlo * r_qlo) - s.e;
    const double p1 = llo * klo, p2 = llo * khi, p3 = lhi * klo, p4 = lhi * khi;
    double plo = p1 < p2 ? p1 : p2, phi = p1 < p2 ? p2 : p1;
    plo = p3 < plo ? p3 : plo;
    plo = p4 < plo ? p4 : plo;
    phi = p3 > phi ? p3 : phi;
    phi = p4 > phi ? p4 : phi;
    const double m1 = s.ev.amb * r_qlo, m2 = s.ev.amb * r_qhi;
    double lo = plo + (m1 < m2 ? m1 : m2) + s.c0, hi = phi + (m1 < m2 ? m2 : m1) + s.c0;
    if (s.d != 0.0) {
        const double r_ulo = upd_div(1.0, ulo), r_uhi = upd_div(1.0, uhi);
        const double na = s.pr - xb, nb = s.pr - xa;              // pr - x in [na, nb]
        const double w1 = na * r_ulo, w2 = na * r_uhi, w3 = nb * r_ulo, w4 = nb * r_uhi;
        double wlo = w1 < w2 ? w1 : w2, whi = w1 < w2 ? w2 : w1;
        wlo = w3 < wlo ? w3 : wlo;
        wlo = w4 < wlo ? w4 : wlo;
        whi = w3 > whi ? w3 : whi;
        whi = w4 > whi ? w4 : whi;
        lo += s.d * wlo;
        hi += s.d * whi;
    }
    const double mag = fabs(plo) + fabs(phi) + fabs(m1) + fabs(m2) + fabs(s.c0) + fabs(lo) + fabs(hi);
    R.glo = lo - 1e-12 * mag;
    R.ghi = hi + 1e-12 * mag;
    return R;
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
// and their decisions are in f->it and f->path; flow_replay + flow_advance take it from there; 5: likewise, and the gradient is
// not known to be monotone on the bracket; 4: set aside at once for the guided bisection, see hand_over; 6: set aside for the
// step-per-round kernels, see the end of the loop).
// Written for wavefronts that run 64 scouts in lock step (all start at step 0): what costs instructions -- the attempt
// to show the gradient monotone, the closing in on the root -- happens at fixed step numbers, the same for every lane,
// and a same-sign step that the constant bound C15 / s1 does not settle ends the scout (the finish pass has the finer
// bound and the quadrature).
// max_steps: give up after that many steps of this call (returns 3: the flow is neither ended nor at a quadrature; its
// completed steps are in f->it and f->path like for 2) -- the first of two scout passes runs a few steps of every flow,
// the second the rest of the few that are still going, with wavefronts full of them.
// find_root = false: no closing in on the root in this call (the far end's evaluation, the false-position loop, the
// tightening): the first of two scout passes leaves it to the second.  A flow that is still going after the first pass's
// steps starts again in the second and would find its root twice -- thirty evaluations, and the false-position loop as long
// as the wavefront's slowest lane -- while the flows that do end within the first pass's steps end there by signs all the
// same (probe's update passes -3.6 %, late iterations -2.6 %, profiles/r05_zzz_ab_first_pass_without_root.log).
// hand_over: the flows set aside go to the guided bisection (below), which needs 3 - 4 quadratures wherever the gradient is
// monotone, however far the scout has come: a flow whose gradient is monotone but whose steps the constant bound does not
// settle is set aside at once (what the scout would spend on it -- the closing in on the root, a dozen evaluations --
// buys the guided bisection nothing).
template <class G>
CNF2_UHD int flow_scout(FlowState* f, G&& rgradient, const SlopeTerms& st, double scalefactor, int* evaluations, int max_steps = 1 << 30,
                        bool hand_over = false, bool find_root = true)
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
    // A flow that cannot move (no finite gradient at its start, or a step size of zero: flow_begin has shut the bracket on the
    // start) takes its 51 steps at the start itself, every one "too far" by the same non-finite value: nothing to evaluate
    if (f->live && f->lo == f->hi && f->lo == f->orig && !(f->lo > f->hilim || f->hi < f->lolim)) {
        f->it = 51;
        f->why = 3;
        f->live = false;
        *evaluations = 0;
        return 0;
    }
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
        if (CNF2_SCOUT_MONO_AT(step)) {
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
                    if (hand_over && !settles) {
                        // Which kind of flow is it?  With |G| falling off linearly at the rate k the rule's value saturates at
                        // C15 / k next to the root: for k x step size well under C15 the flow reaches its step size a fair way
                        // before the root -- the guided bisection's case, set aside at once -- else it closes in on the root and
                        // the steps are settled here, by signs and bounds.  k from the start and one evaluation at the distance
                        // an Euler step would reach (a fact about the root like any other).
                        const double f0 = fabs(1.0 / f->g0);
                        const double far_end = f->falling ? f->orig - f->lo : f->hi - f->orig;
                        double       de = scalefactor * f0;
                        de = de < far_end ? de : far_end;
                        const double xe = f->orig + dir * de, dd = (xe - f->orig) * dir;
                        bool         regular = true;
                        if (!(dd > far_end * CNF2_SCOUT_DEEP)) regular = false;               // next to a root from the start: more steps to the band than a plan looks ahead
                        else if (xe >= eps && xe <= top) {
                            const double ge = flow_pace(rgradient, xe, eps);
                            evals++;
                            if (isfinite(ge) && ge != 0.0) {
                                const double fe = too_far(ge) ? -fabs(1.0 / ge) : fabs(1.0 / ge);
                                regular = !((f0 - fe) * scalefactor > CNF2_SCOUT_KEEP * dd);
                                note(dd, ge);
                            }
                        }
                        if (regular) {
                            *evaluations = evals;
                            return 4;
                        }
                    }
                }
            }
        }
        // A flow that is still here with a monotone gradient closes in on a root (or runs into its cap): look for the root at
        // once -- one evaluation at the bracket's far end, and if the gradient has turned round there the closing in below --
        // instead of waiting for the bisection's own midpoints to bracket it.  (hand_over, i.e. the device kernels' form: a
        // wavefront's lanes then do their closing in together, once, and their steps are answered from what it left; found
        // step by step every lane reaches it at another step number and the wavefront runs the loop below up to five times.)
        if (find_root && hand_over && step == 0 && mono && !refined && !(far_d < HUGE_VAL)) {
            const double xf = f->falling ? f->lo : f->hi, dd = (xf - f->orig) * dir;
            if (dd > 0.0 && xf >= eps && xf <= top) {
                const double gf = flow_pace(rgradient, xf, eps);
                evals++;
                note(dd, gf);
            }
        }
        // the root is bracketed by two solid facts: close in on it (false position, Illinois), each evaluation one more
        if (find_root && (step == 3 || step == 6 || step == 10 || step == 15 || step == 21 || (hand_over && step == 0)) && mono && !refined &&
            far_d < HUGE_VAL && near_g != 0.0) {
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
            // False position closes in on the root from ONE side as a rule: the other fact stays where it was, and every
            // midpoint between the two still costs the bisection an evaluation.  Bring both facts to within a few noise
            // widths of the root: steps of 4, 16, 64, ... times noise / slope away from the estimate, on either side, until
            // the gradient there is solid (each evaluation a fact like any other; hand_over, i.e. the device kernels' form).
            if (hand_over && far_d < HUGE_VAL && far_d > near_d) {
                const double slope = fabs(far_g - near_g) / (far_d - near_d);
                const double dr = fabs(far_g - near_g) > 0.0 ? near_d + (far_d - near_d) * fabs(near_g) / fabs(far_g - near_g) : 0.5 * (near_d + far_d);
                double       step0 = slope > 0.0 ? 4.0 * B.noise / slope : 0.0;
                if (!(step0 > 0.0) || !isfinite(step0)) step0 = 1e-3 * (far_d - near_d);
                for (int sgn = 1; sgn >= -1; sgn -= 2) {
                    double st = step0;
                    for (int k = 0; k < 6; k++, st *= 4.0) {
                        const double dp = dr + sgn * st;
                        if (!(dp > near_d && dp < far_d)) break;              // the fact on this side is that close already
                        const double xp = f->orig + dir * dp, dd = (xp - f->orig) * dir;
                        if (!(xp >= eps && xp <= top) || !(dd > near_d && dd < far_d)) break;
                        const double gp = flow_pace(rgradient, xp, eps);
                        evals++;
                        const double before = sgn > 0 ? far_d : near_d;
                        note(dd, gp);
                        if ((sgn > 0 ? far_d : near_d) != before) break;        // solid: this side is done
                    }
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
            // Which finish?  A flow whose root is bracketed and whose gradient falls off towards it faster than ~1 / step size
            // reaches the band, if at all, within the last per cent before the root, where the rule's value saturates: dozens
            // of cheap steps (a sign, a spared quadrature), the step-per-round kernels' case (6).  Else the guided bisection (2).
            // (The rule's last node sits 0.6 % of the interval before its end: with the root -- a pole of the integrand -- within a
            // per cent or two beyond the end, its value falls short of the integral, the estimates of the guided bisection with it.)
            if (mono && hand_over && far_d < HUGE_VAL &&
                ((far_d - d) < CNF2_SCOUT_NEAR_ROOT * far_d ||
                 (near_g != 0.0 && far_d > near_d && scalefactor * fabs(near_g) > CNF2_SCOUT_SATURATING * (far_d - near_d))))
                return 6;
            return mono ? 2 : 5;
        }
        f->spared++;
        flow_decide(f, 0.0, scalefactor);
    }
}

// ------------------------------------------------------------------ the guided bisection: the literal decisions from three quadratures
// A flow that reaches its step size runs ~10 quadratures in cappedgd: the bisection walks towards the point x* at which the
// rule reports the step size, and every midpoint on the way costs 16 gradient evaluations to learn "not there yet" or "too
// far".  Where the gradient is strictly monotone on the bracket (flow_interval: s1 > 0; the flow runs towards an attractor,
// |G| falls along its way) those verdicts are ordered.  Write f = 1 / |G| (positive and increasing along the way) and
// R(m) = rule[f] over [start, m]; the rule has positive weights w_i at relative positions 0 < c_i < 1, so
//     dR/dm = sum_i w_i d/dx[(x - start) f(x)] at the nodes > 0:      R is increasing in m.
// Hence ONE literal evaluation at a point p is a fact about every midpoint on one side of it:
//   * the rule reports less than the band [sf (1 - 1e-3), sf (1 + 1e-3)] at p  =>  it does at every midpoint before p
//     (and the gradient there has the start's sign: it lies between the start's and p's);
//   * the rule reports more than the band at p, or the gradient has turned round at p  =>  every midpoint beyond p is
//     "too far" (its gradient has turned round, or the rule reports at least what it reported at p: nodes beyond p see a
//     smaller |G| than p does; they still see the start's sign when the last node, 0.6 % of the interval before its end,
//     is clear of the root -- the same condition the time bound above asks for).
// Rounding: a computed gradient is the true one up to `noise` (flow_interval), so a computed 1 / G at a node is good to
// noise / |G| <= noise / |G(p)| relatively, and a fact only counts ("solid") when it stands clear of the band by that.
// The literal bisection is then run as before -- same midpoints, same order, same tests -- but a midpoint that a solid
// fact covers costs nothing, and the points that ARE evaluated are chosen where they decide most: an estimate of x*
// (Newton on the rule's own value: dR/dm = 1 / |G(m)| to leading order) is turned into the path the bisection will take,
// and the path's last three points are evaluated -- the midpoint predicted to land in the band and the two ends of the
// bracket it halves.  If they come out as predicted every other midpoint of the path is covered: 3 - 4 quadratures instead
// of ~10.  If they do not, nothing is lost but the evaluation: every result is a fact or (in the band, not solid) a memo of
// the point itself, the estimate is refined with it, and the bisection goes on; what is never done is take a decision that
// was not either evaluated literally or implied by such an evaluation.  Flows whose gradient is not (yet) known to be
// monotone take literal steps, and the test is repeated as their bracket shrinks.
enum { PT_NONE = 0, PT_NEAR = 1, PT_FAR = 2, PT_BAND = 3, PT_TINY = 4 };
struct FlowPoint {
    int    kind;         // what a step of cappedgd with this midpoint does: PT_NEAR "not there yet", PT_FAR "too far", PT_BAND tolerance met, PT_TINY interval under 1e-10
    bool   by_rule;      // t is the (finite) value the rule reported; false: the gradient's sign, a non-finite value or the 1e-10 test decided
    double t, pace;      // pace = 1 / gradient at the point, as computed
};
// one step's evaluation at the point p, in the order and with the arithmetic of flow_try / flow_quadrature
template <class G>
CNF2_UHD FlowPoint flow_point(const FlowState& f, G&& rgradient, double p, double scalefactor)
{
    FlowPoint r;
    r.by_rule = false;
    r.t = 0.0;
    const double gm = flow_pace(rgradient, p, f.epsilon);
    r.pace = gm;
    if (((gm < 0) != f.falling) || !isfinite(gm)) {
        r.kind = PT_FAR;
        return r;
    }
    double a = f.orig, b = p;
    if (a > b) {
        const double s = a;
        a = b;
        b = s;
    }
    if (b - a < 1e-10) {
        r.kind = PT_TINY;
        return r;
    }
    const double eps = f.epsilon;
    double       t = gauss15([&](double v) CNF2_LI { return flow_pace(rgradient, v, eps); }, a, b);
    if (b != p) t = -t;
    if (!isfinite(t)) t = (scalefactor + 0.1) * 1.1;
    else r.by_rule = true;
    r.t = t;
    r.kind = (fabs(t - scalefactor) < scalefactor * 1e-3) ? PT_BAND : (t < scalefactor ? PT_NEAR : PT_FAR);
    return r;
}

#define CNF2_GUIDE_MEMO 3
struct FlowGuide {
    double s1, noise;          // flow_interval over the bracket and the start when the gradient is known to be monotone there
    bool   mono;
    bool   capped;             // the whole bracket is known to be "not there yet" (flow_guide_seed)
    double near_d, far_d;      // solid facts, as distances from the start along the flow: every midpoint up to near_d is "not there yet", every one from far_d on "too far"
    // "too far" by the rule's value at a point holds for the midpoints beyond it whose nodes all see the start's sign: farc_d
    // is such a fact (the rule reported farc_t there) that could not show this from the slope; it counts up to clear_d, the
    // farthest point at which the gradient has been seen to keep that sign solidly (clear_f = |G| there).  want_clear: ask for
    // the gradient at the bracket's far end (one evaluation) to set clear_d
    double farc_d, farc_t, clear_d, clear_f;
    bool   want_clear;
    double memo_x[CNF2_GUIDE_MEMO];   // literal results that hold for the point itself only (in the band, under 1e-10, not solid)
    int    memo_kind[CNF2_GUIDE_MEMO];
    // the estimate of the distance at which the rule reports the step size: the anchor (a point whose rule value is known:
    // the start, where it is 0, or the evaluation that came closest to the step size) and a linear model of the gradient
    // along the way, |G|(d) = anchor_f - slope (d - anchor_d), under which the rule's value has a closed form
    double anchor_d, anchor_f, anchor_t, slope;
    double last_d, last_f;     // the evaluation before, for the slope (signed: negative once the gradient has turned round)
    double best_dt;            // |step size - rule| at the anchor
    bool   mono_tried;         // ... at the bracket the flow was taken up with
    // the plan in force (flow_guide_plan under the estimate of the time it was made): the points to evaluate, and whether
    // the evaluations so far came out as it expects (else it is made again, from the better estimate they left)
    double plan_term, plan_near, plan_far;
    bool   plan_has_term, plan_has_near, plan_has_far, plan_valid;
    int    spec_left;          // evaluations away from the bisection's own midpoint that the flow may still ask for
    int    evals, points;      // gradient evaluations, literal points (diagnostics)
};
#define CNF2_GUIDE_SPECULATIVE 8
CNF2_UHD void flow_guide_begin(FlowGuide* g)
{
    g->s1 = 0.0;
    g->noise = HUGE_VAL;
    g->mono = false;
    g->mono_tried = false;
    g->plan_term = g->plan_near = g->plan_far = 0.0;
    g->plan_has_term = g->plan_has_near = g->plan_has_far = g->plan_valid = false;
    g->capped = false;
    g->near_d = 0.0;
    g->far_d = HUGE_VAL;
    g->farc_d = HUGE_VAL;
    g->farc_t = 0.0;
    g->clear_d = g->clear_f = 0.0;
    g->want_clear = false;
    for (int i = 0; i < CNF2_GUIDE_MEMO; i++) {
        g->memo_x[i] = -1.0;
        g->memo_kind[i] = PT_NONE;
    }
    g->anchor_d = g->anchor_f = g->anchor_t = g->slope = 0.0;
    g->last_d = g->last_f = 0.0;
    g->best_dt = HUGE_VAL;
    g->spec_left = CNF2_GUIDE_SPECULATIVE;
    g->evals = g->points = 0;
}
// is the gradient monotone on the bracket (and from the start to it)?  Tried when the flow is taken up and at fixed step
// numbers while the bracket shrinks
CNF2_UHD bool flow_guide_mono_due(int it) { return it == 0 || it == 1 || it == 2 || it == 4 || it == 7 || it == 11 || it == 16 || it == 22 || it == 29 || it == 37; }
CNF2_UHD void flow_guide_try_mono(const FlowState& f, FlowGuide* g, const SlopeTerms& st)
{
    const double eps = f.epsilon, top = 1.0 - f.epsilon;
    if (g->mono || !(f.lo >= eps && f.hi <= top) || !isfinite(f.g0) || f.g0 == 0.0) return;
    const double        xa = f.lo < f.orig ? f.lo : f.orig, xb = f.hi > f.orig ? f.hi : f.orig;
    const IntervalFacts B = flow_interval(st, xa, xb, true);
    if (B.s1 > 0.0 && B.noise < HUGE_VAL) {
        g->mono = true;
        g->s1 = B.s1;
        g->noise = B.noise;
    }
}
CNF2_UHD double flow_distance(const FlowState& f, double x) { return f.falling ? f.orig - x : x - f.orig; }
// what is known about a midpoint without evaluating it
CNF2_UHD int flow_guide_known(const FlowState& f, const FlowGuide& g, double mid)
{
    const double d = flow_distance(f, mid);
    if (g.mono && d > 0.0) {
        if (d >= g.far_d || (d >= g.farc_d && d <= g.clear_d)) return PT_FAR;
        if (d <= g.near_d) return d < 1e-10 ? PT_TINY : PT_NEAR;       // the 1e-10 test on sorted ends: b - a = |mid - start|
    }
    int kind = PT_NONE;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = CNF2_GUIDE_MEMO - 1; i >= 0; i--)
        if (g.memo_kind[i] != PT_NONE && g.memo_x[i] == mid) kind = g.memo_kind[i];
    return kind;
}
// the result of a literal evaluation at p: a fact, a memo, a better estimate
CNF2_UHD void flow_guide_feed(const FlowState& f, FlowGuide* g, double p, const FlowPoint& r, double scalefactor)
{
    const double d = flow_distance(f, p);
    g->points++;
    g->evals += r.by_rule ? 16 : 1;
    bool solid = false;
    const bool usable = isfinite(r.pace) && r.pace != 0.0;
    const double gp = usable ? 1.0 / fabs(r.pace) : 0.0;             // |G(p)| as computed
    if (g->mono && d > 0.0 && usable) {
        const double slack = 4.0 * g->noise / gp + 1e-12;
        if (r.kind == PT_NEAR && r.by_rule && gp > 4.0 * g->noise && r.t * (1.0 + slack) < scalefactor * (1.0 - 1e-3)) {
            solid = true;
            if (d > g->near_d) g->near_d = d;
        } else if (r.kind == PT_FAR && !r.by_rule && ((r.pace < 0) != f.falling) && gp > 2.0 * g->noise) {
            solid = true;                                            // the gradient has turned round, by more than its noise
            if (d < g->far_d) g->far_d = d;
        } else if (r.kind == PT_FAR && r.by_rule && gp > 4.0 * g->noise && r.t * (1.0 - slack) > scalefactor * (1.0 + 1e-3)) {
            if (0.006 * d * g->s1 > 8.0 * g->noise) {
                solid = true;
                if (d < g->far_d) g->far_d = d;
            } else if (d < g->farc_d) {
                g->farc_d = d;
                g->farc_t = r.t;
                if (g->clear_d > d && !(r.t * (1.0 - 4.0 * g->noise / g->clear_f - 1e-12) > scalefactor * (1.0 + 1e-3))) g->farc_d = HUGE_VAL;
                else if (!(g->clear_d > d)) g->want_clear = true;
                solid = g->farc_d == d && g->clear_d > d;
            }
        }
        // a gradient that solidly keeps the start's sign: no root up to here
        if (((r.pace < 0) == f.falling) && gp > 4.0 * g->noise && d > g->clear_d) {
            g->clear_d = d;
            g->clear_f = gp;
        }
    }
    if (g->plan_valid && ((g->plan_has_term && p == g->plan_term && r.kind != PT_BAND) || (g->plan_has_near && p == g->plan_near && r.kind != PT_NEAR) ||
                          (g->plan_has_far && p == g->plan_far && r.kind != PT_FAR)))
        g->plan_valid = false;
    // an evaluation that leaves neither a solid fact nor the band's point (inside the gradient's noise next to a root, or the
    // rule's value next to the band): from here on the flow takes the bisection's own steps, each of which is progress
    if (!solid && r.kind != PT_BAND && g->mono) g->spec_left = 0;
    if (!solid) {                                                    // newest first (constant indices: the memo stays in registers)
        g->memo_x[2] = g->memo_x[1];
        g->memo_kind[2] = g->memo_kind[1];
        g->memo_x[1] = g->memo_x[0];
        g->memo_kind[1] = g->memo_kind[0];
        g->memo_x[0] = p;
        g->memo_kind[0] = r.kind;
    }
    if (usable && d > 0.0 && d != g->last_d) {
        const double fs = ((r.pace < 0) != f.falling) ? -gp : gp;     // along the flow: negative beyond the root
        g->slope = (g->last_f - fs) / (d - g->last_d);
        g->last_d = d;
        g->last_f = fs;
        if (r.by_rule) {
            const double dt = fabs(scalefactor - r.t);
            if (dt < g->best_dt) {
                g->best_dt = dt;
                g->anchor_d = d;
                g->anchor_f = gp;
                g->anchor_t = r.t;
            }
        }
    }
}
// the gradient at the bracket's far end (asked for by want_clear), 1 / G as computed
CNF2_UHD void flow_guide_feed_clear(const FlowState& f, FlowGuide* g, double p, double pace)
{
    const double d = flow_distance(f, p);
    g->want_clear = false;
    g->evals++;
    if (!g->mono || !isfinite(pace) || pace == 0.0 || !(d > 0.0)) {
        g->farc_d = HUGE_VAL;
        return;
    }
    const double gp = 1.0 / fabs(pace);
    if (((pace < 0) != f.falling)) {
        if (gp > 2.0 * g->noise && d < g->far_d) g->far_d = d;        // turned round, solidly: a fact of its own
        g->farc_d = HUGE_VAL;
        return;
    }
    if (gp > 4.0 * g->noise && d > g->clear_d) {
        g->clear_d = d;
        g->clear_f = gp;
    }
    if (!(g->clear_d >= d) || !(g->farc_t * (1.0 - 4.0 * g->noise / g->clear_f - 1e-12) > 0.0)) g->farc_d = HUGE_VAL;
}
// the model's answer: where the rule reports the step size (*ds) and |G| there (*fs).  With |G| linear from the anchor the
// time from it to d is -log(1 - slope (d - anchor_d) / anchor_f) / slope.
CNF2_UHD void flow_guide_estimate(const FlowGuide& g, double scalefactor, double* ds, double* fs)
{
    const double rem = scalefactor - g.anchor_t, z = g.slope * rem;
    double       step;
    if (fabs(z) < 1e-4) step = rem * g.anchor_f * (1.0 - 0.5 * z);
    else step = (g.anchor_f / g.slope) * (1.0 - exp(-z));
    if (!isfinite(step)) step = rem * g.anchor_f;
    *ds = g.anchor_d + step;
    const double fl = g.anchor_f - g.slope * step;
    *fs = fl > 0.0 ? fl : g.anchor_f;
}
// the start and one gradient evaluation at the distance an Euler step would reach seed the model (slope from the two)
template <class G>
CNF2_UHD void flow_guide_seed(const FlowState& f, FlowGuide* g, G&& rgradient, const SlopeTerms& st, double scalefactor, bool sharpen = true)
{
    const double f0 = 1.0 / fabs(f.g0);
    g->anchor_d = 0.0;
    g->anchor_f = f0;
    g->anchor_t = 0.0;
    g->last_d = 0.0;
    g->last_f = f0;
    g->slope = 0.0;
    const double far_end = flow_distance(f, f.falling ? f.lo : f.hi);
    double       de = scalefactor * f0;
    if (!(de < far_end)) de = far_end;
    const double x = f.falling ? f.orig - de : f.orig + de;
    const double d = flow_distance(f, x);
    if (!(d > 0.0)) return;
    const double pe = flow_pace(rgradient, x, f.epsilon);
    g->evals++;
    if (!isfinite(pe) || pe == 0.0) return;
    const double fe = ((pe < 0) != f.falling) ? -1.0 / fabs(pe) : 1.0 / fabs(pe);
    g->slope = (f0 - fe) / d;
    g->last_d = d;
    g->last_f = fe;
    // Sharpen the estimate before anything is evaluated literally: the time to the model's answer by a 5-point rule (the
    // integrand is smooth this far from a root, and nothing depends on this value but where the first points are put),
    // the gradient there, and the model anchored on the two.  Six evaluations instead of the sixteen of a literal point.
    if (sharpen && g->mono && de < far_end) {
        double ds, fs;
        flow_guide_estimate(*g, scalefactor, &ds, &fs);
        if (ds > 0.0 && ds < far_end) {
            const double xs = f.falling ? f.orig - ds : f.orig + ds;
            const double z5[2] = {0.5384693101056831, 0.9061798459386640}, w5[3] = {0.5688888888888889, 0.4786286704993665, 0.2369268850561891};
            const double mc = 0.5 * (f.orig + xs), hl = 0.5 * ds;
            bool         ok = true;
            auto pace_at = [&](double xx) CNF2_LI {
                const double pv = flow_pace(rgradient, xx, f.epsilon);
                g->evals++;
                ok = ok && isfinite(pv) && pv != 0.0 && ((pv < 0) == f.falling);
                return fabs(pv);
            };
            double acc = w5[0] * pace_at(mc);
            for (int i = 0; i < 2; i++) acc += w5[i + 1] * (pace_at(mc + hl * z5[i]) + pace_at(mc - hl * z5[i]));
            const double ps = pace_at(xs);
            if (ok) {
                const double t5 = hl * acc, fsx = 1.0 / ps, dx = flow_distance(f, xs);
                if (dx > g->last_d || dx < g->last_d) g->slope = (g->last_f - fsx) / (dx - g->last_d);
                g->last_d = dx;
                g->last_f = fsx;
                g->anchor_d = dx;
                g->anchor_f = fsx;
                g->anchor_t = t5;
                g->best_dt = fabs(scalefactor - t5) + 1e-3 * scalefactor;      // any literal value near the band takes its place
            }
        }
    }
    // The flow runs into its cap (an Euler step from the start overshoots the bracket, so the evaluation above sits at the
    // bracket's far end)?  If |G| >= gmin > 0 on the whole bracket the rule reports at most |m - start| / gmin at any
    // midpoint m: with that under the band every step is "not there yet" and nothing is left to evaluate.  gmin: where the
    // gradient is known to be monotone, its value at the far end; else an enclosure of the gradient on the bracket.
    if (de == far_end && x >= f.epsilon && x <= 1.0 - f.epsilon && f.orig >= f.epsilon && f.orig <= 1.0 - f.epsilon) {
        double gmin = 0.0;
        if (g->mono) {
            if (fe > 4.0 * g->noise) gmin = fe * (1.0 - 4.0 * g->noise / fe - 1e-12);
        } else {
            const double        xa = x < f.orig ? x : f.orig, xb = x < f.orig ? f.orig : x;
            const GradientRange R = flow_gradient_range(st, xa, xb, logit(xa), logit(xb));
            gmin = f.falling ? -R.ghi : R.glo;
        }
        if (gmin > 0.0 && d * (1.0 + 1e-12) < scalefactor * (1.0 - 1e-3) * gmin) {
            g->mono = true;                  // what the facts need of it holds: the sign is the start's on the whole bracket
            if (!(g->noise < HUGE_VAL)) g->noise = 0.0;
            g->capped = true;
            g->near_d = d;
        }
    }
}
// the path the bisection takes from the flow's present bracket if the rule reports the band exactly between the distances
// band_lo and band_hi: its last midpoint (0: the bisection ends without one, on its step count or its bounds) and the two
// ends of the bracket at that time where they are midpoints of the path
#ifndef CNF2_PLAN_DEPTH
#define CNF2_PLAN_DEPTH 24
#endif
struct FlowPlan {
    double term_x, near_x, far_x;
    bool   has_term, has_near, has_far;
};
CNF2_UHD void flow_guide_plan(const FlowState& f, double band_lo, double band_hi, FlowPlan* P)
{
    double lo = f.lo, hi = f.hi;
    P->has_term = P->has_near = P->has_far = false;
    P->term_x = P->near_x = P->far_x = 0.0;
    // at most CNF2_PLAN_DEPTH steps ahead (a wavefront's lanes plan together and wait for the longest; a flow whose band lies
    // deeper gets the ends its bracket has by then, evaluates them and plans again from the bracket they leave).  The
    // bisection's own stops (its bounds, cnF2freq.cpp:4131) are left to the bisection: a planned point beyond them is a
    // fact about the midpoints before it all the same.
    int steps = 51 - f.it;
    steps = steps > CNF2_PLAN_DEPTH ? CNF2_PLAN_DEPTH : steps;
    for (int k = 0; k < steps; k++) {
        const double mid = (lo + hi) / 2;
        const double d = flow_distance(f, mid);
        if (d >= band_hi) {
            P->far_x = mid;
            P->has_far = true;
            if (f.falling) lo = mid;
            else hi = mid;
        } else if (d < 1e-10 || d > band_lo) {
            P->term_x = mid;
            P->has_term = true;
            return;
        } else {
            P->near_x = mid;
            P->has_near = true;
            if (f.falling) hi = mid;
            else lo = mid;
        }
    }
}
// The next thing to do: 0 = the flow has ended; 2 = evaluate *p literally (flow_point) and hand the result to flow_guide_feed;
// 3 = evaluate the gradient alone at *p (flow_pace) and hand 1 / G to flow_guide_feed_clear.
// Midpoints that facts cover are decided here, in the literal order, with flow_decide.
// arbiter_only: decide what the facts decide and stop at the first midpoint they do not (2, *p = that midpoint) without choosing a
// better point to evaluate (the lock-step kernel that evaluates a whole plan's points in a row asks this way at its end)
CNF2_UHD int flow_guide_next(FlowState* f, FlowGuide* g, const SlopeTerms& st, double scalefactor, double* p, bool arbiter_only = false)
{
    for (;;) {
        if (!f->live) return 0;
        if (f->it >= 51 || f->lo > f->hilim || f->hi < f->lolim) {
            f->why = 3;
            f->live = false;
            return 0;
        }
        if (!g->mono && (!g->mono_tried || flow_guide_mono_due(f->it))) flow_guide_try_mono(*f, g, st);
        g->mono_tried = true;
        const double mid = (f->lo + f->hi) / 2;
        if (g->want_clear) {
            *p = f->falling ? f->lo : f->hi;
            return 3;
        }
        const int    kind = flow_guide_known(*f, *g, mid);
        if (kind == PT_NONE) {
            *p = mid;
            if (!g->mono || g->spec_left <= 0 || arbiter_only) return 2;   // literal steps until the gradient is known to be monotone
            g->spec_left--;
            const double far_end = flow_distance(*f, f->falling ? f->lo : f->hi);
            const double open_hi = g->far_d < far_end ? g->far_d : far_end;
            // the estimate; if facts contradict it: halve what is open.  (An estimate beyond the bracket's far end is no
            // contradiction: the flow runs into its cap, every midpoint is "not there yet", the far end is the point to evaluate.)
            auto estimate = [&](double* ds, double* fs) CNF2_LI {
                flow_guide_estimate(*g, scalefactor, ds, fs);
                if (!(*ds > g->near_d && *ds < g->far_d)) *ds = 0.5 * (g->near_d + open_hi);
            };
            if (!(g->best_dt < HUGE_VAL)) {
                // no value of the rule yet: the estimate comes from two gradients (flow_guide_seed) and is good to a few per
                // cent; evaluate the rule right there (any point is a fact) and let its value make the estimate one to plan with
                double ds, fs;
                estimate(&ds, &fs);
                const double de = ds < far_end ? ds : far_end;
                *p = f->falling ? f->orig - de : f->orig + de;
                if (flow_guide_known(*f, *g, *p) != PT_NONE || !(flow_distance(*f, *p) > 0.0)) *p = mid;
                return 2;
            }
            for (int attempt = 0; attempt < 2; attempt++) {
                if (!g->plan_valid) {
                    double ds, fs;
                    estimate(&ds, &fs);
                    const double half = 1e-3 * scalefactor * fs;
                    FlowPlan     P;
                    flow_guide_plan(*f, ds - half, ds + half, &P);
                    g->plan_term = P.term_x;
                    g->plan_near = P.near_x;
                    g->plan_far = P.far_x;
                    g->plan_has_term = P.has_term;
                    g->plan_has_near = P.has_near;
                    g->plan_has_far = P.has_far;
                    g->plan_valid = true;
                    attempt = 1;
                }
                if (g->plan_has_term && flow_guide_known(*f, *g, g->plan_term) == PT_NONE) {
                    *p = g->plan_term;
                    return 2;
                }
                if (g->plan_has_near && flow_guide_known(*f, *g, g->plan_near) == PT_NONE) {
                    *p = g->plan_near;
                    return 2;
                }
                if (g->plan_has_far && flow_guide_known(*f, *g, g->plan_far) == PT_NONE) {
                    *p = g->plan_far;
                    return 2;
                }
                g->plan_valid = false;            // all its points are known and the midpoint still is not: the plan is out of date
            }
            return 2;                             // (*p == mid)
        }
        f->it++;
        f->mid = mid;
        if (kind == PT_BAND) {
            f->why = 1;
            f->live = false;
            return 0;
        }
        if (kind == PT_TINY) {
            f->why = 2;
            f->live = false;
            return 0;
        }
        flow_decide(f, kind == PT_NEAR ? 0.0 : (scalefactor + 0.1) * 1.1, scalefactor);
    }
}
// The points of one plan, to be evaluated in a row (the first lock-step kernel): the plan made from the present estimate; x[0..2] =
// its last midpoint and the two ends of the bracket it halves, use[k] = whether the point exists and nothing is known about it
CNF2_UHD void flow_guide_points(const FlowState& f, FlowGuide* g, double scalefactor, double x[3], bool use[3])
{
    const double far_end = flow_distance(f, f.falling ? f.lo : f.hi);
    double       ds, fs;
    flow_guide_estimate(*g, scalefactor, &ds, &fs);
    if (!(ds > g->near_d && ds < g->far_d)) ds = 0.5 * (g->near_d + (g->far_d < far_end ? g->far_d : far_end));
    const double half = 1e-3 * scalefactor * fs;
    FlowPlan     P;
    flow_guide_plan(f, ds - half, ds + half, &P);
    g->plan_term = x[0] = P.term_x;
    g->plan_near = x[1] = P.near_x;
    g->plan_far = x[2] = P.far_x;
    g->plan_has_term = P.has_term;
    g->plan_has_near = P.has_near;
    g->plan_has_far = P.has_far;
    g->plan_valid = true;
    use[0] = P.has_term && flow_guide_known(f, *g, P.term_x) == PT_NONE;
    use[1] = P.has_near && flow_guide_known(f, *g, P.near_x) == PT_NONE;
    use[2] = P.has_far && flow_guide_known(f, *g, P.far_x) == PT_NONE;
}
// a whole flow, the guided way (host tests; the device kernels run flow_guide_next / flow_point / flow_guide_feed per round)
template <class G>
CNF2_UHD double flow_step_guided(G&& rgradient, const SlopeTerms& st, double orig, double epsilon, double scalefactor, int* hits,
                                 bool breakathalf, FlowGuide* guide = nullptr, FlowState* state = nullptr)
{
    FlowState f;
    FlowGuide g;
    flow_begin(&f, rgradient, orig, epsilon, scalefactor, breakathalf);
    flow_guide_begin(&g);
    if (f.pinned) {
        while (flow_advance(&f, rgradient, scalefactor)) {}
    } else {
        double p;
        flow_guide_try_mono(f, &g, st);
        g.mono_tried = true;
        flow_guide_seed(f, &g, rgradient, st, scalefactor);
        for (int rc; (rc = flow_guide_next(&f, &g, st, scalefactor, &p)) != 0;) {
            if (rc == 3) flow_guide_feed_clear(f, &g, p, flow_pace(rgradient, p, f.epsilon));
            else flow_guide_feed(f, &g, p, flow_point(f, rgradient, p, scalefactor), scalefactor);
        }
    }
    if (guide) *guide = g;
    if (state) *state = f;
    return flow_end(f, scalefactor, hits, breakathalf);
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
