// cnf2_variance.h -- closed form of individ::addvariance (cnF2freq.cpp:1489-1558) for one (individual, marker).
// Host + device; unit-tested on the CPU against the oracle's restatement of the reference's loops and on the GPU
// against the brute-force kernel (cnf2_addvariance) and the reference's own values (goldens G10, G12).
//
// The reference evaluates trackpossible<false, NO_EQUIVALENCE> for the individual's two alleles over shift modes 0-1,
// the 128 flags i = 2 g + firstpar and the admissible paths flag2 -- 65 536 calls per marker -- and squares, per
// (shift, firstpar, flag2 & 1), the signed sum over the two alleles.  In that mode nothing depends on the shift mode or
// on the state bits except through WHICH slots are visited (every level weighs 1/2 instead of a phase weight, only the
// traced line is followed below the root, cnF2freq.cpp:1229-1233, 1291), and a term is a product
//     root match * (line of the other parent) * (line of the traced parent),
// the two lines depending on disjoint bits of i and flag2.  So the sum over (i, flag2) of a class is a product of two
// small sums over each parent's own allele index, which grandparent it traces and that grandparent's allele index,
// times the number of don't-care bit patterns.
#ifndef CNF2_VARIANCE_H
#define CNF2_VARIANCE_H

#include "cnf2_emission.h"
#include "cnf2_window.h"

namespace cnf2 {

// operations as written, none contracted into an FMA (the exact evaluation below depends on it)
#if defined(__clang__)
#define CNF2_FP_LITERAL _Pragma("clang fp contract(off)")
#else
#define CNF2_FP_LITERAL                                     // g++ on x86-64 without -mfma has nothing to contract with
#endif

struct VMatch {
    double base, msv;
    int    mv;
};

// match term of trackpossible in NO_EQUIVALENCE mode (cnF2freq.cpp:303-316, 1191-1210): like match_term(), but an
// unknown incoming allele is not bound to the stored one (and so never turns the error odds into 1)
CNF2_HD VMatch variance_match(int inmv, double sv, int mf, double sf)
{
    CNF2_FP_LITERAL
    VMatch m;
    bool   miss;
    if (inmv == 0) {
        m.mv = 0;
        miss = false;
    } else {
        m.mv = inmv;
        miss = !(mf == 0 && inmv != 9) && (inmv != mf);
    }
    if (miss) {
        m.base = sf;
        m.msv  = (sf != 0.0 && sv != 0.0) ? (1.0 - sf) * sv : 0.0;
    } else {
        m.base = 1.0 - sf;
        m.msv  = (mf == 0 ? 1.0 : sf) * sv;
    }
    return m;
}

// Sum over the 8 state patterns and the admissible path patterns of parent P's subtree of the value of the recursion
// into that parent for incoming allele inmv with error odds sv (recursetrackpossible + trackpossible at genwidth 2 and 1).
CNF2_HD double variance_line(const Window& w, const Slot slot[7], int P, int inmv, double sv)
{
    const int    sp = 1 + 3 * P;
    const int    ig = (w.flag2ignore >> sp) & 7;          // bit 0 parent, 1 / 2 its parents: slot may only use index 0
    const double n_self = (ig & 1) ? 1.0 : 2.0, n_g0 = (ig & 2) ? 1.0 : 2.0, n_g1 = (ig & 4) ? 1.0 : 2.0;
    if (!(w.flags[sp] & SLOT_PRESENT)) return (1.0 + sv) * 8.0 * n_self * n_g0 * n_g1;      // cnF2freq.cpp:1043-1046
    const Slot& par = slot[sp];
    double      tot = 0.0;
    for (int fa = 0; fa < 2; fa++) {
        if ((ig & 1) && fa == 1) continue;
        const VMatch m = variance_match(inmv, sv, fa ? par.a1 : par.a0, fa ? par.s1 : par.s0);
        if (w.flags[sp] & SLOT_FOUNDER) {                   // top of the line (cnF2freq.cpp:1120, 1213-1217)
            tot += (m.base + m.msv) * 0.5 * 8.0 * n_g0 * n_g1;
            continue;
        }
        double msv = m.msv;
        if (msv != 0.0) msv /= m.base;
        const double base = m.base * 0.5;
        if (base == 0.0) continue;                          // cnF2freq.cpp:1271
        for (int fp = 0; fp < 2; fp++) {                    // which grandparent is traced (the parent's firstpar bit)
            const int    sg = sp + 1 + fp;
            const bool   masked = (ig & (2 << fp)) != 0;
            const double n_other = fp ? n_g0 : n_g1;
            double       g = 0.0;
            if (!(w.flags[sg] & SLOT_PRESENT)) g = (1.0 + msv) * (masked ? 1.0 : 2.0);
            else
                for (int fg = 0; fg < 2; fg++) {
                    if (masked && fg == 1) continue;
                    const VMatch t = variance_match(m.mv, msv, fg ? slot[sg].a1 : slot[sg].a0, fg ? slot[sg].s1 : slot[sg].s0);
                    g += (t.base + t.msv) * 0.5;            // genwidth 1: top of the line
                }
            tot += base * g * n_other * 4.0;                // 4 = the parent's two state bits besides firstpar
        }
    }
    return tot;
}

// variances[marker] as addvariance leaves it; *valid = false when every term is zero (the entry is then left alone)
CNF2_HD double variance_closed(const Window& w, const Slot slot[7], bool* valid)
{
    const Slot& root = slot[0];
    const bool  attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    double      npaths = 1.0;                               // admissible patterns of flag2 bits 1-6
    for (int k = 1; k < 7; k++) npaths *= ((w.flag2ignore >> k) & 1) ? 1.0 : 2.0;
    double sq = 0.0, sum = 0.0;
    for (int firstpar = 0; firstpar < 2; firstpar++)
        for (int f2n = 0; f2n < 2; f2n++) {
            if ((w.flag2ignore & 1) && f2n == 1) continue;
            const int    mf = f2n ? root.a1 : root.a0, mo = f2n ? root.a0 : root.a1;
            const double sf = f2n ? root.s1 : root.s0, so = f2n ? root.s0 : root.s1;
            double ok = 0.0, full = 0.0;
            for (int al = 0; al < 2; al++) {
                const VMatch m = variance_match(al ? root.a1 : root.a0, al ? root.s1 : root.s0, mf, sf);
                double term;
                if (attop) {
                    term = (m.base + m.msv) * 0.5 * 64.0 * npaths;
                } else {
                    double msv = m.msv;
                    if (msv != 0.0) msv /= m.base;
                    double base = m.base * 0.5, ssv = 0.0;
                    term = 0.0;
                    if (base != 0.0) {
                        if (so != 0.0) {                    // cnF2freq.cpp:1298-1302
                            base *= (1.0 - so);
                            ssv = so / (1.0 - so);
                        }
                        term = base * variance_line(w, slot, firstpar ^ 1, mo, ssv);
                        if (term != 0.0) term *= variance_line(w, slot, firstpar, m.mv, msv);
                    }
                }
                ok += al ? term : -term;
                full += term;
            }
            sq += ok * ok;
            sum += full;
        }
    *valid = sum != 0.0;
    return 2.0 * sq;                                        // the two shift modes give the same classes
}

// ---------------------------------------------------------------------------------------------------------------------
// The same number with the reference's OWN rounding: lockhaplos (cnF2freq.cpp:3045-3081) takes the first marker of
// strictly largest variance, and markers whose configurations are mirror images of each other (the alleles of two
// parents exchanged, ...) have variances that are equal in exact arithmetic and differ in the reference by the rounding
// of its sums -- which of them it locks is decided by those last bits.  So the markers that can win are evaluated once
// more the way addvariance adds: 8 classes (shift mode, i & 1, flag2 & 1), each a serial sum over i, flag2 and the two
// alleles of trackpossible's value, every value the product of the same factors in the same order
// (cnF2freq.cpp:1191-1340 in NO_EQUIVALENCE mode; no operation contracted).  What makes that affordable: a value depends
// on (i, flag2) only through which of 8 values each parent's line takes -- (firstpar of the parent, its allele index, the
// traced grandparent's allele index) -- so the 16 384 values of a class are products of three table entries and the sums
// are 32 768 additions in the reference's order.  The shift mode changes nothing in this mode: classes 4-7 repeat 0-3.
// Bit-equal to the reference's variances[] on goldens G10 / G12 (tests/test_host_emission.py on the CPU,
// tests/test_gpu_engine.py through cnf2_variances_exact).
// ---------------------------------------------------------------------------------------------------------------------

// trackpossible<false, NO_EQUIVALENCE> below the root: the recursion into parent P (genwidth 2) for incoming allele inmv
// with error odds sv, the parent using allele index f2n_p and tracing grandparent firstpar_p, who uses allele index f2n_g
CNF2_HD double variance_exact_line(const Window& w, const Slot slot[7], int P, int inmv, double sv, int firstpar_p, int f2n_p,
                                   int f2n_g)
{
    CNF2_FP_LITERAL
    const int sp = 1 + 3 * P;
    if (!(w.flags[sp] & SLOT_PRESENT)) return 1.0 + sv;                    // cnF2freq.cpp:1043-1046
    const bool   attop = (w.flags[sp] & SLOT_FOUNDER) != 0;                // cnF2freq.cpp:1120
    const Slot&  d = slot[sp];
    const VMatch m = variance_match(inmv, sv, f2n_p ? d.a1 : d.a0, f2n_p ? d.s1 : d.s0);
    double       baseval = m.base, msv = m.msv;
    if (attop) {                                                           // cnF2freq.cpp:1213-1221
        baseval += msv;
        msv = 0.0;
    } else if (msv != 0.0) msv /= baseval;
    baseval *= 0.5;                                                        // cnF2freq.cpp:1229-1233
    if (baseval == 0.0 || attop) return baseval;                           // cnF2freq.cpp:1271
    const int sg = sp + 1 + firstpar_p;                                    // only the traced line, cnF2freq.cpp:1291
    double    up;
    if (!(w.flags[sg] & SLOT_PRESENT)) up = 1.0 + msv;
    else {
        const Slot&  g = slot[sg];
        const VMatch t = variance_match(m.mv, msv, f2n_g ? g.a1 : g.a0, f2n_g ? g.s1 : g.s0);
        up = t.base;
        up += t.msv;                                                       // genwidth 1: top of the line
        up *= 0.5;
    }
    baseval *= up;                                                         // cnF2freq.cpp:1336-1340
    return baseval;
}

// One class of addvariance's loops (cnF2freq.cpp:1514-1541): i & 1 = firstpar, flag2 & 1 = f2n; *ok the signed sum over the
// two alleles (before fabs), *full the plain one
CNF2_HD void variance_exact_class(const Window& w, const Slot slot[7], int firstpar, int f2n, double* ok_out, double* full_out)
{
    CNF2_FP_LITERAL
    const Slot& root = slot[0];
    const bool  attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    const int    mf = f2n ? root.a1 : root.a0, mo = f2n ? root.a0 : root.a1;
    const double sf = f2n ? root.s1 : root.s0, so = f2n ? root.s0 : root.s1;
    double base[2], lo[8], lt[2][8];
    for (int al = 0; al < 2; al++) {
        const VMatch m = variance_match(al ? root.a1 : root.a0, al ? root.s1 : root.s0, mf, sf);
        double baseval = m.base, msv = m.msv;
        if (attop) {
            baseval += msv;
            msv = 0.0;
        } else if (msv != 0.0) msv /= baseval;
        baseval *= 0.5;
        if (baseval != 0.0 && !attop && so != 0.0) baseval *= (1.0 - so);  // cnF2freq.cpp:1298-1302
        base[al] = baseval;
        for (int k = 0; k < 8; k++)
            lt[al][k] = attop ? 1.0 : variance_exact_line(w, slot, firstpar, m.mv, msv, k & 1, (k >> 1) & 1, k >> 2);
    }
    const double ssv = so != 0.0 ? so / (1.0 - so) : 0.0;
    for (int k = 0; k < 8; k++) lo[k] = attop ? 1.0 : variance_exact_line(w, slot, firstpar ^ 1, mo, ssv, k & 1, (k >> 1) & 1, k >> 2);
    double ok = 0.0, full = 0.0;
    for (int i = firstpar; i < 128; i += 2) {
        const int fp0 = (i >> 1) & 1, fp1 = (i >> 4) & 1;                  // firstpar of parent 0 / 1 (upflagit, cnF2freq.cpp:321-329)
        for (int flag2 = f2n; flag2 < 128; flag2 += 2) {
            if (flag2 & w.flag2ignore) continue;
            const int b0 = (flag2 >> 1) & 7, b1 = (flag2 >> 4) & 7;
            const int k0 = fp0 | ((b0 & 1) << 1) | (((b0 >> 1 >> fp0) & 1) << 2);
            const int k1 = fp1 | ((b1 & 1) << 1) | (((b1 >> 1 >> fp1) & 1) << 2);
            const int ko = firstpar ? k0 : k1, kt = firstpar ? k1 : k0;
            for (int al = 0; al < 2; al++) {
                double term = base[al];
                if (term != 0.0 && !attop) {
                    term *= lo[ko];                                        // cnF2freq.cpp:1322
                    if (term != 0.0) term *= lt[al][kt];                   // cnF2freq.cpp:1336-1340
                }
                if (al) ok += term;
                else ok -= term;
                full += term;
            }
        }
    }
    *ok_out = ok;
    *full_out = full;
}

// addvariance's result from the four class sums (the classes of shift mode 1 repeat those of shift mode 0)
CNF2_HD double variance_exact_finish(const double ok[4], const double full[4], bool* valid)
{
    CNF2_FP_LITERAL
    double sum = 0.0, sqsum = 0.0;
    for (int s = 0; s < 2; s++)
        for (int k = 0; k < 4; k++) {
            const double a = fabs(ok[k]);
            sum += full[k];
            sqsum += a * a;
        }
    *valid = sum != 0.0;
    return sqsum;
}

CNF2_HD double variance_exact(const Window& w, const Slot slot[7], bool* valid)
{
    double ok[4], full[4];
    for (int k = 0; k < 4; k++) variance_exact_class(w, slot, k >> 1, k & 1, &ok[k], &full[k]);
    return variance_exact_finish(ok, full, valid);
}

} // namespace cnf2
#endif
