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

struct VMatch {
    double base, msv;
    int    mv;
};

// match term of trackpossible in NO_EQUIVALENCE mode (cnF2freq.cpp:303-316, 1191-1210): like match_term(), but an
// unknown incoming allele is not bound to the stored one (and so never turns the error odds into 1)
CNF2_HD VMatch variance_match(int inmv, double sv, int mf, double sf)
{
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

} // namespace cnf2
#endif
