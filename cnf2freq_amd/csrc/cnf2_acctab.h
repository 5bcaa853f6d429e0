// cnf2_acctab.h -- table form of the HOT LOOP 2 accumulators (cnF2freq.cpp:5416-5577: updatehaplo / HAPLOS,
// GENOSPROBE + GENOS = infprobs, HOMOZYGOUS = homozyg) for ONE (individual, marker), all shift modes and states
// at once.  Host + device; the algebra is unit-tested on the CPU against the per-(state, mode) closed form of
// cnf2_accum.h, which in turn is pinned on the oracle's 128-path fan-out (tests/test_host_emission.py).
//
// Every accumulator of the reference is a sum over (shift mode s, state g, path) of
//     val = wg(s, g) * c_f(s0) * TA(path in line 0) * TB(path in line 1)          (rank-2 emission, cnf2_emission.h)
// times a weight that depends on ONE line's path only (cnf2_accum.h).  With the per-line sums tabulated per table
// entry e = (P, f, sp, k) -- the same 64 entries the sweep's emission table has -- an accumulator becomes
//     sum_f sum_{e0, e1} W_f[e0][e1] * X_f[e0] * Y_f[e1],    W_f[(s1,a)][(s2,b)] = sum_s0 c_f(s0) wg(s0,s1,s2,a,b)
// and all but one of the pairs (X, Y) have the plain restricted total R on one side.  So two partial contractions
//     v_{f,s0}[e0] = sum_{e1} c_f(s0) wg R1_f[e1]        u_f[e1] = sum_{s0,e0} c_f(s0) wg R0_f[e0]
// (and z_{f,i}[e0] with the HOMOZYGOUS probe sums of line 1 in place of R1) serve every accumulator with a 16-term
// dot product.  wg = exp(scales - factor) alphaminus_s(g) beta_s(g) is what the sweep kernels store per (job,
// marker) in the accumulate mode.
#ifndef CNF2_ACCTAB_H
#define CNF2_ACCTAB_H

#include "cnf2_accum.h"

namespace cnf2 {

// kinds of one table entry
enum {
    AK_R      = 0,    // restricted total of the line (= line_restricted)
    AK_HAP    = 1,    // [3 slots: parent, grandparent 0, grandparent 1][2 phases]: R split by the phase the slot is used with
    AK_WROOT  = 7,    // [i] GENOS reaches the root with a non-zero product
    AK_WPAR   = 9,    // [x][i] ... the parent's allele index x
    AK_WGP    = 13,   // [gp 0 / gp 1][x][i] ... the traced grandparent's allele index x (the other one's entries are 0)
    AK_HZ     = 21,   // [i] line 0: sum term * w_i; line 1: sum of the probe values (HOMOZYGOUS)
    AK_COUNT  = 23
};

// what does not depend on the lines: per root allele f
struct AccRoot {
    double cf[2];        // c_f(s0), s0 = 0, 1
    double pw;           // a non-zero root phase weight (only its being non-zero matters to the tables)
    double Rs[2][2];     // [side][i] probe base of the root for allele value i + 1
    double Xo[2];        // HOMOZYGOUS factor of the root's other allele (cnF2freq.cpp:1304-1318)
    double hzscale[2];   // Xo_i / d0 (root_attop: 1)
    bool   live;         // some c_f(s0) != 0
    RootTerms R;
};

CNF2_HD void acc_root(const Slot& root, bool root_attop, int f, AccRoot* A)
{
    root_terms(root, root_attop, f, &A->R);
    const double p0 = phase_weight(root, f ^ 0), p1 = phase_weight(root, f ^ 1);
    A->cf[0] = A->R.cbase * p0;
    A->cf[1] = A->R.cbase * p1;
    A->live  = (A->cf[0] != 0.0) || (A->cf[1] != 0.0);
    A->pw    = (A->cf[0] != 0.0) ? p0 : p1;
    const int    mf = f ? root.a1 : root.a0, mo = f ? root.a0 : root.a1;
    const double sf = f ? root.s1 : root.s0, so = f ? root.s0 : root.s1;
    const double d0 = (so != 0.0) ? 1.0 - so : 1.0;
    for (int i = 0; i < 2; i++) {
        A->Rs[0][i] = probe_base(i + 1, mf, sf);
        A->Rs[1][i] = probe_base(i + 1, mo, so);
        A->Xo[i]      = (i + 1 != mo) ? (mo != 0 ? so : 1.0) : (1.0 - so);
        A->hzscale[i] = root_attop ? 1.0 : A->Xo[i] / d0;
    }
}

// sums over the admissible paths of one line, split the ways the accumulators need (extends LineAcc of cnf2_accum.h)
struct LineAccH {
    LineAcc A;
    double  rpar[2], rtr[2], rot[2];   // R by the forced phase psi of the parent / traced / other grandparent
};

CNF2_HD void line_accumulate_h(const LineCfg& c, const LineTerms& T, const LineTerms TP[2], const double R[2],
                               double pwroot, const Slot& par, const Slot& tr, int force_par, int force_tr,
                               int force_ot, LineAccH* H)
{
    line_accumulate(c, T, TP, R, pwroot, par, tr, force_par, force_tr, force_ot, &H->A);
    H->rpar[0] = H->rpar[1] = H->rtr[0] = H->rtr[1] = H->rot[0] = H->rot[1] = 0.0;
    for (int fp = 0; fp < 2; fp++)
        for (int fgt = 0; fgt < 2; fgt++)
            for (int fgo = 0; fgo < 2; fgo++) {
                if (!line_path_ok(c, fp, fgt, fgo, force_par, force_tr, force_ot)) continue;
                const double term = line_path_term(T, fp, fgt, fgo);
                H->rpar[(fp ^ c.firstpar) & 1] += term;
                H->rtr[(fgt ^ c.bit_tr) & 1] += term;
                H->rot[(fgo ^ c.bit_ot) & 1] += term;
            }
}

// The AK_COUNT values of table entry `entry` = P<<5 | f<<4 | sp<<3 | k (cnf2_lane.h) for tie combination `combo`.
// slot[k]: data of window slot k at the marker (blank for missing slots).  ar = acc_root() of the entry's f.
// This form walks the (up to) 8 paths of the line like the reference does; acc_entry() below is the factored form
// the kernel uses, checked against this one on the host.
CNF2_HD void acc_entry_paths(const Window& w, const Slot slot[7], int entry, int combo, bool no_ties, const AccRoot& ar,
                             double out[AK_COUNT])
{
    for (int k = 0; k < AK_COUNT; k++) out[k] = 0.0;
    const int  P = entry >> 5;
    const bool root_attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    LaneJob    L;
    make_lane(w, entry, &L);
    const int   sp = 1 + 3 * P;
    const Slot &par = slot[sp], &tr = slot[sp + 1 + L.cfg.firstpar], &ot = slot[sp + 1 + (L.cfg.firstpar ^ 1)];
    LineTerms   TP[2];
    for (int i = 0; i < 2; i++) line_terms(L.cfg, par, tr, ot, i + 1, 0.0, false, &TP[i]);
    if (root_attop) {
        // the root is the top of its lines (cnF2freq.cpp:1120): no path sums; HOMOZYGOUS still recurses along the
        // all-zero path.  R = 1 so that the contractions reduce to sums of wg c_f.
        out[AK_R] = 1.0;
        const double den = ar.Rs[0][0] + ar.Rs[0][1];
        for (int i = 0; i < 2; i++) {
            const double lp = line_path_term(TP[i], 0, 0, 0);
            out[AK_HZ + i] = P ? lp : (ar.Rs[0][i] * ar.Xo[i]) * lp / den;
        }
        return;
    }
    LineTerms T;
    line_terms(L.cfg, par, tr, ot, P ? ar.R.inmv1 : ar.R.inmv0, P ? ar.R.sv1 : ar.R.sv0, P == 0 && ar.R.inmv0 == 2, &T);
    LineAccH H;
    line_accumulate_h(L.cfg, T, TP, ar.Rs[P], ar.pw, par, tr, no_ties ? -1 : tie_force(L.tie_par, combo),
                      no_ties ? -1 : tie_force(L.tie_tr, combo), no_ties ? -1 : tie_force(L.tie_ot, combo), &H);
    out[AK_R] = H.A.rtot;
    // HAPLOS: which slots the recursion reaches on this line (cnF2freq.cpp:1271, 1043-1046) and with which phase
    const LineCfg& c = L.cfg;
    if (c.par & SLOT_PRESENT) {
        for (int ph = 0; ph < 2; ph++) out[AK_HAP + 0 * 2 + ph] = H.rpar[ph ^ c.sp];      // the parent's localshift
        if (!(c.par & SLOT_FOUNDER)) {
            const int gt = c.firstpar, go = c.firstpar ^ 1;                               // grandparent index traced / other
            if (c.tr & SLOT_PRESENT)
                for (int ph = 0; ph < 2; ph++) out[AK_HAP + (1 + gt) * 2 + ph] = H.rtr[ph];
            if (c.ot & SLOT_PRESENT)
                for (int ph = 0; ph < 2; ph++) out[AK_HAP + (1 + go) * 2 + ph] = H.rot[ph];
        }
    }
    for (int i = 0; i < 2; i++) {
        out[AK_WROOT + i] = H.A.wroot[i];
        for (int x = 0; x < 2; x++) {
            out[AK_WPAR + x * 2 + i] = H.A.wpar[x][i];
            out[AK_WGP + (c.firstpar * 2 + x) * 2 + i] = H.A.wgp[x][i];
        }
        out[AK_HZ + i] = P ? H.A.h[i] : H.A.wh[i];
    }
}

// Factored form of acc_entry_paths.  A path (fp, fgt, fgo) of the line has the emission term
// base[fp] * ot[fp][fgo] * tr[fp][fgt]; its GENOS weight w_i = R_i lp_i / (R_0 lp_0 + R_1 lp_1) does not depend on the
// other grandparent (the probe's "other" factor is the emission's own and cancels), so every sum over paths splits
// into a sum over the other grandparent's alleles times a 2 x 2 sum over (fp, fgt).  Where a weight is 0 / 0 (no probe
// is possible on a path the emission allows) the reference adds NaN to the HOMOZYGOUS sums but nothing to infprobs
// (its "non-zero product" tests fail there): kept.
CNF2_HD void acc_entry(const Window& w, const Slot slot[7], int entry, int combo, bool no_ties, const AccRoot& ar,
                       double out[AK_COUNT])
{
    for (int k = 0; k < AK_COUNT; k++) out[k] = 0.0;
    const int  P = entry >> 5;
    const bool root_attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    LaneJob    L;
    make_lane(w, entry, &L);
    const LineCfg& c = L.cfg;
    const int   sp = 1 + 3 * P;
    const Slot &par = slot[sp], &tr = slot[sp + 1 + c.firstpar], &ot = slot[sp + 1 + (c.firstpar ^ 1)];
    LineTerms   TP[2];
    for (int i = 0; i < 2; i++) line_terms(c, par, tr, ot, i + 1, 0.0, false, &TP[i]);
    if (root_attop) {
        out[AK_R] = 1.0;
        const double den = ar.Rs[0][0] + ar.Rs[0][1];
        for (int i = 0; i < 2; i++) {
            const double lp = line_path_term(TP[i], 0, 0, 0);
            out[AK_HZ + i] = P ? lp : (ar.Rs[0][i] * ar.Xo[i]) * lp / den;
        }
        return;
    }
    LineTerms T;
    line_terms(c, par, tr, ot, P ? ar.R.inmv1 : ar.R.inmv0, P ? ar.R.sv1 : ar.R.sv0, P == 0 && ar.R.inmv0 == 2, &T);
    const int  force_par = no_ties ? -1 : tie_force(L.tie_par, combo), force_tr = no_ties ? -1 : tie_force(L.tie_tr, combo),
               force_ot = no_ties ? -1 : tie_force(L.tie_ot, combo);
    const bool par_present = (c.par & SLOT_PRESENT) != 0, par_founder = (c.par & SLOT_FOUNDER) != 0;
    const bool par_is_line = !par_present || par_founder;
    const bool tr_real = !par_is_line && (c.tr & SLOT_PRESENT), ot_real = !par_is_line && (c.ot & SLOT_PRESENT);
    bool okp[2], okt[2], oko[2];
    for (int a = 0; a < 2; a++) {
        okp[a] = !par_present || allele_ok(c.par, a, c.firstpar, force_par);
        okt[a] = tr_real ? allele_ok(c.tr, a, c.bit_tr, force_tr) : (a == 0);
        oko[a] = ot_real ? allele_ok(c.ot, a, c.bit_ot, force_ot) : (a == 0);
    }
    const double* R = ar.Rs[P];
    double rtot = 0.0, wh[2] = {0, 0}, wroot[2] = {0, 0}, wpar[2][2] = {{0, 0}, {0, 0}}, wgp[2][2] = {{0, 0}, {0, 0}};
    double h[2] = {0, 0}, rpar[2] = {0, 0}, rtr[2] = {0, 0}, rot[2] = {0, 0};
    for (int fp = 0; fp < 2; fp++) {
        if (!okp[fp]) continue;
        double O = 0.0, Oh[2] = {0, 0};
        for (int fg = 0; fg < 2; fg++)
            if (oko[fg]) {
                O += T.ot[fp][fg];
                if (T.ot[fp][fg] != 0.0) {
                    Oh[0] += TP[0].ot[fp][fg];
                    Oh[1] += TP[1].ot[fp][fg];
                }
            }
        const double bo = T.base[fp] * O;
        double       trsum = 0.0, th[2] = {0, 0};
        for (int fg = 0; fg < 2; fg++) {
            if (!okt[fg]) continue;
            const double t = T.tr[fp][fg];
            trsum += t;
            if (t != 0.0) {
                th[0] += TP[0].tr[fp][fg];
                th[1] += TP[1].tr[fp][fg];
            }
            const double term = bo * t;                    // sum over the other grandparent's admissible alleles
            rtr[(fg ^ c.bit_tr) & 1] += term;
            if (term == 0.0) continue;                     // no path with a positive val here: never evaluated
            const double l0 = TP[0].base[fp] * TP[0].tr[fp][fg], l1 = TP[1].base[fp] * TP[1].tr[fp][fg];
            const double den = R[0] * l0 + R[1] * l1;
            for (int i = 0; i < 2; i++) {
                const double wi = (R[i] * (i ? l1 : l0)) / den;
                const double tw = term * wi;
                wh[i] += tw;
                if (den != 0.0) {
                    wroot[i] += tw;
                    if (par_present) wpar[fp][i] += tw;
                    if (tr_real) wgp[fg][i] += tw;
                }
            }
        }
        rtot += bo * trsum;
        rpar[(fp ^ c.firstpar) & 1] += bo * trsum;
        for (int fg = 0; fg < 2; fg++)
            if (oko[fg]) rot[(fg ^ c.bit_ot) & 1] += (T.base[fp] * T.ot[fp][fg]) * trsum;
        if (T.base[fp] != 0.0)
            for (int i = 0; i < 2; i++) h[i] += TP[i].base[fp] * Oh[i] * th[i];
    }
    out[AK_R] = rtot;
    if (par_present) {
        for (int ph = 0; ph < 2; ph++) out[AK_HAP + 0 * 2 + ph] = rpar[ph ^ c.sp];
        if (!par_founder) {
            const int gt = c.firstpar, go = c.firstpar ^ 1;
            if (c.tr & SLOT_PRESENT)
                for (int ph = 0; ph < 2; ph++) out[AK_HAP + (1 + gt) * 2 + ph] = rtr[ph];
            if (c.ot & SLOT_PRESENT)
                for (int ph = 0; ph < 2; ph++) out[AK_HAP + (1 + go) * 2 + ph] = rot[ph];
        }
    }
    for (int i = 0; i < 2; i++) {
        out[AK_WROOT + i] = wroot[i];
        for (int x = 0; x < 2; x++) {
            out[AK_WPAR + x * 2 + i] = wpar[x][i];
            out[AK_WGP + (c.firstpar * 2 + x) * 2 + i] = wgp[x][i];
        }
        out[AK_HZ + i] = P ? h[i] : wh[i];
    }
}

// a * b where b carries the weight: a term whose weight is zero is never evaluated by the reference
// (cnF2freq.cpp:5502: val must be positive), so an undefined a must not leak through it
CNF2_HD double mul0(double a, double b) { return b == 0.0 ? 0.0 : a * b; }

// Scalar reference of the whole contraction for one (individual, marker): wg[s][g] -> inf[28], hz[2], hap[14] (added).
// The device kernel computes the same sums wave-parallel; this form is what the host tests check against
// accum_infprobs() and what documents the index bookkeeping.
CNF2_HD void acc_contract_scalar(const Window& w, const Slot slot[7], const double* wg /* [8][64] */, bool no_ties,
                                 double* inf, double* hz, double* hap)
{
    const bool root_attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    const int  n_combo = (no_ties || root_attop) ? 1 : (1 << w.n_groups);
    for (int f = 0; f < 2; f++) {
        AccRoot ar;
        acc_root(slot[0], root_attop, f, &ar);
        if (!ar.live) continue;
        for (int combo = 0; combo < n_combo; combo++) {
            double tab[2][16][AK_COUNT];                    // [P][sp*8 + k]
            for (int P = 0; P < 2; P++)
                for (int e = 0; e < 16; e++) acc_entry(w, slot, (P << 5) | (f << 4) | e, combo, no_ties, ar, tab[P][e]);
            double v[2][16], u[16], z[2][16];
            for (int e = 0; e < 16; e++) {
                v[0][e] = v[1][e] = u[e] = z[0][e] = z[1][e] = 0.0;
            }
            for (int s = 0; s < 8; s++) {
                const int    s0 = s & 1, e0s = ((s >> 1) & 1) << 3, e1s = ((s >> 2) & 1) << 3;
                const double cf = ar.cf[s0];
                if (cf == 0.0) continue;
                for (int g = 0; g < 64; g++) {
                    const double x = wg[s * 64 + g];
                    if (x == 0.0) continue;
                    const int e0 = e0s | (g & 7), e1 = e1s | (g >> 3);
                    v[s0][e0] += (cf * x) * tab[1][e1][AK_R];
                    u[e1] += (cf * x) * tab[0][e0][AK_R];
                    for (int i = 0; i < 2; i++) z[i][e0] += (cf * x) * tab[1][e1][AK_HZ + i];
                }
            }
            for (int e = 0; e < 16; e++) {
                // HAPLOS at the root: phase f ^ s0 (cnF2freq.cpp:1227 with firstpar = 0)
                for (int s0 = 0; s0 < 2; s0++) hap[0 * 2 + (f ^ s0)] += mul0(tab[0][e][AK_R], v[s0][e]);
                if (root_attop) {
                    const double vt = v[0][e] + v[1][e];
                    for (int i = 0; i < 2; i++) {
                        inf[(0 * 2 + f) * 2 + i] += vt * (ar.Rs[0][i] / (ar.Rs[0][0] + ar.Rs[0][1])) * (tab[0][e][AK_R]);
                        inf[(0 * 2 + (f ^ 1)) * 2 + i] += vt * (ar.Rs[1][i] / (ar.Rs[1][0] + ar.Rs[1][1])) * (tab[0][e][AK_R]);
                        hz[i] += mul0(tab[0][e][AK_HZ + i], z[i][e]);
                    }
                    continue;
                }
                const double vt = v[0][e] + v[1][e];
                for (int P = 0; P < 2; P++) {
                    const double* t = tab[P][e];
                    const double  wt = P ? u[e] : vt;                 // weight of this entry: everything of the other line
                    const int     sp = 1 + 3 * P;
                    for (int sl = 0; sl < 3; sl++)
                        for (int ph = 0; ph < 2; ph++) hap[(sp + sl) * 2 + ph] += mul0(t[AK_HAP + sl * 2 + ph], wt);
                    const int fr = P ? (f ^ 1) : f;
                    for (int i = 0; i < 2; i++) {
                        inf[(0 * 2 + fr) * 2 + i] += mul0(t[AK_WROOT + i], wt);
                        for (int x = 0; x < 2; x++) {
                            inf[(sp * 2 + x) * 2 + i] += mul0(t[AK_WPAR + x * 2 + i], wt);
                            for (int gpi = 0; gpi < 2; gpi++)
                                inf[((sp + 1 + gpi) * 2 + x) * 2 + i] += mul0(t[AK_WGP + (gpi * 2 + x) * 2 + i], wt);
                        }
                    }
                }
                for (int i = 0; i < 2; i++) hz[i] += ar.hzscale[i] * mul0(tab[0][e][AK_HZ + i], z[i][e]);
            }
        }
    }
}

} // namespace cnf2
#endif
