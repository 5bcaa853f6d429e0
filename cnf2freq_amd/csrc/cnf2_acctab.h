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

// What a probe of the line needs (line_terms() for a KNOWN incoming allele with zero error odds, the GENOSPROBE /
// HOMOZYGOUS calls of cnF2freq.cpp:5519-5537): with zero odds every match term is just its base value, so the line's
// base[fp] and tr[fp][fg] come out of a few products.  The "other grandparent" factors of a probe are the emission's
// own (they do not see the incoming allele) and are not formed here.
struct ProbeTerms {
    double base[2];
    double tr[2][2];
};
CNF2_HD void probe_terms(const LineCfg& c, const Slot& par, const Slot& tr, int value, ProbeTerms* T)
{
    T->base[0] = T->base[1] = 0.0;
    T->tr[0][0] = T->tr[0][1] = T->tr[1][0] = T->tr[1][1] = 0.0;
    if (!(c.par & SLOT_PRESENT)) {                           // 1 + secondval with secondval = 0
        T->base[0]  = 1.0;
        T->tr[0][0] = 1.0;
        return;
    }
    if (c.par & SLOT_FOUNDER) {                              // the parent is the top of its line
#pragma unroll
        for (int f = 0; f < 2; f++) {
            T->base[f]  = 1.0;
            T->tr[f][0] = probe_base(value, f ? par.a1 : par.a0, f ? par.s1 : par.s0) * phase_weight(par, f ^ c.firstpar ^ c.sp);
        }
        return;
    }
#pragma unroll
    for (int f = 0; f < 2; f++) {
        const double so = f ? par.s0 : par.s1;
        double b = probe_base(value, f ? par.a1 : par.a0, f ? par.s1 : par.s0) * phase_weight(par, f ^ c.firstpar ^ c.sp);
        if (so != 0.0) b *= (1.0 - so);
        if (!(b != 0.0)) continue;                           // cnF2freq.cpp:1271: nothing below a zero base
        T->base[f] = b;
        if (c.tr & SLOT_PRESENT) {
#pragma unroll
            for (int fg = 0; fg < 2; fg++)
                T->tr[f][fg] = probe_base(value, fg ? tr.a1 : tr.a0, fg ? tr.s1 : tr.s0) * phase_weight(tr, fg ^ c.bit_tr);
        } else {
            T->tr[f][0] = 1.0;
        }
    }
}

// Factored form of acc_entry_paths.  A path (fp, fgt, fgo) of the line has the emission term
// base[fp] * ot[fp][fgo] * tr[fp][fgt]; its GENOS weight w_i = R_i lp_i / (R_0 lp_0 + R_1 lp_1) does not depend on the
// other grandparent (the probe's "other" factor is the emission's own and cancels), so every sum over paths splits
// into a sum over the other grandparent's alleles times a 2 x 2 sum over (fp, fgt).  Where a weight is 0 / 0 (no probe
// is possible on a path the emission allows) the reference adds NaN to the HOMOZYGOUS sums but nothing to infprobs
// (its "non-zero product" tests fail there): kept.  Written without run-time array indices (everything is selected),
// so that the device version keeps its state in registers.
CNF2_HD void acc_entry(const Window& w, const Slot slot[7], int entry, int combo, bool no_ties, const AccRoot& ar,
                       double out[AK_COUNT])
{
#pragma unroll
    for (int k = 0; k < AK_COUNT; k++) out[k] = 0.0;
    const int  P = entry >> 5;
    const bool root_attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    LaneJob    L;
    make_lane(w, entry, &L);
    const LineCfg& c = L.cfg;
    const Slot  par = P ? slot[4] : slot[1], gpa = P ? slot[5] : slot[2], gpb = P ? slot[6] : slot[3];
    const Slot  tr = c.firstpar ? gpb : gpa, ot = c.firstpar ? gpa : gpb;
    ProbeTerms  TP0, TP1;
    probe_terms(c, par, tr, 1, &TP0);
    probe_terms(c, par, tr, 2, &TP1);
    if (root_attop) {
        out[AK_R] = 1.0;
        const double den = ar.Rs[0][0] + ar.Rs[0][1];
        // the all-zero path of a probe: base * other grandparent * traced grandparent.  The other grandparent's term
        // (allele index 0 of both) does not depend on the incoming allele: formed once, as line_terms() forms it
        double o00 = 1.0;
        if ((c.par & SLOT_PRESENT) && !(c.par & SLOT_FOUNDER)) {
            const double so = par.s1;
            const double ssv = (so != 0.0) ? so / (1.0 - so) : 0.0;
            if (c.ot & SLOT_PRESENT) {
                double o[2], two[2];
                top_terms(ot, par.a1, ssv, c.bit_ot, o, two);
                o00 = o[0];
            } else o00 = 1.0 + ssv;
        }
        const double lp0 = (TP0.base[0] * o00) * TP0.tr[0][0], lp1 = (TP1.base[0] * o00) * TP1.tr[0][0];
        out[AK_HZ + 0] = P ? lp0 : (ar.Rs[0][0] * ar.Xo[0]) * lp0 / den;
        out[AK_HZ + 1] = P ? lp1 : (ar.Rs[0][1] * ar.Xo[1]) * lp1 / den;
        return;
    }
    LineTerms T;
    line_terms(c, par, tr, ot, P ? ar.R.inmv1 : ar.R.inmv0, P ? ar.R.sv1 : ar.R.sv0, P == 0 && ar.R.inmv0 == 2, &T);
    const int  force_par = no_ties ? -1 : tie_force(L.tie_par, combo), force_tr = no_ties ? -1 : tie_force(L.tie_tr, combo),
               force_ot = no_ties ? -1 : tie_force(L.tie_ot, combo);
    const bool par_present = (c.par & SLOT_PRESENT) != 0, par_founder = (c.par & SLOT_FOUNDER) != 0;
    const bool par_is_line = !par_present || par_founder;
    const bool tr_real = !par_is_line && (c.tr & SLOT_PRESENT), ot_real = !par_is_line && (c.ot & SLOT_PRESENT);
    const double R0 = P ? ar.Rs[1][0] : ar.Rs[0][0], R1 = P ? ar.Rs[1][1] : ar.Rs[0][1];
    double rtot = 0.0, wh0 = 0.0, wh1 = 0.0, wroot0 = 0.0, wroot1 = 0.0, h0 = 0.0, h1 = 0.0;
    double wpar[2][2] = {{0, 0}, {0, 0}}, wgp[2][2] = {{0, 0}, {0, 0}};        // [allele index][i]
    double rpar0 = 0.0, rpar1 = 0.0, rtr0 = 0.0, rtr1 = 0.0, rot0 = 0.0, rot1 = 0.0;   // by forced phase psi
#pragma unroll
    for (int fp = 0; fp < 2; fp++) {
        const bool okp = !par_present || allele_ok(c.par, fp, c.firstpar, force_par);
        if (!okp) continue;
        const bool oko0 = ot_real ? allele_ok(c.ot, 0, c.bit_ot, force_ot) : true;
        const bool oko1 = ot_real ? allele_ok(c.ot, 1, c.bit_ot, force_ot) : false;
        const double O = (oko0 ? T.ot[fp][0] : 0.0) + (oko1 ? T.ot[fp][1] : 0.0);
        const double bo = T.base[fp] * O;
        double       trsum = 0.0, th0 = 0.0, th1 = 0.0;
#pragma unroll
        for (int fg = 0; fg < 2; fg++) {
            const bool okt = tr_real ? allele_ok(c.tr, fg, c.bit_tr, force_tr) : (fg == 0);
            if (!okt) continue;
            const double t = T.tr[fp][fg];
            trsum += t;
            if (t != 0.0) {
                th0 += TP0.tr[fp][fg];
                th1 += TP1.tr[fp][fg];
            }
            const double term = bo * t;                    // summed over the other grandparent's admissible alleles
            if (((fg ^ c.bit_tr) & 1) == 0) rtr0 += term;
            else rtr1 += term;
            if (term == 0.0) continue;                     // no path with a positive val here: never evaluated
            const double l0 = TP0.base[fp] * TP0.tr[fp][fg], l1 = TP1.base[fp] * TP1.tr[fp][fg];
            const double den = R0 * l0 + R1 * l1;
            const double tw0 = term * ((R0 * l0) / den), tw1 = term * ((R1 * l1) / den);
            wh0 += tw0;
            wh1 += tw1;
            if (den != 0.0) {
                wroot0 += tw0;
                wroot1 += tw1;
                if (par_present) {
                    wpar[fp][0] += tw0;
                    wpar[fp][1] += tw1;
                }
                if (tr_real) {
                    wgp[fg][0] += tw0;
                    wgp[fg][1] += tw1;
                }
            }
        }
        rtot += bo * trsum;
        if (((fp ^ c.firstpar) & 1) == 0) rpar0 += bo * trsum;
        else rpar1 += bo * trsum;
        const double bt = T.base[fp] * trsum;
        if (oko0) {
            if ((c.bit_ot & 1) == 0) rot0 += bt * T.ot[fp][0];
            else rot1 += bt * T.ot[fp][0];
        }
        if (oko1) {
            if ((c.bit_ot & 1) == 1) rot0 += bt * T.ot[fp][1];
            else rot1 += bt * T.ot[fp][1];
        }
        if (T.base[fp] != 0.0) {
            // probe sums over the paths the emission allows: the other grandparent's factor is the emission's own
            h0 += TP0.base[fp] * (TP0.base[fp] != 0.0 ? O : 0.0) * th0;
            h1 += TP1.base[fp] * (TP1.base[fp] != 0.0 ? O : 0.0) * th1;
        }
    }
    out[AK_R] = rtot;
    if (par_present) {
        out[AK_HAP + 0] = c.sp ? rpar1 : rpar0;            // phase = psi ^ the parent's localshift
        out[AK_HAP + 1] = c.sp ? rpar0 : rpar1;
        if (!par_founder) {
            // traced / other grandparent -> grandparent index by the parent's firstpar bit
            const bool trp = (c.tr & SLOT_PRESENT) != 0, otp = (c.ot & SLOT_PRESENT) != 0;
            const double a0 = c.firstpar ? (otp ? rot0 : 0.0) : (trp ? rtr0 : 0.0), a1 = c.firstpar ? (otp ? rot1 : 0.0) : (trp ? rtr1 : 0.0);
            const double b0 = c.firstpar ? (trp ? rtr0 : 0.0) : (otp ? rot0 : 0.0), b1 = c.firstpar ? (trp ? rtr1 : 0.0) : (otp ? rot1 : 0.0);
            out[AK_HAP + 2] = a0;
            out[AK_HAP + 3] = a1;
            out[AK_HAP + 4] = b0;
            out[AK_HAP + 5] = b1;
        }
    }
    out[AK_WROOT + 0] = wroot0;
    out[AK_WROOT + 1] = wroot1;
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            out[AK_WPAR + x * 2 + i] = wpar[x][i];
            out[AK_WGP + (0 * 2 + x) * 2 + i] = c.firstpar ? 0.0 : wgp[x][i];
            out[AK_WGP + (1 * 2 + x) * 2 + i] = c.firstpar ? wgp[x][i] : 0.0;
        }
    out[AK_HZ + 0] = P ? h0 : wh0;
    out[AK_HZ + 1] = P ? h1 : wh1;
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
