// cnf2_accum.h -- closed form of the infprobs / homozyg accumulators of HOT LOOP 2
// (cnF2freq.cpp:5513-5577, DOINFPROBS) for one (marker, state g, shift mode s); host + device, so that
// the algebra is unit-tested on the CPU against the oracle's brute-force fan-out (tests/test_host_emission.py).
//
// The reference visits every admissible path flag2 and, with val = exp(query - factor),
//   sideval[side][i] = trackpossible<GENOSPROBE>(allele i on the root's side `side`)          cpp:5519-5528
//   trackpossible<GENOS>(i, side) adds val * sideval[side][i] / sum_i sideval[side][.] to
//       infprobs[n][allele index][i] of every individual n on the traced line                  cpp:5560-5568, 1351-1354
//   homozyg[i] += val * trackpossible<HOMOZYGOUS>(i) / sum_i sideval[0][.]                    cpp:5531-5537, 5571-5575
// Structure used: val = K c_f TA(pathA) TB(pathB) (rank-2 emission, cnf2_emission.h).  A probe value is the
// same product with the traced line evaluated for incoming allele i and zero error odds:
//   sideval[0][i] = R0_i pw (1-so) TB(pathB) LP_0(i; pathA),  sideval[1][i] = R1_i pw (1-sf) TA'(pathA) LP_1(i; pathB)
// so the weight of a path depends on ONE line only:  w[side][i] = R_i LP_side(i; path) / sum_i' (same),
// and the sum over the other line's paths is its restricted total.  The GENOS recursion follows the traced
// line only (cnF2freq.cpp:1291) and adds where the product of its own match and phase terms is non-zero.
// HOMOZYGOUS hands allele i to both parents: its ratio to sum_i sideval[0][.] factorises as
// w[0][i](pathA) * Xo_i LP_1(i; pathB) / ((1-so) TB(pathB)), so TB cancels against val.
#ifndef CNF2_ACCUM_H
#define CNF2_ACCUM_H

#include "cnf2_lane.h"

namespace cnf2 {

// One allele assignment (path) of a line: parent allele fp, traced / other grandparent alleles.
CNF2_HD double line_path_term(const LineTerms& T, int fp, int fg_tr, int fg_ot)
{
    return (T.base[fp] * T.ot[fp][fg_ot]) * T.tr[fp][fg_tr];
}

// admissible under flag2ignore and the tie forces, exactly as line_restricted() counts it
CNF2_HD bool line_path_ok(const LineCfg& c, int fp, int fg_tr, int fg_ot, int force_par, int force_tr, int force_ot)
{
    const bool par_is_line = !(c.par & SLOT_PRESENT) || (c.par & SLOT_FOUNDER);
    if ((c.par & SLOT_PRESENT) && !allele_ok(c.par, fp, c.firstpar, force_par)) return false;
    const bool ok_ot = (par_is_line || !(c.ot & SLOT_PRESENT)) ? (fg_ot == 0) : allele_ok(c.ot, fg_ot, c.bit_ot, force_ot);
    const bool ok_tr = (par_is_line || !(c.tr & SLOT_PRESENT)) ? (fg_tr == 0) : allele_ok(c.tr, fg_tr, c.bit_tr, force_tr);
    return ok_ot && ok_tr;
}

// base value of markermiss + match for a KNOWN incoming allele i with zero odds (cnF2freq.cpp:1198-1210)
CNF2_HD double probe_base(int i, int mf, double sf)
{
    int mv;
    return markermiss(i, mf, &mv) ? sf : 1.0 - sf;
}

struct LineAcc {
    double rtot;          // sum of the emission terms over admissible paths (= line_restricted)
    double wh[2];         // sum term * w_i
    double wroot[2];      // sum term * w_i * [GENOS reaches a non-zero product at the root]
    double wpar[2][2];    // [fp][i]  ... at the parent
    double wgp[2][2];     // [fg][i]  ... at the traced grandparent
    double h[2];          // sum of the probe values LP(i; path)
};

// Sums over the admissible paths of ONE line.  T: emission terms of the line; TP[i-1]: the same line probed
// with allele i and zero odds; R[i-1]: root match base for allele i on this side; pwroot: the root's phase
// weight on these paths.  par/tr: data of the parent and of the traced grandparent at the marker.
CNF2_HD void line_accumulate(const LineCfg& c, const LineTerms& T, const LineTerms TP[2], const double R[2],
                             double pwroot, const Slot& par, const Slot& tr, int force_par, int force_tr,
                             int force_ot, LineAcc* A)
{
    A->rtot = 0.0;
    for (int i = 0; i < 2; i++) {
        A->wh[i] = A->wroot[i] = A->h[i] = 0.0;
        A->wpar[0][i] = A->wpar[1][i] = A->wgp[0][i] = A->wgp[1][i] = 0.0;
    }
    const bool par_present = (c.par & SLOT_PRESENT) != 0;
    const bool par_founder = (c.par & SLOT_FOUNDER) != 0;
    const bool tr_present  = (c.tr & SLOT_PRESENT) != 0;
    for (int fp = 0; fp < 2; fp++)
        for (int fgt = 0; fgt < 2; fgt++)
            for (int fgo = 0; fgo < 2; fgo++) {
                if (!line_path_ok(c, fp, fgt, fgo, force_par, force_tr, force_ot)) continue;
                const double lp1 = line_path_term(TP[0], fp, fgt, fgo), lp2 = line_path_term(TP[1], fp, fgt, fgo);
                const double term = line_path_term(T, fp, fgt, fgo);
                if (term == 0.0) continue;                 // val == 0: the reference never gets here (cnF2freq.cpp:5502)
                A->rtot += term;
                A->h[0] += lp1;
                A->h[1] += lp2;
                const double den = R[0] * lp1 + R[1] * lp2;
                for (int i = 0; i < 2; i++) {
                    const double wi = (R[i] * (i ? lp2 : lp1)) / den;      // 0/0 is the reference's own NaN
                    const double tw = term * wi;
                    A->wh[i] += tw;
                    // GENOS chain for allele i + 1 (cnF2freq.cpp:1191-1245, 1271, 1336, 1351): traced line only
                    double cg = 0.0, cp = 1.0;
                    bool   gp_visited = false, par_visited = false;
                    const double pre_r = R[i] * pwroot;
                    if (pre_r != 0.0 && par_present) {
                        par_visited = true;
                        const double pre_p = probe_base(i + 1, fp ? par.a1 : par.a0, fp ? par.s1 : par.s0) *
                                             phase_weight(par, fp ^ c.firstpar ^ c.sp);
                        cp = pre_p;
                        if (!par_founder && pre_p != 0.0 && tr_present) {
                            gp_visited = true;
                            cg = probe_base(i + 1, fgt ? tr.a1 : tr.a0, fgt ? tr.s1 : tr.s0) * phase_weight(tr, fgt ^ c.bit_tr);
                            cp = pre_p * cg;
                        }
                    }
                    if (pre_r * cp != 0.0) A->wroot[i] += tw;
                    if (par_visited && cp != 0.0) A->wpar[fp][i] += tw;
                    if (gp_visited && cg != 0.0) A->wgp[fgt][i] += tw;
                }
            }
}

// All of it for one (g, s) at one marker.  slot[k]: data of window slot k at the marker (blank for missing
// slots); wg = exp(scales - factor) * alphaminus_s(g) * beta_s(g).  Adds into inf[7][2][2] (slot, allele
// index, markerval - 1) and hz[2].
CNF2_HD void accum_infprobs(const Window& w, const Slot slot[7], int g, int s, double wg, bool no_ties,
                            double* inf, double* hz)
{
    const Slot& root       = slot[0];
    const bool  root_attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    const int   n_combo    = no_ties ? 1 : (1 << w.n_groups);
    for (int f = 0; f < 2; f++) {
        RootTerms R;
        root_terms(root, root_attop, f, &R);
        const double pw = phase_weight(root, f ^ (s & 1));
        const double cf = R.cbase * pw;
        if (cf == 0.0) continue;                              // every val of this f is zero
        const int    mf = f ? root.a1 : root.a0, mo = f ? root.a0 : root.a1;
        const double sf = f ? root.s1 : root.s0, so = f ? root.s0 : root.s1;
        double Rs[2][2], Xo[2];                               // [side][i-1]
        for (int i = 0; i < 2; i++) {
            Rs[0][i] = probe_base(i + 1, mf, sf);             // side 0 probes the root's allele f
            Rs[1][i] = probe_base(i + 1, mo, so);             // side 1 the other one (flag2 ^ 1, cnF2freq.cpp:5525)
            Xo[i]    = (i + 1 != mo) ? (mo != 0 ? so : 1.0) : (1.0 - so);      // cnF2freq.cpp:1304-1318
        }
        if (root_attop) {
            // the root is the top of its lines: no recursion except in HOMOZYGOUS mode (cnF2freq.cpp:1120)
            const double val = wg * cf;
            for (int i = 0; i < 2; i++) {
                inf[(0 * 2 + f) * 2 + i]       += val * (Rs[0][i] / (Rs[0][0] + Rs[0][1]));
                inf[(0 * 2 + (f ^ 1)) * 2 + i] += val * (Rs[1][i] / (Rs[1][0] + Rs[1][1]));
            }
        }
        LaneJob   L[2];
        LineTerms T[2], TP[2][2];
        make_lane(w, (0 << 5) | (f << 4) | (((s >> 1) & 1) << 3) | (g & 7), &L[0]);
        make_lane(w, (1 << 5) | (f << 4) | (((s >> 2) & 1) << 3) | (g >> 3), &L[1]);
        for (int P = 0; P < 2; P++) {
            const int sp = 1 + 3 * P;
            const Slot &par = slot[sp], &tr = slot[sp + 1 + L[P].cfg.firstpar], &ot = slot[sp + 1 + (L[P].cfg.firstpar ^ 1)];
            line_terms(L[P].cfg, par, tr, ot, P ? R.inmv1 : R.inmv0, P ? R.sv1 : R.sv0, P == 0 && R.inmv0 == 2, &T[P]);
            for (int i = 0; i < 2; i++) line_terms(L[P].cfg, par, tr, ot, i + 1, 0.0, false, &TP[P][i]);
        }
        if (root_attop) {
            // HOMOZYGOUS at a founder root still recurses, along the all-zero path (flag2ignore leaves bit 0 only)
            const double val = wg * cf;
            const double den = Rs[0][0] + Rs[0][1];
            for (int i = 0; i < 2; i++)
                hz[i] += val * ((Rs[0][i] * Xo[i] * line_path_term(TP[1][i], 0, 0, 0)) * line_path_term(TP[0][i], 0, 0, 0)) / den;
            continue;
        }
        const double d0 = (so != 0.0) ? 1.0 - so : 1.0;       // the factor GENOSPROBE side 0 applied at the root
        for (int combo = 0; combo < n_combo; combo++) {
            LineAcc A[2];
            for (int P = 0; P < 2; P++) {
                const int sp = 1 + 3 * P;
                line_accumulate(L[P].cfg, T[P], TP[P], Rs[P], pw, slot[sp], slot[sp + 1 + L[P].cfg.firstpar],
                                no_ties ? -1 : tie_force(L[P].tie_par, combo), no_ties ? -1 : tie_force(L[P].tie_tr, combo),
                                no_ties ? -1 : tie_force(L[P].tie_ot, combo), &A[P]);
            }
            if (A[0].rtot == 0.0 || A[1].rtot == 0.0) continue;      // no path of this combination has val != 0
            const double k = wg * cf;
            for (int P = 0; P < 2; P++) {
                const double other = A[P ^ 1].rtot;
                const int    sp = 1 + 3 * P, sg = sp + 1 + L[P].cfg.firstpar;
                const int    fr = P ? (f ^ 1) : f;                 // root allele index probed on this side
                for (int i = 0; i < 2; i++) {
                    inf[(0 * 2 + fr) * 2 + i] += (k * other) * A[P].wroot[i];
                    for (int x = 0; x < 2; x++) {
                        inf[(sp * 2 + x) * 2 + i] += (k * other) * A[P].wpar[x][i];
                        inf[(sg * 2 + x) * 2 + i] += (k * other) * A[P].wgp[x][i];
                    }
                }
            }
            for (int i = 0; i < 2; i++) hz[i] += k * A[0].wh[i] * (Xo[i] / d0) * A[1].h[i];
        }
    }
}

} // namespace cnf2
#endif
