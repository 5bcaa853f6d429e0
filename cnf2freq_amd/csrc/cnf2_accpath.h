// cnf2_accpath.h -- path form of the HOT LOOP 2 accumulators (cnF2freq.cpp:5416-5577) for ONE (individual, marker):
// the same sums as the table form of cnf2_acctab.h, organised so that a wavefront evaluates every match term once.
// Host + device; the host emulation at the end is unit-tested against acc_contract_scalar() (tests/test_host_emission.py).
//
// A table entry of a line (P, f) is indexed by the line's state bits (sp, t, u0, u1): shift bit of the parent, which
// grandparent is traced, the grandparents' own bits.  Its value is a sum over the allele indices (fp, g0, g1) taken
// at the parent and the two grandparents ("a path"), and every state bit enters through ONE factor only:
//     entry(sp, t, u0, u1) = sum_{fp, g0, g1} Epar_t[fp][sp] E0[g0][u0] E1[g1][u1] * term0_t(fp, g0, g1)
// with E = phase weight of the slot at parity (allele ^ bit) times its admissibility under flag2ignore / the tie rule
// (a slot that is missing or the top of its line has E[a][u] = [a == 0]), and term0 the product of the three match
// terms, which do not see the state.  So for a fixed (P, f, t)
//   * paths -> entries is a separable 2x2x2 transform (three butterfly stages), and
//   * an accumulator  sum_e weight(e) * (sum over the paths of e of term * something of the path)  is
//     sum_paths Omega(path) * term0(path) * something,  Omega = the transposed transform of the entry weights.
// The GENOS weights w_i = R_i l_i / (R_0 l_0 + R_1 l_1) of a path are ratios of probe products that carry the same
// phase weights in numerator and denominator, so they are functions of the path alone.  HAPLOS splits a sum by the
// parity (allele ^ bit) of one slot = by the two terms of that slot's butterfly stage.
//
// Lane numbering of this form:  L = P<<5 | f<<4 | b3<<3 | t<<2 | b1<<1 | b0
//     path lane : b3 = fp, b1 = g1, b0 = g0          entry lane: b3 = sp, b1 = u1, b0 = u0
// (t is kept out of the butterfly positions: bits 0, 1 and 3 are plain DPP exchanges on gfx950).
#ifndef CNF2_ACCPATH_H
#define CNF2_ACCPATH_H

#include "cnf2_acctab.h"

namespace cnf2 {

// a / b for the ratios of this form.  Device: reciprocal + two Newton steps + one residual correction (no scaling for
// denormal operands, which probabilities and their products are not); 0 / 0 is NaN as with the plain division.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double path_div(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
#else
inline double path_div(double a, double b) { return a / b; }
#endif

// What line P of a path lane receives from the root for root allele f (root_terms() + acc_root() restricted to what
// one lane reads; the root is not the top of its lines): incoming allele and error odds, probe bases R_i, and the
// HOMOZYGOUS scale Xo_i / d0 (cnF2freq.cpp:1304-1318).
struct PathRoot {
    int    inmv;
    double sv;
    double R0, R1;
    double hzscale0, hzscale1;
};
CNF2_HD void path_root(const Slot& root, int f, int P, PathRoot* A)
{
    const int    mf = f ? root.a1 : root.a0, mo = f ? root.a0 : root.a1;
    const double sf = f ? root.s1 : root.s0, so = f ? root.s0 : root.s1;
    // line 0 gets (mf, msv / baseval), line 1 gets (mo, so / (1 - so)): one division, operands selected first
    const double num = P ? so : ((mf != 0) ? sf : 0.0);
    const double den = P ? 1.0 - so : 1.0 - sf;
    A->inmv = P ? mo : mf;
    A->sv   = (num != 0.0) ? path_div(num, den) : 0.0;
    A->R0   = probe_base(1, P ? mo : mf, P ? so : sf);
    A->R1   = probe_base(2, P ? mo : mf, P ? so : sf);
    const double inv = (so != 0.0) ? path_div(1.0, 1.0 - so) : 1.0;
    A->hzscale0 = ((1 != mo) ? (mo != 0 ? so : 1.0) : (1.0 - so)) * inv;
    A->hzscale1 = ((2 != mo) ? (mo != 0 ? so : 1.0) : (1.0 - so)) * inv;
}
// c_f without the phase weight (root_terms().cbase for a root that is not the top of its lines)
CNF2_HD double path_root_cbase(const Slot& root, int f)
{
    const double sf = f ? root.s1 : root.s0, so = f ? root.s0 : root.s1;
    double       b = 1.0 - sf;
    if (so != 0.0) b *= (1.0 - so);
    return b;
}

// index of entry lane L in the emission-table numbering of cnf2_lane.h (P<<5 | f<<4 | sp<<3 | u1<<2 | u0<<1 | t)
CNF2_HD int path_entry_index(int L) { return (L & 0x38) | ((L & 2) << 1) | ((L & 1) << 1) | ((L >> 2) & 1); }

enum { PO_PAR = 1, PO_G0 = 2, PO_G1 = 4, PO_TR = 8 };

// what does not depend on the state bits, for path lane L
struct PathTerms {
    double   term0;     // product of the line's three match terms on this path
    double   w0, w1;    // GENOS weights of the path (NaN where no probe is possible: the reference's own 0 / 0)
    double   k0, k1;    // probe products of the HOMOZYGOUS sums (used on line 1)
    bool     den_ok;
    uint32_t out;       // PO_PAR: the parent takes part (HAPLOS, infprobs); PO_G0 / PO_G1: grandparent 0 / 1 does
                        // (HAPLOS); PO_TR: the traced grandparent does (infprobs)
};

// value of allele index fa of a slot at the top of a line, without its phase weight (top_terms, cnF2freq.cpp:1213-1217)
CNF2_HD double top_value(const Slot& d, int fa, int inmv, double sv)
{
    double bv, msv;
    int    mv;
    match_term(inmv, sv, fa ? d.a1 : d.a0, fa ? d.s1 : d.s0, &bv, &msv, &mv);
    return bv + msv;
}

// what a lane needs of the line it works on (P = its lane bit 5): data of the parent and its two parents at the marker,
// their window flags and tie groups
struct PathLine {
    Slot     par, gpa, gpb;
    uint32_t fl_par, fl_a, fl_b;
    int8_t   tie_par, tie_a, tie_b;
};
CNF2_HD void path_line(const Window& w, const Slot slot[7], int P, PathLine* ln)
{
    ln->par = P ? slot[4] : slot[1];
    ln->gpa = P ? slot[5] : slot[2];
    ln->gpb = P ? slot[6] : slot[3];
    ln->fl_par = P ? w.flags[4] : w.flags[1];
    ln->fl_a   = P ? w.flags[5] : w.flags[2];
    ln->fl_b   = P ? w.flags[6] : w.flags[3];
    ln->tie_par = P ? w.tie[4] : w.tie[1];
    ln->tie_a   = P ? w.tie[5] : w.tie[2];
    ln->tie_b   = P ? w.tie[6] : w.tie[3];
}

CNF2_HD void path_terms(const PathLine& ln, int L, const PathRoot& ar, PathTerms* T)
{
    const int fp = (L >> 3) & 1, t = (L >> 2) & 1, g1 = (L >> 1) & 1, g0 = L & 1;
    const int fg = t ? g1 : g0, fo = t ? g0 : g1;
    const uint32_t fl_par = ln.fl_par, fl_a = ln.fl_a, fl_b = ln.fl_b;
    const uint32_t fl_tr = t ? fl_b : fl_a, fl_ot = t ? fl_a : fl_b;
    const Slot& par = ln.par;
    const Slot  tr = t ? ln.gpb : ln.gpa, ot = t ? ln.gpa : ln.gpb;
    const int    inmv = ar.inmv;
    const double sv = ar.sv;
    const double R0 = ar.R0, R1 = ar.R1;
    const bool   par_present = (fl_par & SLOT_PRESENT) != 0, par_founder = (fl_par & SLOT_FOUNDER) != 0;
    const bool   general = par_present && !par_founder;
    const int    mf = fp ? par.a1 : par.a0, mo = fp ? par.a0 : par.a1;
    const double sf = fp ? par.s1 : par.s0, so = fp ? par.s0 : par.s1;
    double B, TRm, OTm, pb0, pb1, pt0, pt1;
    if (!par_present) {                                     // line_terms(): 1 + sv on the all-zero path
        B   = (fp | fg | fo) == 0 ? 1.0 : 0.0;
        OTm = 1.0;
        TRm = 1.0 + sv;
        pb0 = pb1 = pt0 = pt1 = 1.0;
    } else if (par_founder) {                               // the parent is the top of its line
        B   = (fg | fo) == 0 ? 1.0 : 0.0;
        OTm = 1.0;
        TRm = top_value(par, fp, inmv, sv);
        pb0 = pb1 = 1.0;
        pt0 = probe_base(1, mf, sf);
        pt1 = probe_base(2, mf, sf);
    } else {
        double bv, msv;
        int    mv;
        match_term(inmv, sv, mf, sf, &bv, &msv, &mv);
        if (msv != 0.0) msv = path_div(msv, bv);             // cnF2freq.cpp:1220
        double ssv = 0.0, keep = 1.0;
        if (so != 0.0) {                                    // cnF2freq.cpp:1298-1302
            keep = 1.0 - so;
            ssv  = path_div(so, 1.0 - so);
        }
        B = bv * keep;
        if (!(B != 0.0)) {                                  // cnF2freq.cpp:1271: nothing below a zero base
            B   = 0.0;
            msv = 0.0;
            ssv = 0.0;
        }
        OTm = (fl_ot & SLOT_PRESENT) ? top_value(ot, fo, mo, ssv) : (fo == 0 ? 1.0 + ssv : 0.0);
        TRm = (fl_tr & SLOT_PRESENT) ? top_value(tr, fg, mv, msv) : (fg == 0 ? 1.0 + msv : 0.0);
        pb0 = probe_base(1, mf, sf) * keep;
        pb1 = probe_base(2, mf, sf) * keep;
        if (fl_tr & SLOT_PRESENT) {
            pt0 = probe_base(1, fg ? tr.a1 : tr.a0, fg ? tr.s1 : tr.s0);
            pt1 = probe_base(2, fg ? tr.a1 : tr.a0, fg ? tr.s1 : tr.s0);
        } else {
            pt0 = pt1 = (fg == 0) ? 1.0 : 0.0;
        }
    }
    const bool live = B != 0.0;
    T->term0 = live ? (B * OTm) * TRm : 0.0;
    const double l0 = pb0 * pt0, l1 = pb1 * pt1;
    const double den = R0 * l0 + R1 * l1;
    const double inv = path_div(1.0, den);
    T->w0 = (R0 * l0) * inv;
    T->w1 = (R1 * l1) * inv;
    T->den_ok = den != 0.0;
    const bool trlive = live && TRm != 0.0;
    T->k0 = (trlive && pb0 != 0.0) ? (pb0 * OTm) * pt0 : 0.0;
    T->k1 = (trlive && pb1 != 0.0) ? (pb1 * OTm) * pt1 : 0.0;
    T->out = (par_present ? PO_PAR : 0) | ((general && (fl_a & SLOT_PRESENT)) ? PO_G0 : 0) |
             ((general && (fl_b & SLOT_PRESENT)) ? PO_G1 : 0) | ((general && (fl_tr & SLOT_PRESENT)) ? PO_TR : 0);
}

// E[a][u] of one slot of the line = its phase weight at parity a ^ u times its admissibility; [a == 0] for a slot that
// is not walked as a slot of its own (missing, or below a parent that is the top of its line).
// The three coefficients a lane needs per butterfly position: s = E[b][b] (own term in both directions),
// f = E[!b][b] (partner's term, paths -> entries), r = E[b][!b] (partner's term, entries -> paths); b = the lane's bit.
struct PathCoef {
    double par_s, par_f, par_r;
    double g0_s, g0_f, g0_r;
    double g1_s, g1_f, g1_r;
};

// Which value each of the nine coefficients takes -- 0: 0.0, 1: 1.0, 2 / 3: the slot's phase weight at phase 0 / 1 --
// as 2-bit codes (par_s, par_f, par_r, g0_s, g0_f, g0_r, g1_s, g1_f, g1_r from bit 0 up).  Depends on the window, the
// lane and the tie combination only: formed once per job where there are no tie groups.
CNF2_HD int path_coef_code(bool real, uint32_t fl, int a, int okbit, int phase, int force)
{
    if (!real) return a == 0 ? 1 : 0;
    return allele_ok(fl, a, okbit, force) ? 2 + (phase & 1) : 0;
}
CNF2_HD uint32_t path_coef_plan(const PathLine& ln, int L, int combo, bool no_ties)
{
    const int b3 = (L >> 3) & 1, t = (L >> 2) & 1, b1 = (L >> 1) & 1, b0 = L & 1;
    const bool par_present = (ln.fl_par & SLOT_PRESENT) != 0, general = par_present && !(ln.fl_par & SLOT_FOUNDER);
    const int  force_par = no_ties ? -1 : tie_force(ln.tie_par, combo), force_a = no_ties ? -1 : tie_force(ln.tie_a, combo),
               force_b = no_ties ? -1 : tie_force(ln.tie_b, combo);
    const bool ra = general && (ln.fl_a & SLOT_PRESENT), rb = general && (ln.fl_b & SLOT_PRESENT);
    uint32_t plan = 0;
    // parent: admissibility by fp ^ t, phase by fp ^ t ^ sp (cnF2freq.cpp:1227-1245 with the parent's localshift)
    plan |= path_coef_code(par_present, ln.fl_par, b3, t, t, force_par) << 0;            // fp = sp = b3
    plan |= path_coef_code(par_present, ln.fl_par, b3 ^ 1, t, t ^ 1, force_par) << 2;    // partner's fp, own sp
    plan |= path_coef_code(par_present, ln.fl_par, b3, t, t ^ 1, force_par) << 4;        // own fp, partner's sp
    // grandparents: admissibility and phase by g ^ u
    plan |= path_coef_code(ra, ln.fl_a, b0, b0, 0, force_a) << 6;
    plan |= path_coef_code(ra, ln.fl_a, b0 ^ 1, b0, 1, force_a) << 8;
    plan |= path_coef_code(ra, ln.fl_a, b0, b0 ^ 1, 1, force_a) << 10;
    plan |= path_coef_code(rb, ln.fl_b, b1, b1, 0, force_b) << 12;
    plan |= path_coef_code(rb, ln.fl_b, b1 ^ 1, b1, 1, force_b) << 14;
    plan |= path_coef_code(rb, ln.fl_b, b1, b1 ^ 1, 1, force_b) << 16;
    return plan;
}
CNF2_HD double path_coef_value(uint32_t code, double q0, double q1)
{
    code &= 3;
    return code == 0 ? 0.0 : (code == 1 ? 1.0 : (code == 2 ? q0 : q1));
}
CNF2_HD void path_coef_apply(uint32_t plan, const PathLine& ln, PathCoef* C)
{
    const double p0 = phase_weight(ln.par, 0), p1 = phase_weight(ln.par, 1);
    const double a0 = phase_weight(ln.gpa, 0), a1 = phase_weight(ln.gpa, 1);
    const double b0 = phase_weight(ln.gpb, 0), b1 = phase_weight(ln.gpb, 1);
    C->par_s = path_coef_value(plan >> 0, p0, p1);
    C->par_f = path_coef_value(plan >> 2, p0, p1);
    C->par_r = path_coef_value(plan >> 4, p0, p1);
    C->g0_s  = path_coef_value(plan >> 6, a0, a1);
    C->g0_f  = path_coef_value(plan >> 8, a0, a1);
    C->g0_r  = path_coef_value(plan >> 10, a0, a1);
    C->g1_s  = path_coef_value(plan >> 12, b0, b1);
    C->g1_f  = path_coef_value(plan >> 14, b0, b1);
    C->g1_r  = path_coef_value(plan >> 16, b0, b1);
}
CNF2_HD void path_coef(const PathLine& ln, int L, int combo, bool no_ties, PathCoef* C)
{
    path_coef_apply(path_coef_plan(ln, L, combo, no_ties), ln, C);
}

// ---------------------------------------------------------------------------------------------------------------
// Host emulation of the wavefront algorithm (arrays of 64 in place of lanes): wg[s][g] -> inf[28], hz[2], hap[14]
// (added), for a window whose root is not the top of its lines.  Same contract as acc_contract_scalar().
// ---------------------------------------------------------------------------------------------------------------
#if !defined(__HIP_DEVICE_COMPILE__)
inline void path_butterfly(double X[64], const double s[64], const double o[64], int bit)
{
    double Y[64];
    for (int L = 0; L < 64; L++) Y[L] = s[L] * X[L] + o[L] * X[L ^ (1 << bit)];
    for (int L = 0; L < 64; L++) X[L] = Y[L];
}

inline void acc_contract_paths(const Window& w, const Slot slot[7], const double* wg /* [8][64] */, bool no_ties,
                               double* inf, double* hz, double* hap)
{
    const int n_combo = no_ties ? 1 : (1 << w.n_groups);
    AccRoot   ar[2];
    acc_root(slot[0], false, 0, &ar[0]);
    acc_root(slot[0], false, 1, &ar[1]);
    PathLine  ln[2];
    path_line(w, slot, 0, &ln[0]);
    path_line(w, slot, 1, &ln[1]);
    PathTerms T[64];
    for (int L = 0; L < 64; L++) {
        PathRoot pr;
        path_root(slot[0], (L >> 4) & 1, L >> 5, &pr);
        path_terms(ln[L >> 5], L, pr, &T[L]);
    }
    for (int combo = 0; combo < n_combo; combo++) {
        PathCoef C[64];
        double   ps[64], pf[64], pr[64], as[64], af[64], ar_[64], bs[64], bf[64], br[64];
        for (int L = 0; L < 64; L++) {
            path_coef(ln[L >> 5], L, combo, no_ties, &C[L]);
            ps[L] = C[L].par_s, pf[L] = C[L].par_f, pr[L] = C[L].par_r;
            as[L] = C[L].g0_s, af[L] = C[L].g0_f, ar_[L] = C[L].g0_r;
            bs[L] = C[L].g1_s, bf[L] = C[L].g1_f, br[L] = C[L].g1_r;
        }
        // 1. paths -> entries: restricted totals of all 64 entries, HOMOZYGOUS probe sums of line 1
        double R[64], H0[64], H1[64];
        for (int L = 0; L < 64; L++) {
            const bool live = ar[(L >> 4) & 1].live;
            R[L]  = live ? T[L].term0 : 0.0;
            H0[L] = live ? T[L].k0 : 0.0;
            H1[L] = live ? T[L].k1 : 0.0;
        }
        for (double* X : {R, H0, H1}) {
            path_butterfly(X, as, af, 0);
            path_butterfly(X, bs, bf, 1);
            path_butterfly(X, ps, pf, 3);
        }
        double Rt[64], Ht[2][64];
        for (int L = 0; L < 64; L++) {
            Rt[path_entry_index(L)]    = R[L];
            Ht[0][path_entry_index(L)] = H0[L];
            Ht[1][path_entry_index(L)] = H1[L];
        }
        // 2. contractions of wg with the other line's totals (as acc_contract_scalar)
        double v[2][2][16], u[2][16], z[2][2][16];
        for (int f = 0; f < 2; f++)
            for (int e = 0; e < 16; e++) v[f][0][e] = v[f][1][e] = u[f][e] = z[f][0][e] = z[f][1][e] = 0.0;
        for (int f = 0; f < 2; f++) {
            if (!ar[f].live) continue;
            for (int s = 0; s < 8; s++) {
                const int    s0 = s & 1, e0s = ((s >> 1) & 1) << 3, e1s = ((s >> 2) & 1) << 3;
                const double cf = ar[f].cf[s0];
                if (cf == 0.0) continue;
                for (int g = 0; g < 64; g++) {
                    const double x = wg[s * 64 + g];
                    if (x == 0.0) continue;
                    const int e0 = e0s | (g & 7), e1 = e1s | (g >> 3);
                    v[f][s0][e0] += (cf * x) * Rt[(1 << 5) | (f << 4) | e1];
                    u[f][e1] += (cf * x) * Rt[(0 << 5) | (f << 4) | e0];
                    for (int i = 0; i < 2; i++) z[f][i][e0] += (cf * x) * Ht[i][(1 << 5) | (f << 4) | e1];
                }
            }
        }
        // 3. entry weights; HAPLOS of the root straight from the entries (phase f ^ s0, cnF2freq.cpp:1227)
        double wt[64], z0[64], z1[64];
        for (int L = 0; L < 64; L++) {
            const int P = L >> 5, f = (L >> 4) & 1, e4 = path_entry_index(L) & 15;
            wt[L] = P ? u[f][e4] : v[f][0][e4] + v[f][1][e4];
            z0[L] = P ? 0.0 : z[f][0][e4];
            z1[L] = P ? 0.0 : z[f][1][e4];
            if (P == 0 && ar[f].live)
                for (int s0 = 0; s0 < 2; s0++) hap[0 * 2 + (f ^ s0)] += mul0(R[L], v[f][s0][e4]);
        }
        // 4. entries -> paths, with the split stage last for each slot
        double X1[64], X01[64], X0[64], Y[64], Z[64];
        for (int L = 0; L < 64; L++) X1[L] = X0[L] = wt[L];
        path_butterfly(X1, bs, br, 1);                       // grandparent 1 done
        for (int L = 0; L < 64; L++) X01[L] = Y[L] = X1[L];
        path_butterfly(X01, as, ar_, 0);                     // both grandparents done: the parent's stage is split
        path_butterfly(Y, ps, pr, 3);                        // parent + grandparent 1 done: grandparent 0 is split
        path_butterfly(X0, as, ar_, 0);
        for (int L = 0; L < 64; L++) Z[L] = X0[L];
        path_butterfly(Z, ps, pr, 3);                        // parent + grandparent 0 done: grandparent 1 is split
        for (double* X : {z0, z1}) {
            path_butterfly(X, bs, br, 1);
            path_butterfly(X, as, ar_, 0);
            path_butterfly(X, ps, pr, 3);
        }
        // 5. per path
        for (int L = 0; L < 64; L++) {
            const int P = L >> 5, f = (L >> 4) & 1, fp = (L >> 3) & 1, t = (L >> 2) & 1, g1 = (L >> 1) & 1, g0 = L & 1;
            if (!ar[f].live) continue;
            const PathTerms& p = T[L];
            const int        sl = 1 + 3 * P;
            const double o_par_self = ps[L] * X01[L], o_par_part = pr[L] * X01[L ^ 8];      // phase t / !t
            const double o_g0_self = as[L] * Y[L], o_g0_part = ar_[L] * Y[L ^ 1];           // phase 0 / 1
            const double o_g1_self = bs[L] * Z[L], o_g1_part = br[L] * Z[L ^ 2];
            const double omega = o_par_self + o_par_part;
            if (p.out & PO_PAR) {
                hap[sl * 2 + t] += mul0(p.term0, o_par_self);
                hap[sl * 2 + (t ^ 1)] += mul0(p.term0, o_par_part);
            }
            if (p.out & PO_G0) {
                hap[(sl + 1) * 2 + 0] += mul0(p.term0, o_g0_self);
                hap[(sl + 1) * 2 + 1] += mul0(p.term0, o_g0_part);
            }
            if (p.out & PO_G1) {
                hap[(sl + 2) * 2 + 0] += mul0(p.term0, o_g1_self);
                hap[(sl + 2) * 2 + 1] += mul0(p.term0, o_g1_part);
            }
            const double c = omega * p.term0;
            if (c != 0.0 && p.den_ok) {
                const double tw[2] = {c * p.w0, c * p.w1};
                const int    fr = f ^ P;
                for (int i = 0; i < 2; i++) {
                    inf[(0 * 2 + fr) * 2 + i] += tw[i];
                    if (p.out & PO_PAR) inf[(sl * 2 + fp) * 2 + i] += tw[i];
                    if (p.out & PO_TR) inf[((sl + 1 + t) * 2 + (t ? g1 : g0)) * 2 + i] += tw[i];
                }
            }
            if (P == 0) {
                const double c0 = z0[L] * p.term0, c1 = z1[L] * p.term0;
                PathRoot pr;
                path_root(slot[0], f, 0, &pr);
                if (c0 != 0.0) hz[0] += pr.hzscale0 * (c0 * p.w0);
                if (c1 != 0.0) hz[1] += pr.hzscale1 * (c1 * p.w1);
            }
        }
    }
}
#endif

} // namespace cnf2
#endif
