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
// Tile form: one lane = one (P, f, t) of one marker, its 8 paths / 8 entries in registers.  The lane numbers them by
// ROLE, r = b2<<2 | b1<<1 | b0 with b2 = fp / sp, b1 = allele / state bit of the OTHER grandparent, b0 = of the TRACED
// one, so that nothing in registers is indexed by the run-time t; which grandparent (0 / 1) a role is only decides where
// an entry lands in the table (tile_entry4) and which slot's accumulators a sum belongs to.  The match logic of a
// parent allele is shared by its four paths, the butterflies need no lane exchange, and a wavefront covers 8 markers x
// 8 parts at once (the pattern of the sweep's tile producer, cnf2_emtab.h).
// ---------------------------------------------------------------------------------------------------------------
struct PathTile {
    double   term0[8];          // [r]
    double   w0[4], w1[4];      // GENOS weights by (fp, allele of the traced grandparent): index fp<<1 | fg
    double   k0[8], k1[8];      // [r] probe products of the HOMOZYGOUS sums (meaningful on line 1)
    uint32_t den_ok;            // bit fp<<1 | fg
};

template <int V>
struct PathConst {
    static constexpr int value = V;
};

// low four bits (sp<<3 | u1<<2 | u0<<1 | t) of the emission-table index of entry r of a lane with traced grandparent t
CNF2_HD int tile_entry4(int r, int t)
{
    const int sp = r >> 2, b_ot = (r >> 1) & 1, b_tr = r & 1;
    return (sp << 3) | (t ? (b_tr << 2) | (b_ot << 1) : (b_ot << 2) | (b_tr << 1)) | t;
}

CNF2_HD void path_terms8(const PathLine& ln, int t, const PathRoot& ar, PathTile* T)
{
    const uint32_t fl_par = ln.fl_par, fl_tr = t ? ln.fl_b : ln.fl_a, fl_ot = t ? ln.fl_a : ln.fl_b;
    const Slot&    par = ln.par;
    const Slot     tr = t ? ln.gpb : ln.gpa, ot = t ? ln.gpa : ln.gpb;
    const bool     par_present = (fl_par & SLOT_PRESENT) != 0, par_founder = (fl_par & SLOT_FOUNDER) != 0;
    const bool     tr_present = (fl_tr & SLOT_PRESENT) != 0, ot_present = (fl_ot & SLOT_PRESENT) != 0;
    T->den_ok = 0;
    // the body runs for fp = 0 and 1 with fp a compile-time constant (every index into T is static)
    auto one_parent_allele = [&](auto fpc) {
        constexpr int fp = decltype(fpc)::value;
        const int    mf = fp ? par.a1 : par.a0, mo = fp ? par.a0 : par.a1;
        const double sf = fp ? par.s1 : par.s0, so = fp ? par.s0 : par.s1;
        double B, TR[2], OT[2], l0[2], l1[2];
        bool   pbnz0, pbnz1;                                  // the probe's base factor is non-zero
        if (!par_present) {                                  // 1 + sv on the all-zero path
            B     = fp == 0 ? 1.0 : 0.0;
            TR[0] = 1.0 + ar.sv;
            TR[1] = 0.0;
            OT[0] = 1.0;
            OT[1] = 0.0;
            l0[0] = l1[0] = 1.0;
            l0[1] = l1[1] = 0.0;
            pbnz0 = pbnz1 = true;
        } else if (par_founder) {                            // the parent is the top of its line
            B     = 1.0;
            TR[0] = top_value(par, fp, ar.inmv, ar.sv);
            TR[1] = 0.0;
            OT[0] = 1.0;
            OT[1] = 0.0;
            l0[0] = probe_base(1, mf, sf);
            l1[0] = probe_base(2, mf, sf);
            l0[1] = l1[1] = 0.0;
            pbnz0 = pbnz1 = true;
        } else {
            double bv, msv;
            int    mv;
            match_term(ar.inmv, ar.sv, mf, sf, &bv, &msv, &mv);
            if (msv != 0.0) msv = path_div(msv, bv);         // cnF2freq.cpp:1220
            double ssv = 0.0, keep = 1.0;
            if (so != 0.0) {                                 // cnF2freq.cpp:1298-1302
                keep = 1.0 - so;
                ssv  = path_div(so, 1.0 - so);
            }
            B = bv * keep;
            if (!(B != 0.0)) {                               // cnF2freq.cpp:1271: nothing below a zero base
                B   = 0.0;
                msv = 0.0;
                ssv = 0.0;
            }
            const double pb0 = probe_base(1, mf, sf) * keep, pb1 = probe_base(2, mf, sf) * keep;
            pbnz0 = pb0 != 0.0;
            pbnz1 = pb1 != 0.0;
#pragma unroll
            for (int g = 0; g < 2; g++) {
                OT[g] = ot_present ? top_value(ot, g, mo, ssv) : (g == 0 ? 1.0 + ssv : 0.0);
                TR[g] = tr_present ? top_value(tr, g, mv, msv) : (g == 0 ? 1.0 + msv : 0.0);
                const double pt0 = tr_present ? probe_base(1, g ? tr.a1 : tr.a0, g ? tr.s1 : tr.s0) : (g == 0 ? 1.0 : 0.0);
                const double pt1 = tr_present ? probe_base(2, g ? tr.a1 : tr.a0, g ? tr.s1 : tr.s0) : (g == 0 ? 1.0 : 0.0);
                l0[g] = pb0 * pt0;
                l1[g] = pb1 * pt1;
            }
        }
        const bool live = B != 0.0;
#pragma unroll
        for (int fg = 0; fg < 2; fg++) {
            const double n0 = ar.R0 * l0[fg], n1 = ar.R1 * l1[fg];
            const double den = n0 + n1;
            const double inv = path_div(1.0, den);
            T->w0[fp * 2 + fg] = n0 * inv;
            T->w1[fp * 2 + fg] = n1 * inv;
            if (den != 0.0) T->den_ok |= 1u << (fp * 2 + fg);
            const bool trlive = live && TR[fg] != 0.0;
#pragma unroll
            for (int fo = 0; fo < 2; fo++) {
                const int r = (fp << 2) | (fo << 1) | fg;
                T->term0[r] = live ? (B * OT[fo]) * TR[fg] : 0.0;
                T->k0[r]    = (trlive && pbnz0) ? OT[fo] * l0[fg] : 0.0;
                T->k1[r]    = (trlive && pbnz1) ? OT[fo] * l1[fg] : 0.0;
            }
        }
    };
    one_parent_allele(PathConst<0>());
    one_parent_allele(PathConst<1>());
}

// The 2 x 2 factors of the three roles, [allele][state bit]; codes as in path_coef_plan: 12 x 2 bits from bit 0 up in
// the order par[0][0], par[0][1], par[1][0], par[1][1], tr[..], ot[..]
struct PathMats {
    double par[2][2], tr[2][2], ot[2][2];
};
CNF2_HD uint32_t path_mats_plan(const PathLine& ln, int t, int combo, bool no_ties)
{
    const bool par_present = (ln.fl_par & SLOT_PRESENT) != 0, general = par_present && !(ln.fl_par & SLOT_FOUNDER);
    const int  force_par = no_ties ? -1 : tie_force(ln.tie_par, combo);
    const int  force_tr = no_ties ? -1 : tie_force(t ? ln.tie_b : ln.tie_a, combo);
    const int  force_ot = no_ties ? -1 : tie_force(t ? ln.tie_a : ln.tie_b, combo);
    const uint32_t fl_tr = t ? ln.fl_b : ln.fl_a, fl_ot = t ? ln.fl_a : ln.fl_b;
    const bool rt = general && (fl_tr & SLOT_PRESENT), ro = general && (fl_ot & SLOT_PRESENT);
    uint32_t plan = 0;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = a * 2 + u;
            plan |= (uint32_t)path_coef_code(par_present, ln.fl_par, a, t, a ^ t ^ u, force_par) << (2 * k);
            plan |= (uint32_t)path_coef_code(rt, fl_tr, a, u, a ^ u, force_tr) << (8 + 2 * k);
            plan |= (uint32_t)path_coef_code(ro, fl_ot, a, u, a ^ u, force_ot) << (16 + 2 * k);
        }
    return plan;
}
CNF2_HD void path_mats_apply(uint32_t plan, const PathLine& ln, int t, PathMats* M)
{
    const double p0 = phase_weight(ln.par, 0), p1 = phase_weight(ln.par, 1);
    const double a0 = phase_weight(ln.gpa, 0), a1 = phase_weight(ln.gpa, 1);
    const double b0 = phase_weight(ln.gpb, 0), b1 = phase_weight(ln.gpb, 1);
    const double t0 = t ? b0 : a0, t1 = t ? b1 : a1, o0 = t ? a0 : b0, o1 = t ? a1 : b1;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = a * 2 + u;
            M->par[a][u] = path_coef_value(plan >> (2 * k), p0, p1);
            M->tr[a][u]  = path_coef_value(plan >> (8 + 2 * k), t0, t1);
            M->ot[a][u]  = path_coef_value(plan >> (16 + 2 * k), o0, o1);
        }
}
// one butterfly position of an 8-vector in registers: (out0, out1) = C (in0, in1) for every pair differing in bit BIT
template <int BIT>
CNF2_HD void tile_stage(double (&X)[8], double c00, double c01, double c10, double c11)
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (r & (1 << BIT)) continue;
        const double x0 = X[r], x1 = X[r | (1 << BIT)];
        X[r]              = c00 * x0 + c01 * x1;
        X[r | (1 << BIT)] = c10 * x0 + c11 * x1;
    }
}
// paths -> entries (out[u] = sum_a E[a][u] in[a]) and entries -> paths (out[a] = sum_u E[a][u] in[u]) of one role
template <int BIT>
CNF2_HD void tile_fwd(double (&X)[8], const double (&E)[2][2]) { tile_stage<BIT>(X, E[0][0], E[1][0], E[0][1], E[1][1]); }
template <int BIT>
CNF2_HD void tile_bwd(double (&X)[8], const double (&E)[2][2]) { tile_stage<BIT>(X, E[0][0], E[0][1], E[1][0], E[1][1]); }

// What one (P, f, t) lane adds up over its paths once the weights of its 8 entries are known
struct TileSums {
    double inf_root[2];       // [i]       -> infprobs of the root, allele index f ^ P
    double inf_par[2][2];     // [fp][i]
    double inf_tr[2][2];      // [allele of the traced grandparent][i]
    double hz[2];             // [i] (line 0; already scaled)
    double hap_par[2], hap_tr[2], hap_ot[2];   // [phase]
};
// wt[r]: weights of the lane's entries.  Flags of the line: do_par: the parent takes part; do_tr / do_ot: the traced /
// other grandparent has HAPLOS sums; do_inftr: the traced grandparent has infprobs.  Everything but homozyg; written as
// a sequence of short passes so that only one or two 8-vectors are live at a time.
CNF2_HD void tile_accumulate(const PathTile& T, const PathMats& M, int t, const double (&wt)[8], bool do_par, bool do_tr,
                             bool do_ot, bool do_inftr, TileSums* S)
{
    S->hap_tr[0] = S->hap_tr[1] = S->hap_ot[0] = S->hap_ot[1] = 0.0;
    {
        // parent + traced grandparent done: the other grandparent's stage is split by phase fo ^ its state bit
        double Z[8];
#pragma unroll
        for (int r = 0; r < 8; r++) Z[r] = wt[r];
        tile_bwd<0>(Z, M.tr);
        tile_bwd<2>(Z, M.par);
        if (do_ot) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int fo = (r >> 1) & 1;
                S->hap_ot[0] += T.term0[r] * (M.ot[fo][fo] * Z[r]);
                S->hap_ot[1] += T.term0[r] * (M.ot[fo][fo ^ 1] * Z[r ^ 2]);
            }
        }
    }
    double X[8];
#pragma unroll
    for (int r = 0; r < 8; r++) X[r] = wt[r];
    tile_bwd<1>(X, M.ot);                                    // other grandparent done
    {
        double Y[8];
#pragma unroll
        for (int r = 0; r < 8; r++) Y[r] = X[r];
        tile_bwd<2>(Y, M.par);                               // parent + other done: the traced grandparent's stage is split
        if (do_tr) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int fg = r & 1;
                S->hap_tr[0] += T.term0[r] * (M.tr[fg][fg] * Y[r]);
                S->hap_tr[1] += T.term0[r] * (M.tr[fg][fg ^ 1] * Y[r ^ 1]);
            }
        }
    }
    tile_bwd<0>(X, M.tr);                                    // both grandparents done: the parent's stage is split by
    double hp_same = 0.0, hp_flip = 0.0;                     // phase fp ^ t ^ sp: sp == fp (phase t) / sp != fp (phase !t)
    if (do_par) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int fp = r >> 2;
            hp_same += T.term0[r] * (M.par[fp][fp] * X[r]);
            hp_flip += T.term0[r] * (M.par[fp][fp ^ 1] * X[r ^ 4]);
        }
    }
    S->hap_par[0] = t ? hp_flip : hp_same;
    S->hap_par[1] = t ? hp_same : hp_flip;
    tile_bwd<2>(X, M.par);                                   // Omega of every path
    S->inf_root[0] = S->inf_root[1] = 0.0;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int i = 0; i < 2; i++) S->inf_par[a][i] = S->inf_tr[a][i] = 0.0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int    fp = r >> 2, fg = r & 1;
        const bool   dok = ((T.den_ok >> (fp * 2 + fg)) & 1) != 0;
        const double c = X[r] * T.term0[r];
        if (c != 0.0 && dok) {
            const double tw0 = c * T.w0[fp * 2 + fg], tw1 = c * T.w1[fp * 2 + fg];
            S->inf_root[0] += tw0;
            S->inf_root[1] += tw1;
            if (do_par) {
                S->inf_par[fp][0] += tw0;
                S->inf_par[fp][1] += tw1;
            }
            if (do_inftr) {
                S->inf_tr[fg][0] += tw0;
                S->inf_tr[fg][1] += tw1;
            }
        }
    }
}
// homozyg[i] of a line-0 lane: z[r] = HOMOZYGOUS weights of its entries for allele value i + 1, w = T.w0 or T.w1
CNF2_HD double tile_homozyg(const PathTile& T, const PathMats& M, const double (&z)[8], const double (&w)[4], double hzs)
{
    double zz[8];
#pragma unroll
    for (int r = 0; r < 8; r++) zz[r] = z[r];
    tile_bwd<1>(zz, M.ot);
    tile_bwd<0>(zz, M.tr);
    tile_bwd<2>(zz, M.par);
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double c = zz[r] * T.term0[r];
        if (c != 0.0) acc += hzs * (c * w[(r >> 2) * 2 + (r & 1)]);
    }
    return acc;
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
// Host emulation of the tile form for one marker: the 8 parts in a loop, tables and weights in arrays.
inline void acc_contract_tile(const Window& w, const Slot slot[7], const double* wg /* [8][64] */, bool no_ties,
                              double* inf, double* hz, double* hap)
{
    const int n_combo = no_ties ? 1 : (1 << w.n_groups);
    AccRoot   ar[2];
    acc_root(slot[0], false, 0, &ar[0]);
    acc_root(slot[0], false, 1, &ar[1]);
    PathLine ln[2];
    path_line(w, slot, 0, &ln[0]);
    path_line(w, slot, 1, &ln[1]);
    PathTile T[8];
    PathRoot pr[8];
    for (int part = 0; part < 8; part++) {
        const int P = part >> 2, f = (part >> 1) & 1, t = part & 1;
        path_root(slot[0], f, P, &pr[part]);
        path_terms8(ln[P], t, pr[part], &T[part]);
        if (!ar[f].live)
            for (int r = 0; r < 8; r++) T[part].term0[r] = T[part].k0[r] = T[part].k1[r] = 0.0;
    }
    for (int combo = 0; combo < n_combo; combo++) {
        PathMats M[8];
        double   Rt[64], Ht[2][64], Rent[8][8];
        for (int part = 0; part < 8; part++) {
            const int P = part >> 2, f = (part >> 1) & 1, t = part & 1;
            path_mats_apply(path_mats_plan(ln[P], t, combo, no_ties), ln[P], t, &M[part]);
            double R[8], H0[8], H1[8];
            for (int r = 0; r < 8; r++) {
                R[r]  = T[part].term0[r];
                H0[r] = T[part].k0[r];
                H1[r] = T[part].k1[r];
            }
            for (auto* X : {&R, &H0, &H1}) {
                tile_fwd<0>(*X, M[part].tr);
                tile_fwd<1>(*X, M[part].ot);
                tile_fwd<2>(*X, M[part].par);
            }
            for (int r = 0; r < 8; r++) {
                const int e = (P << 5) | (f << 4) | tile_entry4(r, t);
                Rt[e]    = R[r];
                Ht[0][e] = H0[r];
                Ht[1][e] = H1[r];
                Rent[part][r] = R[r];
            }
        }
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
        for (int part = 0; part < 8; part++) {
            const int P = part >> 2, f = (part >> 1) & 1, t = part & 1, sl = 1 + 3 * P;
            if (!ar[f].live) continue;
            double wt[8], z0[8], z1[8];
            for (int r = 0; r < 8; r++) {
                const int e4 = tile_entry4(r, t);
                wt[r] = P ? u[f][e4] : v[f][0][e4] + v[f][1][e4];
                z0[r] = P ? 0.0 : z[f][0][e4];
                z1[r] = P ? 0.0 : z[f][1][e4];
                if (P == 0)
                    for (int s0 = 0; s0 < 2; s0++) hap[0 * 2 + (f ^ s0)] += Rent[part][r] * v[f][s0][e4];
            }
            const bool     general = (ln[P].fl_par & (SLOT_PRESENT | SLOT_FOUNDER)) == SLOT_PRESENT;
            const uint32_t fl_tr = t ? ln[P].fl_b : ln[P].fl_a, fl_ot = t ? ln[P].fl_a : ln[P].fl_b;
            const bool     trp = general && (fl_tr & SLOT_PRESENT), otp = general && (fl_ot & SLOT_PRESENT);
            TileSums S;
            tile_accumulate(T[part], M[part], t, wt, (ln[P].fl_par & SLOT_PRESENT) != 0, trp, otp, trp, &S);
            S.hz[0] = P ? 0.0 : tile_homozyg(T[part], M[part], z0, T[part].w0, pr[part].hzscale0);
            S.hz[1] = P ? 0.0 : tile_homozyg(T[part], M[part], z1, T[part].w1, pr[part].hzscale1);
            const int fr = f ^ P;
            for (int i = 0; i < 2; i++) {
                inf[(0 * 2 + fr) * 2 + i] += S.inf_root[i];
                for (int a = 0; a < 2; a++) {
                    inf[(sl * 2 + a) * 2 + i] += S.inf_par[a][i];
                    inf[((sl + 1 + t) * 2 + a) * 2 + i] += S.inf_tr[a][i];
                }
                hz[i] += S.hz[i];
            }
            for (int ph = 0; ph < 2; ph++) {
                hap[sl * 2 + ph] += S.hap_par[ph];
                hap[(sl + 1 + t) * 2 + ph] += S.hap_tr[ph];
                hap[(sl + 1 + (t ^ 1)) * 2 + ph] += S.hap_ot[ph];
            }
        }
    }
}
#endif

} // namespace cnf2
#endif
