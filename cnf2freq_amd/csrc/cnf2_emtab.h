// cnf2_emtab.h -- tile producer of the emission tables: division-free, branch-light form of
// cnf2_emission.h used by the fast sweep kernel (host + device; unit-tested on the host
// against cnf2_emission.h and the oracle).
//
// Work split: a wavefront prepares the tables of 8 consecutive markers at once,
//     lane = part << 3 | marker_in_tile,   part = P << 2 | f << 1 | firstpar
// and each lane produces the 8 entries (sp, bit_a, bit_b) of its (P, f, firstpar):
//     entry index = P<<5 | f<<4 | sp<<3 | k,   k = firstpar | bit_a<<1 | bit_b<<2
// (same table as cnf2_lane.h).  With firstpar fixed per lane the traced / other grandparent
// are fixed too, so the grandparent terms are evaluated once per lane and shared by its entries.
//
// Algebra.  trackpossible hands "error odds" (secondval) down the recursion after dividing
// them by the caller's baseval (cnF2freq.cpp:1220, 1301).  Every level is affine in the odds
// it receives:  value(v, sv) = V0(v) + sv * V1(v),  because the only uses of sv are
// (1-sure)*sv on a mismatch and sure*sv on a match (cnF2freq.cpp:1198-1210).  Substituting,
//     baseval * child(v, msv / baseval) = baseval * V0 + msv * V1
// so no division is needed; the reference's "!baseval => contributes 0" rule
// (cnF2freq.cpp:1271) is kept as an explicit zero.  Rounding differs from the reference at the
// 1e-16 level only.
#ifndef CNF2_EMTAB_H
#define CNF2_EMTAB_H

#include "cnf2_emission.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define CNF2_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define CNF2_SCHED_FENCE() ((void)0)
#endif

namespace cnf2 {

struct PartCfg {
    uint32_t par, tr, ot;   // SLOT_* flags of the parent, the traced and the other grandparent
    int      P, f, firstpar;
    bool     root_attop;
    // TIES only: ignoreflag2's all-or-none rule for the tie combination at hand (cnF2freq.cpp:3484-3486): -1 free, else
    // only alleles with (allele index ^ the slot's firstpar bit) == force survive in the restricted tables
    int      force_par = -1, force_tr = -1, force_ot = -1;
};

// Affine match term: value contributed before the phase weight is  Bv + C + K * sv,
// split as base (Bv), constant odds (C) and odds slope (K); *mv = allele value passed on.
CNF2_HD void match_affine(int inmv, int mf, double sf, double* Bv, double* K, double* C, int* mv)
{
    bool miss;
    if (inmv == 0) {
        *mv  = mf;
        miss = false;
    } else {
        *mv  = inmv;
        miss = !(mf == 0 && inmv != 9) && (inmv != mf);
    }
    const double effms = (mf == 0) ? 1.0 : sf;
    const bool   bound = (inmv == 0 && *mv != 0);      // effectivesecondval == 1 (cnF2freq.cpp:1205)
    *Bv = miss ? sf : 1.0 - sf;
    *K  = miss ? ((sf != 0.0) ? 1.0 - sf : 0.0) : (bound ? 0.0 : effms);
    *C  = (!miss && bound) ? effms : 0.0;
}

// What the combination below needs from a window member at one marker, in two forms:
//   SlotDirect  computes it from the raw data on the fly;
//   SlotTable   reads it from a 20-double record prepared once per (slot, marker) by slot_table() -- the tile
//               producer prepares the 7 records of a marker with 7 lanes and the 8 parts share them, instead of
//               every part redoing the match logic of its three ancestors (11 matches per part).
// A match outcome depends on the incoming value only through its class: 0 incoming unknown, 1 compatible
// (equal, or the slot's allele is unknown and the value is not the sex-marker sentinel), 2 mismatch.
struct SlotDirect {
    const Slot& d;
    CNF2_HD explicit SlotDirect(const Slot& s) : d(s) {}
    CNF2_HD int    allele(int fa) const { return fa ? d.a1 : d.a0; }
    CNF2_HD double sure(int fa) const { return fa ? d.s1 : d.s0; }
    CNF2_HD double w(int phase) const { return phase_weight(d, phase); }
    CNF2_HD void   match(int v, int fa, double* Bv, double* K, double* C, int* mv) const
    {
        match_affine(v, allele(fa), sure(fa), Bv, K, C, mv);
    }
};

enum { SLOTTAB_DOUBLES = 20 };
// record: [fa*7 + cls*2 + {0,1}] = (Bv, K) of class cls, [fa*7 + 6] = C of class 0; [14], [15] phase weights;
// [16..19] = a0, a1, s0, s1
CNF2_HD void slot_table(const Slot& d, double* t)
{
#pragma unroll
    for (int fa = 0; fa < 2; fa++) {
        const int    mf = fa ? d.a1 : d.a0;
        const double sf = fa ? d.s1 : d.s0;
        const double effms = (mf == 0) ? 1.0 : sf;
        double* q = t + fa * 7;
        q[0] = 1.0 - sf;                              // incoming unknown (cnF2freq.cpp:1203-1210, bound iff mf known)
        q[1] = (mf != 0) ? 0.0 : effms;
        q[6] = (mf != 0) ? effms : 0.0;
        q[2] = 1.0 - sf;                              // compatible
        q[3] = effms;
        q[4] = sf;                                    // mismatch (cnF2freq.cpp:1198-1202)
        q[5] = (sf != 0.0) ? 1.0 - sf : 0.0;
    }
    t[14] = phase_weight(d, 0);
    t[15] = phase_weight(d, 1);
    t[16] = (double)d.a0;
    t[17] = (double)d.a1;
    t[18] = d.s0;
    t[19] = d.s1;
}

struct SlotTable {
    const double* t;
    CNF2_HD explicit SlotTable(const double* p) : t(p) {}
    CNF2_HD int    allele(int fa) const { return (int)t[16 + fa]; }
    CNF2_HD double sure(int fa) const { return t[18 + fa]; }
    CNF2_HD double w(int phase) const { return t[14 + phase]; }
    CNF2_HD void   match(int v, int fa, double* Bv, double* K, double* C, int* mv) const
    {
        const int mf  = allele(fa);
        const int cls = (v == 0) ? 0 : (((mf == 0 && v != 9) || v == mf) ? 1 : 2);
        const double* q = t + fa * 7;
        *Bv = q[cls * 2];
        *K  = q[cls * 2 + 1];
        *C  = (cls == 0) ? q[6] : 0.0;
        *mv = (v == 0) ? mf : v;
    }
};

// Grandparent (top of a line) as seen with incoming allele v.  Per allele index fg the products
//     p0[fg][bit] = w(fg ^ bit) * t0[fg],   p1[fg][bit] = w(fg ^ bit) * t1[fg]
// so that the leaf's value for state bit `bit` and incoming odds sv is sum_fg (p0 + sv * p1), with the
// restrictions applied as 0/1 factors on the fg terms (no selects in the combination):
//     m1  = 0 if the slot is restricted to allele index 0 (flag2ignore), else 1      -> kinds 1, 2: fg = 1 term
//     i2[fg] = 1 if allele fg is 2 (class of the line), else 0                        -> kind 2
// A missing grandparent (recursetrackpossible on a null parent: 1 + secondval, cnF2freq.cpp:1043-1046) is the
// same form with t0 = t1 = (1, 0) and both weights 1; its class flag is the parent's allele (the parent is
// then the top of the traced line, cnF2freq.cpp:1260-1268).
struct Leaf {
    double p0[2][2], p1[2][2];
    double m1, i2[2];
    double mk[2][2];        // TIES: [fg][bit] 0/1 admissibility of allele fg under state bit `bit` (restriction and tie rule)
};

// HOMLEAF: the caller guarantees that the slot is present and homozygous with equal sure at this marker: both
// allele indices match the incoming value alike, so one match is evaluated (its weights are 0/1, the sums over
// fg keep their bits).
// COMPLETE: the caller guarantees a complete window (every slot present, nothing restricted): the slot's presence and
// restriction are compile-time facts.
template <bool HOMLEAF = false, bool TIES = false, bool COMPLETE = false, class View>
CNF2_HD void leaf_make(const View& d, uint32_t flags, int v, bool parent_is2, Leaf* L, int force = -1)
{
    const bool present = COMPLETE || (flags & SLOT_PRESENT) != 0;
    const bool r0      = !COMPLETE && (flags & SLOT_RESTRICT0) != 0;
    const double w0 = present ? d.w(0) : 1.0;
    const double w1 = present ? d.w(1) : 1.0;
    double t0[2], t1[2];
#pragma unroll
    for (int fg = 0; fg < 2; fg++) {
        if (HOMLEAF && fg == 1) {
            t0[1] = t0[0];
            t1[1] = t1[0];
            continue;
        }
        double Bv, K, C;
        int    mv;
        d.match(v, fg, &Bv, &K, &C, &mv);
        const double miss = fg ? 0.0 : 1.0;
        t0[fg] = present ? Bv + C : miss;
        t1[fg] = present ? K : miss;
    }
    L->p0[0][0] = w0 * t0[0];      // fg = 0, bit 0 -> weight of phase 0
    L->p0[0][1] = w1 * t0[0];
    L->p0[1][0] = w1 * t0[1];      // fg = 1, bit 0 -> phase 1
    L->p0[1][1] = w0 * t0[1];
    L->p1[0][0] = w0 * t1[0];
    L->p1[0][1] = w1 * t1[0];
    L->p1[1][0] = w1 * t1[1];
    L->p1[1][1] = w0 * t1[1];
    L->m1    = (present && r0) ? 0.0 : 1.0;
    L->i2[0] = (present ? d.allele(0) == 2 : parent_is2) ? 1.0 : 0.0;
    L->i2[1] = (present && d.allele(1) == 2 && !r0) ? 1.0 : 0.0;
    if (TIES) {
#pragma unroll
        for (int fg = 0; fg < 2; fg++)
#pragma unroll
            for (int bit = 0; bit < 2; bit++)
                L->mk[fg][bit] = (present && ((r0 && fg == 1) || (force >= 0 && ((fg ^ bit) & 1) != force))) ? 0.0 : 1.0;
    }
}

// (V0, V1) of the leaf for state bit `bit`; kind 0 = all alleles, 1 = restricted, 2 = class-2 part of restricted
template <bool TIES = false>
CNF2_HD void leaf_value(const Leaf& L, int bit, int kind, double* v0, double* v1)
{
    if (kind == 0) {
        *v0 = L.p0[0][bit] + L.p0[1][bit];
        *v1 = L.p1[0][bit] + L.p1[1][bit];
    } else if (TIES) {
        const double a = (kind == 2 ? L.i2[0] : 1.0) * L.mk[0][bit], b = (kind == 2 ? L.i2[1] : 1.0) * L.mk[1][bit];
        *v0 = a * L.p0[0][bit] + b * L.p0[1][bit];
        *v1 = a * L.p1[0][bit] + b * L.p1[1][bit];
    } else if (kind == 1) {
        *v0 = L.p0[0][bit] + L.m1 * L.p0[1][bit];
        *v1 = L.p1[0][bit] + L.m1 * L.p1[1][bit];
    } else {
        *v0 = L.i2[0] * L.p0[0][bit] + L.i2[1] * L.p0[1][bit];
        *v1 = L.i2[0] * L.p1[0][bit] + L.i2[1] * L.p1[1][bit];
    }
}

// The 8 entries of one lane.  Output index e = sp*4 + bit_ot*2 + bit_tr (see part_entry_index).
// tot: unrestricted (forward/backward recursion); rtot/two only if CLASSES.
// cw[s0] = root weight c_f(s0) (written by every lane; identical across parts with equal f).
// `out(kind, e, v)` receives entry e of table kind (0 tot, 1 restricted, 2 class-2) as soon as it is known
// (the tile producer stores it straight into LDS: nothing is held in registers until the end).
// HOMPAR: the caller guarantees that the parent is homozygous with equal sure at this marker (SLOT_HOM): the
// terms of its two allele indices are then the same numbers and only fp = 0 is evaluated (the parent's phase
// weights are 0/1, so  wl0 * h + w1 * h  ==  (wl0 + w1) * h  bit for bit).
// NORESTR: the caller guarantees that no slot of the window is restricted (flag2ignore == 0, a complete
// window: all six ancestors present and genotyped, cnf2_window.cpp): the restricted table is the unrestricted one, bit
// for bit, and is copied instead of recomputed; the branches for a root at the top of its lines, a missing or founder
// parent and missing grandparents are not compiled in.
// TIES: the restricted tables (kinds 1, 2) are those of the tie combination c.force_* (line_restricted() with forces).
template <bool CLASSES, bool HOMPAR, bool HOMLEAF, bool NORESTR, bool TIES = false, class VR, class VP, class VT, class VO, class Out>
CNF2_HD void emtab_part_views(const PartCfg& c, const VR& root, const VP& par, const VT& trs, const VO& ots,
                              Out&& out, double cw[2])
{
    static_assert(!TIES || (!HOMPAR && !HOMLEAF && !NORESTR), "tie combinations take the general form");
    constexpr int NFP = HOMPAR ? 1 : 2;
    // ---- root (cnF2freq.cpp:1191-1245 at genwidth 4, inmarkerval unknown)
    const int    mf = root.allele(c.f), mo = root.allele(c.f ^ 1);
    const double sf = root.sure(c.f), so_r = root.sure(c.f ^ 1);
    const double base_r = 1.0 - sf;
    const double msv_r  = (mf != 0) ? sf : 0.0;
    cw[0] = root.w(c.f ^ 0);
    cw[1] = root.w(c.f ^ 1);
    if (!NORESTR && c.root_attop) {
        // the root is the top of its only line: e = sum_f (base + odds) * weight
        const double v = (c.P == 0) ? base_r + msv_r : 1.0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            out(0, e, v);
            if (CLASSES) {
                out(1, e, v);
                out(2, e, (c.P == 0 && mf == 2) ? v : 0.0);
            }
        }
        return;
    }
    if (base_r == 0.0) cw[0] = cw[1] = 0.0;          // !baseval at the root (cnF2freq.cpp:1271)
    const int    inmv = c.P ? mo : mf;
    const double u0 = c.P ? 1.0 - so_r : base_r;      // value = u0 * R0 + u1 * R1
    const double u1 = c.P ? so_r : msv_r;

    const bool par_present = NORESTR || (c.par & SLOT_PRESENT) != 0;
    const bool par_founder = !NORESTR && (c.par & SLOT_FOUNDER) != 0;
    const bool par_r0      = !NORESTR && (c.par & SLOT_RESTRICT0) != 0;
    const bool rootcls     = (c.P == 0 && mf == 2);
    // admissibility of the parent's allele fp in the restricted tables (cnF2freq.cpp:3462-3496)
    double pmk[2] = {1.0, par_r0 ? 0.0 : 1.0};
    if (TIES && c.force_par >= 0) {
        pmk[0] = ((0 ^ c.firstpar) & 1) == c.force_par ? pmk[0] : 0.0;
        pmk[1] = ((1 ^ c.firstpar) & 1) == c.force_par ? pmk[1] : 0.0;
    }

    // ---- parent match terms per parent allele fp
    double alpha[2], beta[2];      // entry = sum_fp wl * OO * (alpha*T0 + beta*T1)
    double pw[2];                  // parent phase weights
    bool   bzero[2], pis2[2];
    int    vtr[2], vot[2];
    double so_p[2];
    pw[0] = par.w(0);
    pw[1] = par.w(1);
#pragma unroll
    for (int fp = 0; fp < 2; fp++) {
        if (HOMPAR && fp == 1) {
            alpha[1] = alpha[0];
            beta[1]  = beta[0];
            bzero[1] = bzero[0];
            pis2[1]  = pis2[0];
            vtr[1]   = vtr[0];
            vot[1]   = vot[0];
            so_p[1]  = so_p[0];
            continue;
        }
        double Bp, Kp, Cp;
        int    mv;
        par.match(inmv, fp, &Bp, &Kp, &Cp, &mv);
        alpha[fp] = u0 * Bp;
        beta[fp]  = u0 * Cp + u1 * Kp;
        bzero[fp] = (Bp == 0.0);
        pis2[fp]  = (par.allele(fp) == 2);
        vtr[fp]   = mv;
        vot[fp]   = par.allele(fp ^ 1);
        so_p[fp]  = par.sure(fp ^ 1);
    }

    if (!par_present || par_founder) {
        // no recursion below the parent: missing (1 + sv) or founder (top of the line)
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
            double v = 0.0, vr = 0.0, v2 = 0.0;
            if (!par_present) {
                v  = u0 + u1;
                vr = v;
                v2 = rootcls ? v : 0.0;
            } else {
#pragma unroll
                for (int fp = 0; fp < 2; fp++) {
                    // founder parent: baseval + odds, then the phase weight (cnF2freq.cpp:1213-1245)
                    const double t = (c.firstpar ? pw[(fp ^ sp) ^ 1] : pw[fp ^ sp]) * (alpha[fp] + beta[fp]);
                    v += t;
                    if (TIES ? pmk[fp] != 0.0 : !(par_r0 && fp == 1)) {
                        vr += t;
                        if (pis2[fp]) v2 += t;
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < 4; b++) {
                out(0, sp * 4 + b, v);
                if (CLASSES) {
                    out(1, sp * 4 + b, vr);
                    out(2, sp * 4 + b, v2);
                }
            }
        }
        return;
    }

    // ---- grandparents: traced (gets the matched allele and its odds) and other.  One at a time, and only
    // their per-kind values are kept (12 + 8 doubles instead of four Leaf records): the producer runs with the
    // whole backward state of the wave live around it, so its own footprint decides what spills.
    const int KINDS = CLASSES ? 3 : 1;
    double G[3][2][2];       // [kind][fp][bit_tr]   traced line: alpha * t0 + beta * t1
    double OO[2][2][2];      // [kind 0/1][fp][bit_ot] other line (never carries the class: kind 2 uses kind 1)
#pragma unroll
    for (int fp = 0; fp < NFP; fp++) {
        Leaf L;
        leaf_make<HOMLEAF, TIES, NORESTR>(trs, c.tr, vtr[fp], pis2[fp], &L, c.force_tr);
#pragma unroll
        for (int kind = 0; kind < KINDS; kind++) {
            if (NORESTR && kind == 1) continue;
#pragma unroll
            for (int bit = 0; bit < 2; bit++) {
                double t0, t1;
                leaf_value<TIES>(L, bit, kind, &t0, &t1);
                G[kind][fp][bit] = alpha[fp] * t0 + beta[fp] * t1;
            }
        }
    }
    CNF2_SCHED_FENCE();
#pragma unroll
    for (int fp = 0; fp < NFP; fp++) {
        Leaf L;
        leaf_make<HOMLEAF, TIES, NORESTR>(ots, c.ot, vot[fp], false, &L, c.force_ot);
#pragma unroll
        for (int kind = 0; kind < ((CLASSES && !NORESTR) ? 2 : 1); kind++)
#pragma unroll
            for (int bit = 0; bit < 2; bit++) {
                double o0, o1;
                leaf_value<TIES>(L, bit, kind, &o0, &o1);
                OO[kind][fp][bit] = (1.0 - so_p[fp]) * o0 + so_p[fp] * o1;
            }
    }
    CNF2_SCHED_FENCE();
    // parent phase weight of allele fp under shift bit sp: pw[fp ^ sp ^ firstpar], zeroed where !baseval
    // (cnF2freq.cpp:1271); wq folds firstpar so that the index below is static
    const double wq0 = c.firstpar ? pw[1] : pw[0], wq1 = c.firstpar ? pw[0] : pw[1];
    double wl[2][2];         // [sp][fp]
    wl[0][0] = (bzero[0] || wq0 == 0.0) ? 0.0 : wq0;
    wl[0][1] = (bzero[1] || wq1 == 0.0) ? 0.0 : wq1;
    wl[1][0] = (bzero[0] || wq1 == 0.0) ? 0.0 : wq1;
    wl[1][1] = (bzero[1] || wq0 == 0.0) ? 0.0 : wq0;
    const double pm1 = par_r0 ? 0.0 : 1.0;     // the parent's own restriction: kinds 1, 2 drop fp = 1
    // Output order: e = sp*4 + bit_ot*2 + bit_tr (the caller maps it to the table index: the traced
    // grandparent is pars[firstpar], so (bit_a, bit_b) = firstpar ? (bit_ot, bit_tr) : (bit_tr, bit_ot))
#pragma unroll
    for (int kind = 0; kind < KINDS; kind++) {
        if (NORESTR && kind == 1) continue;
        const int ko = NORESTR ? 0 : (kind == 2 ? 1 : kind);
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
            const double w0 = (TIES && kind >= 1) ? pmk[0] * wl[sp][0] : wl[sp][0];
            const double w1 = (kind >= 1 && !NORESTR) ? (TIES ? pmk[1] : pm1) * wl[sp][1] : wl[sp][1];
#pragma unroll
            for (int bo = 0; bo < 2; bo++)
#pragma unroll
                for (int bt = 0; bt < 2; bt++) {
                    const double v = HOMPAR ? (w0 + w1) * (OO[ko][0][bo] * G[kind][0][bt])
                                            : w0 * (OO[ko][0][bo] * G[kind][0][bt]) + w1 * (OO[ko][1][bo] * G[kind][1][bt]);
                    out(kind, sp * 4 + bo * 2 + bt, v);
                    if (NORESTR && CLASSES && kind == 0) out(1, sp * 4 + bo * 2 + bt, v);
                }
        }
    }
}

// from raw slot data (every part does its own match logic)
template <bool CLASSES, bool HOMPAR = false, bool HOMLEAF = false, bool NORESTR = false, bool TIES = false, class Out>
CNF2_HD void emtab_part_to(const PartCfg& c, const Slot& root, const Slot& par, const Slot& trs, const Slot& ots,
                           Out&& out, double cw[2])
{
    emtab_part_views<CLASSES, HOMPAR, HOMLEAF, NORESTR, TIES>(c, SlotDirect(root), SlotDirect(par), SlotDirect(trs), SlotDirect(ots), out, cw);
}

// from the 7 slot records of the marker (`recs` = 7 x SLOTTAB_DOUBLES, slot order of the window)
template <bool CLASSES, bool HOMPAR = false, bool HOMLEAF = false, class Out>
CNF2_HD void emtab_part_tables(const PartCfg& c, const double* recs, Out&& out, double cw[2])
{
    const int sp = 1 + 3 * c.P;
    emtab_part_views<CLASSES, HOMPAR, HOMLEAF, false, false>(c, SlotTable(recs), SlotTable(recs + sp * SLOTTAB_DOUBLES),
                              SlotTable(recs + (sp + 1 + c.firstpar) * SLOTTAB_DOUBLES),
                              SlotTable(recs + (sp + 1 + (c.firstpar ^ 1)) * SLOTTAB_DOUBLES), out, cw);
}

// array form (host tests)
template <bool CLASSES>
CNF2_HD void emtab_part(const PartCfg& c, const Slot& root, const Slot& par, const Slot& trs, const Slot& ots,
                        double tot[8], double rtot[8], double two[8], double cw[2])
{
    emtab_part_to<CLASSES>(c, root, par, trs, ots,
                           [&](int kind, int e, double v) { (kind == 0 ? tot : (kind == 1 ? rtot : two))[e] = v; }, cw);
}

// Static part of a lane of the tile producer.  The window's arrays are read with constant indices and the lane's
// slots chosen by selects: an index that depends on the lane would put a copy of the window into scratch memory.
CNF2_HD void make_part(const Window& w, int part, PartCfg* c, int32_t* row_par, int32_t* row_tr, int32_t* row_ot)
{
    const int  P = part >> 2, f = (part >> 1) & 1, firstpar = part & 1;
    const bool hi = P != 0, fp = firstpar != 0;
    c->P = P;
    c->f = f;
    c->firstpar   = firstpar;
    // slot_par = 1 + 3 P, slot_tr = slot_par + 1 + firstpar, slot_ot = slot_par + 1 + (firstpar ^ 1)
    const uint32_t fa = hi ? w.flags[5] : w.flags[2], fb = hi ? w.flags[6] : w.flags[3];
    const int32_t  ra = hi ? w.row[5] : w.row[2], rb = hi ? w.row[6] : w.row[3];
    const int32_t  rp = hi ? w.row[4] : w.row[1];
    c->par        = hi ? w.flags[4] : w.flags[1];
    c->tr         = fp ? fb : fa;
    c->ot         = fp ? fa : fb;
    c->root_attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    const int32_t rt = fp ? rb : ra, ro = fp ? ra : rb;
    *row_par = rp < 0 ? 0 : rp;
    *row_tr  = rt < 0 ? 0 : rt;
    *row_ot  = ro < 0 ? 0 : ro;
}

// the forces of tie combination `combo` for the three slots of a part (cnf2_lane.h tie_force)
CNF2_HD void part_forces(const Window& w, int part, int combo, PartCfg* c)
{
    const bool   hi = (part >> 2) != 0, fp = (part & 1) != 0;
    const int8_t tp = hi ? w.tie[4] : w.tie[1];
    const int8_t ta = hi ? w.tie[5] : w.tie[2], tb = hi ? w.tie[6] : w.tie[3];
    const int8_t tt = fp ? tb : ta, to = fp ? ta : tb;
    c->force_par = tp < 0 ? -1 : ((combo >> tp) & 1);
    c->force_tr  = tt < 0 ? -1 : ((combo >> tt) & 1);
    c->force_ot  = to < 0 ? -1 : ((combo >> to) & 1);
}

// table index of entry e (= sp*4 + bit_ot*2 + bit_tr) of a part: (bit_a, bit_b) = firstpar ? (bit_ot, bit_tr)
// : (bit_tr, bit_ot), k = firstpar | bit_a << 1 | bit_b << 2
CNF2_HD int part_entry_index(int part, int e)
{
    const int P = part >> 2, f = (part >> 1) & 1, firstpar = part & 1;
    const int sp = e >> 2, bo = (e >> 1) & 1, bt = e & 1;
    const int ba = firstpar ? bo : bt, bb = firstpar ? bt : bo;
    return (P << 5) | (f << 4) | (sp << 3) | (firstpar | (ba << 1) | (bb << 2));
}

} // namespace cnf2
#endif
