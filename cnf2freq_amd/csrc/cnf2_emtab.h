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

// Grandparent (top of a line) as seen with incoming allele v:  per allele index fg the pair
// (t0, t1) with value = sum_fg w(fg ^ bit) * (t0[fg] + sv * t1[fg]).
struct Leaf {
    double t0[2], t1[2];
    bool   is2[2];    // allele fg of this ancestor is 2 (class of the line)
    double w[2];      // phase weight for phase 0 / 1
    bool   present;
    bool   restrict0;
};

CNF2_HD void leaf_prepare(const Slot& d, uint32_t flags, Leaf* L)
{
    L->present   = (flags & SLOT_PRESENT) != 0;
    L->restrict0 = (flags & SLOT_RESTRICT0) != 0;
    L->w[0] = phase_weight(d, 0);
    L->w[1] = phase_weight(d, 1);
    L->is2[0] = d.a0 == 2;
    L->is2[1] = d.a1 == 2;
}

CNF2_HD void leaf_match(const Slot& d, int v, Leaf* L)
{
#pragma unroll
    for (int fg = 0; fg < 2; fg++) {
        double Bv, K, C;
        int    mv;
        match_affine(v, fg ? d.a1 : d.a0, fg ? d.s1 : d.s0, &Bv, &K, &C, &mv);
        L->t0[fg] = Bv + C;
        L->t1[fg] = K;
    }
}

// (V0, V1) of the leaf for state bit `bit`; kind 0 = all alleles, 1 = restricted, 2 = class-2 part
// of restricted.  parent_is2: used when the grandparent is missing and the parent is the top of
// the traced line (cnF2freq.cpp:1260-1268).
CNF2_HD void leaf_value(const Leaf& L, int bit, int kind, bool parent_is2, double* v0, double* v1)
{
    if (!L.present) {                      // recursetrackpossible on a null parent: 1 + secondval
        const double z = (kind == 2 && !parent_is2) ? 0.0 : 1.0;
        *v0 = z;
        *v1 = z;
        return;
    }
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int fg = 0; fg < 2; fg++) {
        bool ok = true;
        if (kind >= 1 && L.restrict0 && fg == 1) ok = false;
        if (kind == 2 && !L.is2[fg]) ok = false;
        const double w = ok ? L.w[fg ^ bit] : 0.0;
        a0 += w * L.t0[fg];
        a1 += w * L.t1[fg];
    }
    *v0 = a0;
    *v1 = a1;
}

// The 8 entries of one lane.  Output index e = sp*4 + bit_a + 2*bit_b.
// tot: unrestricted (forward/backward recursion); rtot/two only if CLASSES.
// cw[s0] = root weight c_f(s0) (written by every lane; identical across parts with equal f).
template <bool CLASSES>
CNF2_HD void emtab_part(const PartCfg& c, const Slot& root, const Slot& par, const Slot& trs, const Slot& ots,
                        double tot[8], double rtot[8], double two[8], double cw[2])
{
    // ---- root (cnF2freq.cpp:1191-1245 at genwidth 4, inmarkerval unknown)
    const int    mf = c.f ? root.a1 : root.a0, mo = c.f ? root.a0 : root.a1;
    const double sf = c.f ? root.s1 : root.s0, so_r = c.f ? root.s0 : root.s1;
    const double base_r = 1.0 - sf;
    const double msv_r  = (mf != 0) ? sf : 0.0;
    cw[0] = phase_weight(root, c.f ^ 0);
    cw[1] = phase_weight(root, c.f ^ 1);
    if (c.root_attop) {
        // the root is the top of its only line: e = sum_f (base + odds) * weight
        const double v = (c.P == 0) ? base_r + msv_r : 1.0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            tot[e] = v;
            if (CLASSES) {
                rtot[e] = v;
                two[e]  = (c.P == 0 && mf == 2) ? v : 0.0;
            }
        }
        return;
    }
    if (base_r == 0.0) cw[0] = cw[1] = 0.0;          // !baseval at the root (cnF2freq.cpp:1271)
    const int    inmv = c.P ? mo : mf;
    const double u0 = c.P ? 1.0 - so_r : base_r;      // value = u0 * R0 + u1 * R1
    const double u1 = c.P ? so_r : msv_r;

    const bool par_present = (c.par & SLOT_PRESENT) != 0;
    const bool par_founder = (c.par & SLOT_FOUNDER) != 0;
    const bool par_r0      = (c.par & SLOT_RESTRICT0) != 0;
    const bool rootcls     = (c.P == 0 && mf == 2);

    // ---- parent match terms per parent allele fp
    double alpha[2], beta[2];      // entry = sum_fp wl * OO * (alpha*T0 + beta*T1)
    double pw[2];                  // parent phase weights
    bool   bzero[2], pis2[2];
    int    vtr[2], vot[2];
    double so_p[2];
    pw[0] = phase_weight(par, 0);
    pw[1] = phase_weight(par, 1);
#pragma unroll
    for (int fp = 0; fp < 2; fp++) {
        double Bp, Kp, Cp;
        int    mv;
        const int    mfp = fp ? par.a1 : par.a0;
        const double sfp = fp ? par.s1 : par.s0;
        match_affine(inmv, mfp, sfp, &Bp, &Kp, &Cp, &mv);
        alpha[fp] = u0 * Bp;
        beta[fp]  = u0 * Cp + u1 * Kp;
        bzero[fp] = (Bp == 0.0);
        pis2[fp]  = (mfp == 2);
        vtr[fp]   = mv;
        vot[fp]   = fp ? par.a0 : par.a1;
        so_p[fp]  = fp ? par.s0 : par.s1;
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
                    if (!(par_r0 && fp == 1)) {
                        vr += t;
                        if (pis2[fp]) v2 += t;
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < 4; b++) {
                tot[sp * 4 + b] = v;
                if (CLASSES) {
                    rtot[sp * 4 + b] = vr;
                    two[sp * 4 + b]  = v2;
                }
            }
        }
        return;
    }

    // ---- grandparents: traced (gets the matched allele and its odds) and other
    Leaf Ltr[2], Lot[2];     // per parent allele fp
#pragma unroll
    for (int fp = 0; fp < 2; fp++) {
        leaf_prepare(trs, c.tr, &Ltr[fp]);
        leaf_prepare(ots, c.ot, &Lot[fp]);
        leaf_match(trs, vtr[fp], &Ltr[fp]);
        leaf_match(ots, vot[fp], &Lot[fp]);
    }
    double wl[2][2];         // [sp][fp]: parent phase weight, zeroed where !baseval (cnF2freq.cpp:1271)
#pragma unroll
    for (int sp = 0; sp < 2; sp++)
#pragma unroll
        for (int fp = 0; fp < 2; fp++) {
            const double w = c.firstpar ? pw[(fp ^ sp) ^ 1] : pw[fp ^ sp];   // no dynamic register indexing
            wl[sp][fp] = (bzero[fp] || w == 0.0) ? 0.0 : w;
        }
    const int KINDS = CLASSES ? 3 : 1;
    // one kind at a time (0 tot, 1 restricted, 2 class-2 part): keeps few values live
#pragma unroll
    for (int kind = 0; kind < KINDS; kind++) {
        double H[2][2][2];   // [fp][bit_ot][bit_tr]
#pragma unroll
        for (int fp = 0; fp < 2; fp++) {
            double G[2], OO[2];
#pragma unroll
            for (int bit = 0; bit < 2; bit++) {
                double t0, t1, o0, o1;
                leaf_value(Ltr[fp], bit, kind, pis2[fp], &t0, &t1);
                // the other line never carries the class: kinds 1 and 2 use its restricted total
                leaf_value(Lot[fp], bit, kind == 2 ? 1 : kind, false, &o0, &o1);
                G[bit]  = alpha[fp] * t0 + beta[fp] * t1;
                OO[bit] = (1.0 - so_p[fp]) * o0 + so_p[fp] * o1;
            }
#pragma unroll
            for (int bo = 0; bo < 2; bo++)
#pragma unroll
                for (int bt = 0; bt < 2; bt++) H[fp][bo][bt] = OO[bo] * G[bt];
        }
        double* out = kind == 0 ? tot : (kind == 1 ? rtot : two);
#pragma unroll
        for (int sp = 0; sp < 2; sp++)
#pragma unroll
            for (int bb = 0; bb < 2; bb++)
#pragma unroll
                for (int ba = 0; ba < 2; ba++) {
                    // traced grandparent = pars[firstpar]: (bit_ot, bit_tr) = firstpar ? (ba, bb) : (bb, ba);
                    // selects between statically indexed registers (no dynamic indexing => no scratch)
                    const double h0 = c.firstpar ? H[0][ba][bb] : H[0][bb][ba];
                    const double h1 = c.firstpar ? H[1][ba][bb] : H[1][bb][ba];
                    const double w1 = (kind >= 1 && par_r0) ? 0.0 : wl[sp][1];
                    out[sp * 4 + ba + 2 * bb] = wl[sp][0] * h0 + w1 * h1;
                }
    }
}

// Static part of a lane of the tile producer.
CNF2_HD void make_part(const Window& w, int part, PartCfg* c, int32_t* row_par, int32_t* row_tr, int32_t* row_ot)
{
    const int P = part >> 2, f = (part >> 1) & 1, firstpar = part & 1;
    const int slot_par = 1 + 3 * P;
    const int slot_tr  = slot_par + 1 + firstpar;
    const int slot_ot  = slot_par + 1 + (firstpar ^ 1);
    c->P = P;
    c->f = f;
    c->firstpar   = firstpar;
    c->par        = w.flags[slot_par];
    c->tr         = w.flags[slot_tr];
    c->ot         = w.flags[slot_ot];
    c->root_attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    *row_par = w.row[slot_par] < 0 ? 0 : w.row[slot_par];
    *row_tr  = w.row[slot_tr] < 0 ? 0 : w.row[slot_tr];
    *row_ot  = w.row[slot_ot] < 0 ? 0 : w.row[slot_ot];
}

// table index of entry e (= sp*4 + bit_a + 2*bit_b) of a part
CNF2_HD int part_entry_index(int part, int e)
{
    const int P = part >> 2, f = (part >> 1) & 1, firstpar = part & 1;
    const int sp = e >> 2, ba = e & 1, bb = (e >> 1) & 1;
    return (P << 5) | (f << 4) | (sp << 3) | (firstpar | (ba << 1) | (bb << 2));
}

} // namespace cnf2
#endif
