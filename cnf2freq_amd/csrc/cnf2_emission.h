// cnf2_emission.h -- the pedigree-recursive emission of cnF2freq flattened into a
// fixed-depth, per-lane lookup (host + device code; the host build exists so that the
// arithmetic can be unit-tested on a machine without a GPU).
//
// What it replaces: individ::trackpossible<0,0> / recursetrackpossible / markermiss
// (cnF2freq.cpp:303-329, 955-1359) as called from adjustprobs (cnF2freq.cpp:1602-1632),
// and the class bookkeeping of the zero-propagate call that yields `mapval`
// (cnF2freq.cpp:1260-1268, 5511-5512).
//
// Structure used: the root call has flag = 2g, so the root's allele f goes
// to parent 0 with state bits 0-2 and the other allele to parent 1 with bits 3-5
// (cnF2freq.cpp:1156,1383).  Hence for shift mode s = (s0 | s1<<1 | s2<<2)
//     e_s(g) = sum_f  c_f(s0) * A_f[s1][g & 7] * B_f[s2][g >> 3]
// with A/B the value of the recursion into parent 0/1.  One wavefront lane evaluates one
// table entry (P, f, sp, k): P parent side, f root allele, sp that parent's shift bit,
// k the parent's 3 state bits -- 2*2*2*8 = 64 entries per (individual, marker), shared by
// all 8 shift modes.
//
// Every entry is kept as its terms (parent allele fp x grandparent allele fg) so that the
// per-locus dosage row can re-sum them under path restrictions: flag2ignore (a slot may
// only use allele index 0, cnF2freq.cpp:3478) and the all-or-none rule for an ancestor
// that occupies several window slots (cnF2freq.cpp:3484-3486).
#ifndef CNF2_EMISSION_H
#define CNF2_EMISSION_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CNF2_HD __host__ __device__ __forceinline__
#else
#define CNF2_HD inline
#endif

namespace cnf2 {

// One individual's data at one marker (cnF2freq.cpp:876-887).
struct Slot {
    int    a0, a1;   // MarkerVal: 0 unknown, 1, 2, 9 sex-marker sentinel
    double s0, s1;   // markersure
    double hw;       // haploweight
};

// Static (per job, per lane) description of the line this lane evaluates.
struct LineCfg {
    // bit 0 present, bit 1 founder (stops the recursion, cnF2freq.cpp:1120),
    // bit 2 restrict0 (slot's bit set in flag2ignore)
    uint32_t par, tr, ot;
    int      firstpar;   // k & 1: which grandparent the traced allele came from
    int      bit_tr;     // state bit of the traced grandparent
    int      bit_ot;     // state bit of the other grandparent
    int      sp;         // shift bit of this parent (localshift at genwidth 2)
};

// SLOT_HOM: the slot's genotype row is homozygous (or doubly unknown) with equal sure at EVERY marker
// (derived from the data, not from the pedigree): its two allele indices are then interchangeable
enum { SLOT_PRESENT = 1, SLOT_FOUNDER = 2, SLOT_RESTRICT0 = 4, SLOT_HOM = 8 };

// Terms of one table entry.  For parent allele fp and grandparent allele fg:
//   value = sum_fp base[fp] * (sum_fg ot[fp][fg]) * (sum_fg tr[fp][fg])
// trtwo[fp][fg] = tr[fp][fg] where the allele at the top of the traced line is 2, else 0.
struct LineTerms {
    double base[2];
    double ot[2][2];
    double tr[2][2];
    double trtwo[2][2];
};

// markermiss<false> (cnF2freq.cpp:303-316): returns mismatch, *mv = value after binding.
CNF2_HD bool markermiss(int a, int b, int* mv)
{
    if (a == 0) {
        *mv = b;
        return false;
    }
    *mv = a;
    if (b == 0 && a != 9) return false;
    return a != b;
}

// Match term of trackpossible (cnF2freq.cpp:1191-1210): baseval and mainsecondval.
CNF2_HD void match_term(int inmv, double sv, int mf, double sf, double* baseval, double* msv, int* mv)
{
    if (markermiss(inmv, mf, mv)) {
        *baseval = sf;
        *msv     = (sf != 0.0 && sv != 0.0) ? (1.0 - sf) * sv : 0.0;
    } else {
        double esv = (inmv == 0 && *mv != 0) ? 1.0 : sv;
        *baseval   = 1.0 - sf;
        *msv       = (mf == 0 ? 1.0 : sf) * esv;
    }
}

// Phase weight (cnF2freq.cpp:1227-1245, CORRECTIONINFERENCE == false during sweeps).
CNF2_HD double phase_weight(const Slot& d, int phase)
{
    double p = phase ? 1.0 : 0.0;
    if (d.a0 == d.a1 && d.s0 == d.s1) return p;
    return fabs(p - d.hw);
}

// Ancestor at the top of a line: genwidth == 1 or founder (cnF2freq.cpp:1120,1213-1217).
// out[fg] = value of allele index fg, two[fg] = same if that allele is 2.
CNF2_HD void top_terms(const Slot& d, int inmv, double sv, int phasebit, double out[2], double two[2])
{
#pragma unroll
    for (int f = 0; f < 2; f++) {
        int    mf = f ? d.a1 : d.a0;
        double sf = f ? d.s1 : d.s0;
        double baseval, msv;
        int    mv;
        match_term(inmv, sv, mf, sf, &baseval, &msv, &mv);
        baseval += msv;
        baseval *= phase_weight(d, f ^ phasebit);
        out[f] = baseval;
        two[f] = (mf == 2) ? baseval : 0.0;
    }
}

// recursetrackpossible towards a missing ancestor (cnF2freq.cpp:1043-1046): 1 + secondval,
// carried by allele index 0 so that restrictions never drop it.
CNF2_HD void missing_terms(double sv, double out[2], double two[2])
{
    out[0] = 1.0 + sv;
    out[1] = 0.0;
    two[0] = two[1] = 0.0;
}

// One table entry: the recursion into parent P for incoming allele inmv with error odds sv.
// rootclass2: P == 0 and the root's own allele f is 2 (used when parent 0 is missing and the
// root itself is the top of line 0, cnF2freq.cpp:1260-1268 with firstpar == 0).
CNF2_HD void line_terms(const LineCfg& c, const Slot& par, const Slot& tr, const Slot& ot, int inmv,
                        double sv, bool rootclass2, LineTerms* T)
{
#pragma unroll
    for (int f = 0; f < 2; f++) {
        T->base[f] = 0.0;
        T->ot[f][0] = T->ot[f][1] = 0.0;
        T->tr[f][0] = T->tr[f][1] = 0.0;
        T->trtwo[f][0] = T->trtwo[f][1] = 0.0;
    }
    if (!(c.par & SLOT_PRESENT)) {
        T->base[0]  = 1.0;
        T->ot[0][0] = 1.0;
        T->tr[0][0] = 1.0 + sv;
        T->trtwo[0][0] = rootclass2 ? T->tr[0][0] : 0.0;
        return;
    }
    if (c.par & SLOT_FOUNDER) {
        // the parent is the top of its line: its own alleles play the role of tr
        double o[2], t2[2];
        top_terms(par, inmv, sv, c.firstpar ^ c.sp, o, t2);
#pragma unroll
        for (int f = 0; f < 2; f++) {
            T->base[f]     = 1.0;
            T->ot[f][0]    = 1.0;
            T->tr[f][0]    = o[f];
            T->trtwo[f][0] = t2[f];
        }
        return;
    }
#pragma unroll
    for (int f = 0; f < 2; f++) {
        int    mf = f ? par.a1 : par.a0, mo = f ? par.a0 : par.a1;
        double sf = f ? par.s1 : par.s0, so = f ? par.s0 : par.s1;
        double baseval, msv;
        int    mv;
        match_term(inmv, sv, mf, sf, &baseval, &msv, &mv);
        if (msv != 0.0) msv /= baseval;                                  // cnF2freq.cpp:1220
        baseval *= phase_weight(par, f ^ c.firstpar ^ c.sp);             // cnF2freq.cpp:1227-1245
        double ssv = 0.0;
        if (so != 0.0) {                                                 // cnF2freq.cpp:1298-1302
            baseval *= (1.0 - so);
            ssv = so / (1.0 - so);
        }
        bool dead = !(baseval != 0.0);                                   // cnF2freq.cpp:1271 (!baseval)
        if (dead) {
            baseval = 0.0;
            msv     = 0.0;
            ssv     = 0.0;
        }
        T->base[f] = baseval;
        if (c.ot & SLOT_PRESENT) top_terms(ot, mo, ssv, c.bit_ot, T->ot[f], T->trtwo[f]); // trtwo scratch
        else missing_terms(ssv, T->ot[f], T->trtwo[f]);
        if (c.tr & SLOT_PRESENT) {
            top_terms(tr, mv, msv, c.bit_tr, T->tr[f], T->trtwo[f]);
        } else {
            missing_terms(msv, T->tr[f], T->trtwo[f]);
            // !pars[firstpar]: the parent is the top of the traced line (cnF2freq.cpp:1260-1268)
            T->trtwo[f][0] = (mf == 2) ? T->tr[f][0] : 0.0;
        }
        if (dead) {
            T->ot[f][0] = T->ot[f][1] = 0.0;
            T->tr[f][0] = T->tr[f][1] = 0.0;
            T->trtwo[f][0] = T->trtwo[f][1] = 0.0;
        }
    }
}

// Unrestricted value of the entry (what the forward/backward recursion uses).
CNF2_HD double line_total(const LineTerms& T)
{
    double r = 0.0;
#pragma unroll
    for (int f = 0; f < 2; f++) {
        double b = T.base[f] * (T.ot[f][0] + T.ot[f][1]);
        r += b * (T.tr[f][0] + T.tr[f][1]);
    }
    return r;
}

// allowed(allele index fa of a slot) under flag2ignore and the tie rule.
// force < 0: free; else only alleles with (fa ^ firstpar_of_slot) == force survive.
CNF2_HD bool allele_ok(uint32_t slotflags, int fa, int firstpar_of_slot, int force)
{
    if ((slotflags & SLOT_RESTRICT0) && fa == 1) return false;
    if (force >= 0 && ((fa ^ firstpar_of_slot) & 1) != force) return false;
    return true;
}

// Restricted value (rtot) and its class-2 part (two) for the dosage row.
CNF2_HD void line_restricted(const LineCfg& c, const LineTerms& T, int force_par, int force_tr,
                             int force_ot, double* rtot, double* two)
{
    double r = 0.0, t = 0.0;
    const bool par_is_line = !(c.par & SLOT_PRESENT) || (c.par & SLOT_FOUNDER);
#pragma unroll
    for (int f = 0; f < 2; f++) {
        if ((c.par & SLOT_PRESENT) && !allele_ok(c.par, f, c.firstpar, force_par)) continue;
        double so = 0.0, st = 0.0, s2 = 0.0;
#pragma unroll
        for (int fg = 0; fg < 2; fg++) {
            // for a missing/founder parent and for missing grandparents the single pseudo-term
            // sits at index 0 and is never restricted
            bool ok_ot = par_is_line || !(c.ot & SLOT_PRESENT) ? (fg == 0)
                                                                : allele_ok(c.ot, fg, c.bit_ot, force_ot);
            bool ok_tr = par_is_line || !(c.tr & SLOT_PRESENT) ? (fg == 0)
                                                                : allele_ok(c.tr, fg, c.bit_tr, force_tr);
            if (ok_ot) so += T.ot[f][fg];
            if (ok_tr) {
                st += T.tr[f][fg];
                s2 += T.trtwo[f][fg];
            }
        }
        double b = T.base[f] * so;
        r += b * st;
        t += b * s2;
    }
    *rtot = r;
    *two  = t;
}

// Root-level scalars (cnF2freq.cpp:1191-1245 at genwidth 4 with inmarkerval unknown).
//   inmv/sv handed to parent 0 (allele f) and parent 1 (the other allele), and
//   c_f(s0) = weight of root allele f, including the (1 - sure) of the other allele.
struct RootTerms {
    int    inmv0, inmv1;
    double sv0, sv1;
    double cbase;      // c_f without the phase weight
};

CNF2_HD void root_terms(const Slot& r, bool root_attop, int f, RootTerms* R)
{
    int    mf = f ? r.a1 : r.a0, mo = f ? r.a0 : r.a1;
    double sf = f ? r.s1 : r.s0, so = f ? r.s0 : r.s1;
    double baseval = 1.0 - sf;
    double msv     = (mf != 0) ? sf : 0.0;      // effectivemarkersure * effectivesecondval
    if (root_attop) {
        baseval += msv;
        msv = 0.0;
    } else if (msv != 0.0) {
        msv /= baseval;
    }
    double ssv = 0.0;
    if (!root_attop && so != 0.0) {
        baseval *= (1.0 - so);                  // commutes with the phase weight applied later
        ssv = so / (1.0 - so);
    }
    R->inmv0 = mf;
    R->inmv1 = mo;
    R->sv0   = msv;
    R->sv1   = ssv;
    R->cbase = baseval;
}

} // namespace cnf2
#endif
