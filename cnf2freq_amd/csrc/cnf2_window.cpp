// cnf2_window.cpp -- see cnf2_window.h.
#include "cnf2_window.h"

#include <string.h>

#include "cnf2_emission.h"

namespace cnf2 {

static bool informative_parent(const HostPedigree& P, int lev1i)
{
    // cnF2freq.cpp:3135-3168: a parent counts if it is non-empty or has a non-empty parent
    if (lev1i < 0) return false;
    if (!P.empty[lev1i]) return true;
    for (int lev2 = 0; lev2 < 2; lev2++) {
        int lev2i = P.par[lev1i * 2 + lev2];
        if (lev2i >= 0 && !P.empty[lev2i]) return true;
    }
    return false;
}

void derive_founders(HostPedigree& P)
{
    P.founder.assign(P.n_rec, 0);
    for (int r = 0; r < P.n_rec; r++) {
        bool anylev1 = informative_parent(P, P.par[r * 2]) || informative_parent(P, P.par[r * 2 + 1]);
        P.founder[r] = anylev1 ? 0 : 1;
    }
}

void derive_descendants(const HostPedigree& P, int32_t* desc)
{
    std::vector<int32_t> upsent(P.n_rec, 0);
    for (int r = 0; r < P.n_rec; r++) desc[r] = 0;
    bool any;
    do {
        any = false;
        for (int r = 0; r < P.n_rec; r++) {
            int now = desc[r] ? desc[r] : 1;
            now -= upsent[r];
            if (now > 0) {
                for (int k = 0; k < 2; k++)
                    if (P.par[r * 2 + k] >= 0) desc[P.par[r * 2 + k]] += now;
                upsent[r] += now;
                any = true;
            }
        }
    } while (any);
    for (int r = 0; r < P.n_rec; r++)
        if (!desc[r]) desc[r] = 1;
}

void derive_window(const HostPedigree& P, int rec, Window* w, int32_t* slot_rec_out)
{
    memset(w, 0, sizeof(*w));
    int32_t slot_rec[7];
    for (int i = 0; i < 7; i++) {
        slot_rec[i] = -1;
        w->row[i]   = -1;
        w->tie[i]   = -1;
    }
    slot_rec[0] = rec;
    int flag2ignore = 1, shiftignore = 0;                    // cnF2freq.cpp:3117-3118
    bool anylev1 = false;
    bool in_relmap[7] = {true, false, false, false, false, false, false}; // cnF2freq.cpp:3112
    for (int lev1 = 0; lev1 < 2; lev1++) {
        int lev1i = P.par[rec * 2 + lev1];
        if (lev1i < 0) continue;
        int flag2index = 1 + lev1 * 3;                        // cnF2freq.cpp:3124
        int shiftval   = 2 << lev1;                           // cnF2freq.cpp:3125
        slot_rec[flag2index] = lev1i;
        if (!P.empty[lev1i]) {                                // cnF2freq.cpp:3127-3133
            flag2ignore |= 1 << flag2index;
            in_relmap[flag2index] = true;
        }
        bool anypars = false;
        for (int lev2 = 0; lev2 < 2; lev2++) {                // cnF2freq.cpp:3139-3153
            int lev2i = P.par[lev1i * 2 + lev2];
            if (lev2i < 0) continue;
            slot_rec[flag2index + lev2 + 1] = lev2i;
            if (!P.empty[lev2i]) {
                flag2ignore |= 1 << (flag2index + lev2 + 1);
                in_relmap[flag2index + lev2 + 1] = true;
                anypars = true;
            }
        }
        if (anypars) shiftignore |= shiftval;                 // cnF2freq.cpp:3156-3159
        if (anypars || !P.empty[lev1i]) anylev1 = true;       // cnF2freq.cpp:3165-3168
    }
    if (anylev1) shiftignore |= 1;                            // cnF2freq.cpp:3170-3173
    flag2ignore ^= 127;                                       // cnF2freq.cpp:3178-3179
    shiftignore ^= 7;

    w->rec         = rec;
    w->shiftignore = (uint8_t)shiftignore;
    w->flag2ignore = (uint8_t)flag2ignore;
    w->shiftend    = (P.gen[rec] < 2) ? 2 : 8;                // cnF2freq.cpp:5359
    for (int i = 0; i < 7; i++) {
        int r = slot_rec[i];
        if (r < 0) continue;
        w->row[i]   = P.row_of[r];
        uint8_t f   = SLOT_PRESENT;
        if (P.founder[r]) f |= SLOT_FOUNDER;
        if ((flag2ignore >> i) & 1) f |= SLOT_RESTRICT0;
        if (!P.row_hom.empty() && P.row_hom[P.row_of[r]]) f |= SLOT_HOM;
        w->flags[i] = f;
    }
    // relmap groups: one ancestor in several (non-ignored) slots (cnF2freq.cpp:3130,3147)
    int ng = 0;
    for (int i = 1; i < 7; i++) {
        if (!in_relmap[i] || w->tie[i] >= 0) continue;
        bool multi = false;
        for (int j = i + 1; j < 7; j++)
            if (in_relmap[j] && slot_rec[j] == slot_rec[i]) multi = true;
        // an ancestor that is also the individual itself cannot occur (slot 0 is the child)
        if (!multi) continue;
        // An ancestor whose two alleles are equal (or both unknown) with equal sure at every marker
        // can never activate the all-or-none rule: the only paths it would prune have weight 0
        // (cnF2freq.cpp:1235-1239 and 3488).  Such windows stay on the fast kernel.
        // Only for grandparent slots: their localshift is 0 in every slot (cnF2freq.cpp:986), so the
        // forced phase is the same nonzero-weight allele everywhere; a parent in both parent slots
        // sees two different shift bits and keeps its group.
        bool gp_only = true;
        for (int j = i; j < 7; j++)
            if (in_relmap[j] && slot_rec[j] == slot_rec[i] && (j == 1 || j == 4)) gp_only = false;
        if (gp_only && !P.row_hom.empty() && P.row_hom[P.row_of[slot_rec[i]]]) continue;
        for (int j = i; j < 7; j++)
            if (in_relmap[j] && slot_rec[j] == slot_rec[i]) w->tie[j] = (int8_t)ng;
        ng++;
    }
    w->n_groups = (uint8_t)ng;
    if (slot_rec_out) memcpy(slot_rec_out, slot_rec, sizeof(slot_rec));
}

} // namespace cnf2
