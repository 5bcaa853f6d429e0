// cnf2_window.h -- host-side derivation of the 3-generation window of an analysed
// individual: what fixtrees (cnF2freq.cpp:3099-3187) leaves in shiftignore, flag2ignore,
// reltreeordered, relmap and individ::founder, flattened into a 64-byte record that the
// kernels read.  Plain C++ (no HIP) so that it is unit-testable without a GPU.
#ifndef CNF2_WINDOW_H
#define CNF2_WINDOW_H

#include <stdint.h>
#include <vector>

namespace cnf2 {

// Slot order is the bit order of flag2 / reltreeordered (cnF2freq.cpp:3124,3146):
// 0 self, 1 parent0, 2 gp00, 3 gp01, 4 parent1, 5 gp10, 6 gp11.
struct Window {
    int32_t row[7];       // genotype row of the slot's individual, -1 if there is none
    uint8_t flags[7];     // SLOT_PRESENT | SLOT_FOUNDER | SLOT_RESTRICT0 (cnf2_emission.h)
    int8_t  tie[7];       // index of the multi-slot ancestor group of this slot, -1 if single
    uint8_t shiftignore;  // cnF2freq.cpp:3158-3179
    uint8_t shiftend;     // 8, or 2 when gen < 2 (cnF2freq.cpp:5359)
    uint8_t n_groups;     // number of tie groups (0..3)
    uint8_t flag2ignore;  // cnF2freq.cpp:3117-3178
    int32_t rec;          // record index of the individual
    uint8_t pad[10];
};
static_assert(sizeof(Window) == 64, "Window must stay 64 bytes");

struct HostPedigree {
    int                  n_rec = 0;
    std::vector<int32_t> par;      // [n_rec][2]
    std::vector<uint8_t> empty;    // [n_rec]
    std::vector<int32_t> gen;      // [n_rec]
    std::vector<int32_t> row_of;   // [n_rec]
    std::vector<uint8_t> founder;  // [n_rec] derived
    std::vector<uint8_t> row_hom;  // [n_rows] optional: row is homozygous with equal sure at EVERY marker
    std::vector<int32_t> dous;     // analysed records
};

// individ::founder for every record: fixtrees sets it when no parent is non-empty or has a
// non-empty parent (cnF2freq.cpp:3119-3177); postmarkerdata runs fixtrees on everybody
// (cnF2freq.cpp:3373-3389) before the first sweep.
void derive_founders(HostPedigree& P);

// individ::descendants as postmarkerdata leaves it (cnF2freq.cpp:3224-3255): every individual sends
// max(1, own count) to both parents until nothing changes; zeros become 1.  desc_out[n_rec].
void derive_descendants(const HostPedigree& P, int32_t* desc_out);

// Window of one analysed record.  slot_rec_out (optional, 7 ints) receives the record per slot.
void derive_window(const HostPedigree& P, int rec, Window* w, int32_t* slot_rec_out);

} // namespace cnf2
#endif
