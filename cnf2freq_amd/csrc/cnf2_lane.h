// cnf2_lane.h -- mapping of the 64 emission-table entries of one (individual, marker) onto
// the 64 lanes of a wavefront, shared by the HIP kernels and the host unit-test shim.
//   lane = P<<5 | f<<4 | sp<<3 | k
//     P  parent side (0: state bits 0-2, 1: state bits 3-5)
//     f  root allele index handed to parent 0 (the other one goes to parent 1)
//     sp shift bit of that parent (shift bit 1 for P=0, bit 2 for P=1; cnF2freq.cpp:986)
//     k  the parent's 3 state bits: bit0 = which grandparent is traced, bits 1-2 the
//        grandparents' own bits (upflagit, cnF2freq.cpp:321-329,984)
#ifndef CNF2_LANE_H
#define CNF2_LANE_H

#include "cnf2_emission.h"
#include "cnf2_window.h"

namespace cnf2 {

struct LaneJob {
    LineCfg cfg;
    int32_t row_par, row_tr, row_ot;   // genotype rows (blank row 0 when the slot is missing)
    int8_t  tie_par, tie_tr, tie_ot;   // tie group per slot or -1
    int     P, f;
};

CNF2_HD void make_lane(const Window& w, int lane, LaneJob* L)
{
    const int P = lane >> 5, f = (lane >> 4) & 1, sp = (lane >> 3) & 1, k = lane & 7;
    const int firstpar = k & 1, upflag = k >> 1;
    const int slot_par = 1 + 3 * P;
    const int slot_tr  = slot_par + 1 + firstpar;
    const int slot_ot  = slot_par + 1 + (firstpar ^ 1);
    L->P = P;
    L->f = f;
    L->cfg.par      = w.flags[slot_par];
    L->cfg.tr       = w.flags[slot_tr];
    L->cfg.ot       = w.flags[slot_ot];
    L->cfg.firstpar = firstpar;
    L->cfg.bit_tr   = (upflag >> firstpar) & 1;
    L->cfg.bit_ot   = (upflag >> (firstpar ^ 1)) & 1;
    L->cfg.sp       = sp;
    L->row_par = w.row[slot_par] < 0 ? 0 : w.row[slot_par];
    L->row_tr  = w.row[slot_tr] < 0 ? 0 : w.row[slot_tr];
    L->row_ot  = w.row[slot_ot] < 0 ? 0 : w.row[slot_ot];
    L->tie_par = w.tie[slot_par];
    L->tie_tr  = w.tie[slot_tr];
    L->tie_ot  = w.tie[slot_ot];
}

CNF2_HD int tie_force(int8_t tie, int combo)
{
    return tie < 0 ? -1 : ((combo >> tie) & 1);
}

CNF2_HD Slot unpack_slot(uint8_t allele_pair, double s0, double s1, double hw)
{
    Slot d;
    d.a0 = allele_pair & 15;
    d.a1 = allele_pair >> 4;
    d.s0 = s0;
    d.s1 = s1;
    d.hw = hw;
    return d;
}

} // namespace cnf2
#endif
