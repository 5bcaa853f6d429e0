// cnf2_kernels.hip -- gfx950 kernels of the cnF2freq forward-backward sweep.
//
// One wavefront = one job (analysed individual x chromosome), all 8 shift modes at once:
//     lane = chain<<3 | lo      chain = shift mode s (cnF2freq.cpp:5375), lo = state bits 0-2
//     register j = 0..7         state bits 3-5            state g = j*8 + lo
// so the 8 x 64 state values of the 8 concurrent HMM chains sit in 8 VGPR pairs per lane.
//   * emission (adjustprobs + trackpossible, cnF2freq.cpp:1579-1670, 1075-1359): the same 64 lanes
//     first act as the 64 entries of the rank-2 emission table of this (individual, marker)
//     (cnf2_emission.h / cnf2_lane.h), hand them over through 512 B of LDS, then every lane
//     forms e(g) = sum_f c_f A_f[lo] B_f[j] for its 8 states;
//   * transition (realanalyze, cnF2freq.cpp:2273-2367): the 64x64 XOR-indexed matrix is the
//     6-fold Kronecker product of [[1-r, r],[r, 1-r]]; bits 3-5 are register pairs, bits 0-2
//     are DPP lane exchanges inside a row -- no LDS, no MFMA;
//   * scaling (cnF2freq.cpp:1656-1669): per-chain sum by 3 DPP steps; the 8 chains' reciprocals
//     are computed by one instruction stream (8 lanes each); log scales are carried as
//     mantissa x 2^exponent and only turned into a logarithm once per chromosome.
// The forward pass spills alpha-minus (before emission) in register-major order, 8 coalesced
// 512 B stores per marker; the backward pass of the same wave reads it back, forms the per-locus
// allele-2 dosage row (genotypereporter, cnF2freq.cpp:3532-3538, closed form of the fan-out of
// cnF2freq.cpp:5416-5553) and carries beta.
#include <hip/hip_runtime.h>

#include <string.h>
#include <type_traits>

#include "cnf2_device.h"
#include "cnf2_lane.h"
#include "cnf2_emtab.h"
#include "cnf2_accum.h"
#include "cnf2_acctab.h"
#include "cnf2_accpath.h"
#include "cnf2_variance.h"

namespace cnf2 {

// ------------------------------------------------------------------ lane exchange helpers
// full-wave DPP move without a tied "old" operand (every lane is written, so no copy is needed)
template <int CTRL>
__device__ __forceinline__ double dpp_mov_all(double v)
{
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// lane exchange through the LDS crossbar (no LDS memory is touched): BitMode swizzle, xor mask K
template <int K>
__device__ __forceinline__ double swizzle_xor(double v)
{
    int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), (K << 10) | 0x1F);
    int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), (K << 10) | 0x1F);
    return __hiloint2double(hi, lo);
}

// value of lane (id ^ 1), (id ^ 2), (id ^ 4)
__device__ __forceinline__ double lane_xor1(double v) { return dpp_mov_all<0xB1>(v); } // quad_perm [1,0,3,2]
__device__ __forceinline__ double lane_xor2(double v) { return dpp_mov_all<0x4E>(v); } // quad_perm [2,3,0,1]
// xor 4 has no single DPP form on gfx950 (row_shl/row_shr with bank masks costs two tied moves per
// dword plus hazard nops); the LDS crossbar does it in one ds_swizzle per dword and runs beside the VALU
__device__ __forceinline__ double lane_xor4(double v) { return swizzle_xor<4>(v); }
// State bit 2 does not have to live in lane bit 2.  With lane(lo) = b0 ^ 2 b1 ^ 7 b2 (an invertible
// GF(2) map of the three low state bits) flipping b2 is "lane i <-> lane 7-i" inside each group of 8 =
// DPP row_half_mirror, a plain VALU move like the two quad_perm stages: no trip through the LDS pipe.
// CNF2_XCHG (A/B timing): 0 = all three lane stages on DPP; 1 = bit 2, 2 = bits 1 and 2 through the LDS
// crossbar (ds_swizzle: fewer VALU slots, longer latency).
#ifndef CNF2_XCHG
#define CNF2_XCHG 0
#endif
__device__ __forceinline__ int state_lo(int lane) { return (lane & 7) ^ ((lane & 4) ? 3 : 0); }
#if CNF2_XCHG == 0
__device__ __forceinline__ double lane_flip_b1(double v) { return lane_xor2(v); }
__device__ __forceinline__ double lane_flip_b2(double v) { return dpp_mov_all<0x141>(v); }
#elif CNF2_XCHG == 1
__device__ __forceinline__ double lane_flip_b1(double v) { return lane_xor2(v); }
__device__ __forceinline__ double lane_flip_b2(double v) { return swizzle_xor<7>(v); }
#else
__device__ __forceinline__ double lane_flip_b1(double v) { return swizzle_xor<2>(v); }
__device__ __forceinline__ double lane_flip_b2(double v) { return swizzle_xor<7>(v); }
#endif
__device__ __forceinline__ double lane_xor8(double v) { return dpp_mov_all<0x128>(v); } // row_ror:8
__device__ __forceinline__ double lane_xor16(double v) { return swizzle_xor<16>(v); }
// lane id of a full wavefront, recomputed where it is asked for (volatile: not hoisted out of a job loop, where it would
// occupy a register -- or a scratch slot -- for the whole kernel)
__device__ __forceinline__ int fresh_lane()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ double lane_xor32(double v)
{
    const int a  = (fresh_lane() ^ 32) << 2;
    const int lo = __builtin_amdgcn_ds_bpermute(a, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(a, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// sum over the 8 lanes of a chain (lanes differing in bits 0-2); result in every lane.
// Third step pairs lane i with lane 7-i of its group of 8 (row_half_mirror): after the two quad
// steps every lane of a quad holds the quad total, so any quad<->quad pairing finishes the sum,
// and this one is a plain DPP move (no trip through the LDS crossbar on the critical path).
__device__ __forceinline__ double chain_sum(double v)
{
    v += lane_xor1(v);
    v += lane_xor2(v);
    v += dpp_mov_all<0x141>(v);
    return v;
}
// sum over the 8 chains of values that are already uniform inside each chain
__device__ __forceinline__ double across_chains_sum(double v)
{
    v += lane_xor8(v);
    v += lane_xor16(v);
    v += lane_xor32(v);
    return v;
}
__device__ __forceinline__ double across_chains_max(double v)
{
    v = fmax(v, lane_xor8(v));
    v = fmax(v, lane_xor16(v));
    v = fmax(v, lane_xor32(v));
    return v;
}

// ------------------------------------------------------------------ transition
// One Kronecker factor on a register pair / on a lane pair: x' = (1-r) x + r partner
// (recombprec of cnF2freq.cpp:2329-2340 factorised over the 6 state bits).
__device__ __forceinline__ void transition(double (&a)[8], double r0, double r1)
{
    // bit t uses genrec[TYPEGENS[t]], TYPEGENS = {1,0,0,1,0,0} (settings.h:23)
    const double k0 = 1.0 - r0, k1 = 1.0 - r1;
    // bits 0..2: lanes.  All partner values of a stage are requested before any is used, so the
    // eight exchanges overlap (the xor-4 stage goes through the LDS crossbar: one latency, not eight).
    double q[8];
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = lane_xor1(a[j]);
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = k1 * a[j] + r1 * q[j];
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = lane_flip_b1(a[j]);
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = k0 * a[j] + r0 * q[j];
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = lane_flip_b2(a[j]);
#if CNF2_XCHG >= 1
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = k0 * a[j] + r0 * q[j];
    // bits 3..5: registers
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        double x = a[j], y = a[j + 1];
        a[j]     = k1 * x + r1 * y;
        a[j + 1] = k1 * y + r1 * x;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j & 2) continue;
        double x = a[j], y = a[j + 2];
        a[j]     = k0 * x + r0 * y;
        a[j + 2] = k0 * y + r0 * x;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double x = a[j], y = a[j + 4];
        a[j]     = k0 * x + r0 * y;
        a[j + 4] = k0 * y + r0 * x;
    }
}

// Same operator up to a per-gap constant: with t = r / (1 - r) every butterfly is x' = x + t * partner
// and the dropped factor (1-r0)^4 (1-r1)^2 is a scalar common to all states and shift modes, so it
// only moves the normaliser of the step.  The host adds its logarithm, summed over the chromosome,
// to the reported log-likelihoods; ratios (the per-locus rows) never see it.  One fused multiply-add
// per element and stage instead of a multiply and an FMA.
__device__ __forceinline__ void transition_scaled(double (&a)[8], double t0, double t1)
{
    double q[8];
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = lane_xor1(a[j]);
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = fma(t1, q[j], a[j]);
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = lane_flip_b1(a[j]);
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = fma(t0, q[j], a[j]);
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = lane_flip_b2(a[j]);
#if CNF2_XCHG >= 1
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = fma(t0, q[j], a[j]);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const double x = a[j], y = a[j + 1];
        a[j]     = fma(t1, y, x);
        a[j + 1] = fma(t1, x, y);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j & 2) continue;
        const double x = a[j], y = a[j + 2];
        a[j]     = fma(t0, y, x);
        a[j + 2] = fma(t0, x, y);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const double x = a[j], y = a[j + 4];
        a[j]     = fma(t0, y, x);
        a[j + 4] = fma(t0, x, y);
    }
}

// The three butterflies on the state bits that currently sit in the registers (bit 0 of the triple belongs to a
// parent's meiosis: genrec[1], the other two to grandparental ones: genrec[0]; settings.h:23).
__device__ __forceinline__ void register_stages(double (&a)[8], double t0, double t1)
{
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const double x = a[j], y = a[j + 1];
        a[j]     = fma(t1, y, x);
        a[j + 1] = fma(t1, x, y);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j & 2) continue;
        const double x = a[j], y = a[j + 2];
        a[j]     = fma(t0, y, x);
        a[j + 2] = fma(t0, x, y);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const double x = a[j], y = a[j + 4];
        a[j]     = fma(t0, y, x);
        a[j + 4] = fma(t0, x, y);
    }
}

// Transposing form of transition_scaled (CNF2_XPOSE variant of the fast kernel): instead of exchanging the three
// lane-held state bits by DPP (48 v_mov_dpp per transition, a quarter of the kernel's VALU work), the wave swaps
// which triple of state bits sits in the lanes: register stages on the triple held in registers, an 8 x 8 transpose
// of every chain's (lane, register) block through LDS (8 ds_write_b64 + 8 ds_read_b64, no VALU slot), register
// stages on the other triple.  The layout therefore alternates from marker to marker: even local markers keep state
// bits 0-2 in the lanes (as the spill rows and the table reads of the DPP variant assume), odd ones bits 3-5.
// xb: this wave's 64 x XPOSE_RS doubles.  Row stride 9 spreads the eight chains over the LDS banks.
#define XPOSE_RS 9
__device__ __forceinline__ void transition_xpose(double (&a)[8], double t0, double t1, double* xb, int lane)
{
    register_stages(a, t0, t1);
    const int s8 = lane & 56, l = lane & 7;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // earlier reads of xb are done
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; j++) xb[(s8 + j) * XPOSE_RS + l] = a[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = xb[(s8 + l) * XPOSE_RS + j];
    register_stages(a, t0, t1);
}

// ------------------------------------------------------------------ per-job lane state
struct LaneCtx {
    LaneJob L;            // producer role
    int     row_root;
    bool    root_attop;
    int     s0, s1, s2;   // consumer role: bits of this lane's shift mode
    int     lo;
    bool    active;       // chain takes part (not masked by shiftignore / shiftend)
    int     n_combo;      // 1 << n_groups (1 with CNF2_NO_TIES)
};

__device__ __forceinline__ Slot load_slot(const KernelParams& p, int row, int m)
{
    size_t  i  = (size_t)row * p.n_markers + m;
    uint8_t ap = p.allele8[i];
    double2 s  = p.sure[i];
    return unpack_slot(ap, s.x, s.y, p.hw[i]);
}

// Producer half: this lane's table entry for marker m.
// Returns the unrestricted value; if WITH_CLASSES also the restricted value and its class-2
// part for tie combination `combo` (LineTerms kept by the caller).
__device__ __forceinline__ double produce_entry(const KernelParams& p, const LaneCtx& c, const Slot& root, int m,
                                                LineTerms* T)
{
    RootTerms R;
    root_terms(root, c.root_attop, c.L.f, &R);
    Slot par = load_slot(p, c.L.row_par, m);
    Slot tr  = load_slot(p, c.L.row_tr, m);
    Slot ot  = load_slot(p, c.L.row_ot, m);
    line_terms(c.L.cfg, par, tr, ot, c.L.P ? R.inmv1 : R.inmv0, c.L.P ? R.sv1 : R.sv0,
               c.L.P == 0 && R.inmv0 == 2, T);
    double tot = line_total(*T);
    if (c.root_attop) {
        // the root is the top of its single line (cnF2freq.cpp:1260-1271): tables are 1
        tot = 1.0;
    }
    return tot;
}

__device__ __forceinline__ void restricted_entry(const LaneCtx& c, const Slot& root, const LineTerms& T, int combo,
                                                 double* rtot, double* two)
{
    if (c.root_attop) {
        *rtot = 1.0;
        int mf = c.L.f ? root.a1 : root.a0;
        *two   = (c.L.P == 0 && mf == 2) ? 1.0 : 0.0;
        return;
    }
    line_restricted(c.L.cfg, T, tie_force(c.L.tie_par, combo), tie_force(c.L.tie_tr, combo),
                    tie_force(c.L.tie_ot, combo), rtot, two);
}

// Consumer half: root weights c_f(s0) of this lane's shift mode.
__device__ __forceinline__ void root_weights(const LaneCtx& c, const Slot& root, double* c0, double* c1)
{
    RootTerms R0, R1;
    root_terms(root, c.root_attop, 0, &R0);
    root_terms(root, c.root_attop, 1, &R1);
    *c0 = R0.cbase * phase_weight(root, 0 ^ c.s0);
    *c1 = R1.cbase * phase_weight(root, 1 ^ c.s0);
}

__device__ __forceinline__ void wave_lds_fence()
{
    // producer lanes -> consumer lanes of the SAME wave: order the LDS write before the reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// e[j] for the lane's 8 states from one 64-entry table in LDS.
__device__ __forceinline__ void emission_from_table(const double* tab, const LaneCtx& c, double c0, double c1,
                                                    double (&e)[8])
{
    const double A0 = tab[(0 << 5) | (0 << 4) | (c.s1 << 3) | c.lo];
    const double A1 = tab[(0 << 5) | (1 << 4) | (c.s1 << 3) | c.lo];
    const double* B0 = tab + ((1 << 5) | (0 << 4) | (c.s2 << 3));
    const double* B1 = tab + ((1 << 5) | (1 << 4) | (c.s2 << 3));
#pragma unroll
    for (int j = 0; j < 8; j++) {
        // same association as the reference: ((c * other parent) * traced parent), f = 0 then 1
        e[j] = (c0 * B0[j]) * A0 + (c1 * B1[j]) * A1;
    }
}

// adjustprobs' scaling for the 8 chains at once (cnF2freq.cpp:1656-1669).
// mant/expo carry prod(sum) as mant * 2^expo; dead = a sum <= 0 was seen (factor := MINFACTOR).
// 1/x for a positive normal double: hardware estimate + two Newton steps (full double accuracy up
// to the last bit or two; the reference divides each state by sum, cnF2freq.cpp:1664-1667).
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r        = fma(fma(-x, r, 1.0), r, r);
    r        = fma(fma(-x, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ double scale_chain(double (&v)[8], double* mant, int* expo, bool* dead, double* inv_out = nullptr)
{
    double sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    sum        = chain_sum(sum);
    // sum <= 0: probs stay as they are (they are all zero), factor = MINFACTOR (cnF2freq.cpp:1656-1660);
    // written branch-free: a dead step scales by 1 and leaves the running scale alone
    const bool   ok  = sum > 0.0;
    const double ss  = ok ? sum : 1.0;
    const double inv = fast_rcp(ss);
    if (!ok) *dead = true;
    if (inv_out) *inv_out = inv;
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] *= inv;
    int    ex;
    double mm = frexp(*mant * ss, &ex);
    *mant     = mm;
    *expo += ex;
    return sum;
}

// Same bookkeeping, but the vector is left as it is: returns the reciprocal for the caller to fold into
// the next emission product (it is a per-chain scalar and every step is linear), which takes the
// reduction -> reciprocal chain off the critical path and saves six of the eight multiplies.
__device__ __forceinline__ double chain_normaliser(const double (&v)[8], double* mant, int* expo, bool* dead)
{
    double sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    sum        = chain_sum(sum);
    const bool   ok  = sum > 0.0;
    const double ss  = ok ? sum : 1.0;
    const double inv = fast_rcp(ss);
    if (!ok) *dead = true;
    int    ex;
    double mm = frexp(*mant * ss, &ex);
    *mant     = mm;
    *expo += ex;
    return inv;
}

// STOREW: 0 = plain sweep; 1 = accumulate mode (posterior weights of every marker into p.wbuf); 2 = turn-scan mode
// (alpha after emission and beta of every marker with their scales into p.wbuf, CNF2_TURN_ROW doubles per marker)
template <bool DEBUG_STORE, int STOREW = 0>
__global__ __launch_bounds__(CNF2_BLOCK) void fb_kernel(KernelParams p)
{
    __shared__ double lds_tab[CNF2_WAVES_PER_BLOCK][64];
    __shared__ double lds_rt[CNF2_WAVES_PER_BLOCK][64];
    __shared__ double lds_two[CNF2_WAVES_PER_BLOCK][64];

    const int lane  = threadIdx.x & 63;
    const int wib   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave  = blockIdx.x * CNF2_WAVES_PER_BLOCK + wib;
    const int nwave = gridDim.x * CNF2_WAVES_PER_BLOCK;
    double*   tab   = lds_tab[wib];
    double*   tabr  = lds_rt[wib];
    double*   tab2  = lds_two[wib];
    double*   spill = p.spill + (size_t)wave * p.spill_stride;

    for (int job = wave; job < p.n_jobs; job += nwave) {
        const Job    jb = p.jobs[job];
        const Window w  = p.windows[jb.ind];
        if (p.path_log && lane == 0) p.path_log[(size_t)jb.ind * p.n_chrom + jb.chrom] = PATH_GENERAL;
        LaneCtx      c;
        make_lane(w, lane, &c.L);
        c.row_root   = w.row[0];
        c.root_attop = (w.flags[0] & SLOT_FOUNDER) != 0;
        const int s  = lane >> 3;
        c.s0 = s & 1;
        c.s1 = (s >> 1) & 1;
        c.s2 = (s >> 2) & 1;
        c.lo      = state_lo(lane);
        c.active  = !(s & w.shiftignore) && s < w.shiftend;
        c.n_combo = (p.flags & KP_NO_TIES) ? 1 : (1 << w.n_groups);
        if (p.flags & KP_NO_TIES) c.L.tie_par = c.L.tie_tr = c.L.tie_ot = -1;
        const int first = jb.first, last = jb.last, len = last - first + 1;

        // ---------------------------------------------------------------- forward
        double a[8];
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = 1.0 / 64.0;           // EVENGEN, cnF2freq.cpp:2100-2104
        double mant = 1.0;
        int    expo = 0;
        bool   dead = false;
        double dbg_factor = 0.0;
        for (int m = first; m <= last; m++) {
            const Slot root = load_slot(p, c.row_root, m);
            LineTerms  T;
            double     tot = produce_entry(p, c, root, m, &T);
            tab[lane]      = tot;
            wave_lds_fence();
            double c0, c1, e[8];
            root_weights(c, root, &c0, &c1);
            emission_from_table(tab, c, c0, c1, e);
            wave_lds_fence();                                   // table may be overwritten next round

            // alpha-minus: register-major spill, 8 x 512 B coalesced
            double* sp = spill + (size_t)(m - first) * 512 + lane;
#pragma unroll
            for (int j = 0; j < 8; j++) sp[j * 64] = a[j];
            if (DEBUG_STORE) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    p.dbg_fwbw[(((size_t)s * len + (m - first)) * 3 + 0) * 64 + j * 8 + c.lo] = a[j];
                if (c.lo == 0) p.dbg_factors[((size_t)s * len + (m - first)) * 3 + 0] = dbg_factor;
            }
            // "We will multiply this already small number with an even smaller number... let's assume it's zero" (adjustprobs,
            // cnF2freq.cpp:1607-1611): the vector is normalised here as it is there, so the threshold is the reference's
            if (p.flags & KP_FLUSH_TINY) {
#pragma unroll
                for (int j = 0; j < 8; j++) a[j] = a[j] < 1e-300 ? 0.0 : a[j];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] *= e[j];
            bool   was_dead = dead;
            double sum      = scale_chain(a, &mant, &expo, &dead);
            if (DEBUG_STORE) {
                if (dead) dbg_factor = (double)CNF2_MINFACTOR_F;
                else dbg_factor += log(sum);
                (void)was_dead;
#pragma unroll
                for (int j = 0; j < 8; j++)
                    p.dbg_fwbw[(((size_t)s * len + (m - first)) * 3 + 2) * 64 + j * 8 + c.lo] = a[j];
                if (c.lo == 0) p.dbg_factors[((size_t)s * len + (m - first)) * 3 + 2] = dbg_factor;
            }
            if (m < last) {
                const double2 r = p.rho[m];                     // gap m -> m+1 (0 when dist <= 0)
                transition(a, r.x, r.y);
            }
        }

        // ---------------------------------------------------------------- likelihoods
        // factors[s] (cnF2freq.cpp:5375-5382) and factor = logsumexp (cnF2freq.cpp:5384-5400)
        double fs = dead ? (double)CNF2_MINFACTOR_F : (log(mant) + (double)expo * 0.69314718055994530942);
        if (!c.active) fs = CNF2_IGNORED_D;
        double fmaxv = across_chains_max(fs);
        fmaxv        = fmax(fmaxv, -1e15);                      // cnF2freq.cpp:5373
        double term  = c.active ? exp(fs - fmaxv) : 0.0;
        double real  = across_chains_sum(term);
        double factor = fmaxv + log(real);
        if (c.lo == 0) p.factors[((size_t)jb.ind * p.n_chrom + jb.chrom) * 8 + s] = fs;
        if (lane == 0) p.loglik[(size_t)jb.ind * p.n_chrom + jb.chrom] = factor;
        const bool skip = isnan(factor) || factor < (double)CNF2_MINFACTOR_F;      // cnF2freq.cpp:5403
        // weight of this chain in the per-locus row; cnF2freq.cpp:5421 drops modes 40 log units down
        const double ws = (c.active && !skip && !(factor - fs > 40.0)) ? exp(fs - factor) : 0.0;

        if ((p.flags & KP_NO_DOSAGE) && !DEBUG_STORE) continue;

        // ---------------------------------------------------------------- backward + rows
        double b[8];
#pragma unroll
        for (int j = 0; j < 8; j++) b[j] = 1.0;                  // cnF2freq.cpp:2111-2114
        double bmant = 1.0;
        int    bexpo = 0;
        bool   bdead = false;
        double dbg_bfactor = 0.0;
        for (int m = last; m >= first; m--) {
            const Slot root = load_slot(p, c.row_root, m);
            LineTerms  T;
            double     tot = produce_entry(p, c, root, m, &T);
            double     c0, c1, e[8];
            root_weights(c, root, &c0, &c1);
            tab[lane] = tot;
            wave_lds_fence();
            emission_from_table(tab, c, c0, c1, e);

            if (DEBUG_STORE) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    p.dbg_fwbw[(((size_t)s * len + (m - first)) * 3 + 1) * 64 + j * 8 + c.lo] = b[j];
                if (c.lo == 0) p.dbg_factors[((size_t)s * len + (m - first)) * 3 + 1] = dbg_bfactor;
            }

            if (!(p.flags & KP_NO_DOSAGE)) {
                // w_j = alpha-minus * beta ; D = sum_g w e  (normaliser of this chain at this locus)
                const double* sp = spill + (size_t)(m - first) * 512 + lane;
                double        wj[8], D = 0.0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    wj[j] = sp[j * 64] * b[j];
                    D += wj[j] * e[j];
                }
                double n_tot = 0.0, n_a1 = 0.0, n_b1 = 0.0, n_2 = 0.0;
                for (int combo = 0; combo < c.n_combo; combo++) {
                    double rt, tw;
                    restricted_entry(c, root, T, combo, &rt, &tw);
                    wave_lds_fence();
                    tabr[lane] = rt;
                    tab2[lane] = tw;
                    wave_lds_fence();
#pragma unroll
                    for (int f = 0; f < 2; f++) {
                        const int    ia = (0 << 5) | (f << 4) | (c.s1 << 3) | c.lo;
                        const double* Br = tabr + ((1 << 5) | (f << 4) | (c.s2 << 3));
                        const double* B1 = tab2 + ((1 << 5) | (f << 4) | (c.s2 << 3));
                        const double  av = tabr[ia], a1 = tab2[ia];
                        double        sb = 0.0, sb1 = 0.0;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            sb += wj[j] * Br[j];
                            sb1 += wj[j] * B1[j];
                        }
                        const double cf = f ? c1 : c0;
                        n_tot += cf * av * sb;
                        n_a1 += cf * a1 * sb;
                        n_b1 += cf * av * sb1;
                        n_2 += cf * a1 * sb1;
                    }
                }
                D     = chain_sum(D);
                if (STOREW == 2) {
                    // turn-scan mode: A = alphaminus e, B = beta (both as the normalised vectors held here) and their
                    // scales as mantissa and binary exponent: log2 P(data, mode) = fs / ln 2 = lgA + lgB + log2 D
                    double* wp = p.wbuf + ((size_t)job * p.wstride + (m - first)) * CNF2_TURN_ROW;
                    const double lgA = fs * 1.4426950408889634074 - log2(D) - (log2(bmant) + (double)bexpo);
                    const double eA  = floor(lgA);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        *(double2*)(wp + k * 128 + lane * 2) = make_double2(sp[(2 * k) * 64] * e[2 * k], sp[(2 * k + 1) * 64] * e[2 * k + 1]);
                        *(double2*)(wp + 512 + k * 128 + lane * 2) = make_double2(b[2 * k], b[2 * k + 1]);
                    }
                    if (c.lo == 0) {
                        // a chain with no likelihood at all (fs = -inf or the sentinel): scale 0, the turn scan's floor
                        const bool none = !(lgA > -1e300);
                        *(double2*)(wp + 1024 + 4 * s)     = make_double2(none ? 0.0 : exp2(lgA - eA), none ? 0.0 : eA);
                        *(double2*)(wp + 1024 + 4 * s + 2) = make_double2(bmant, (double)bexpo);
                    }
                }
                if (STOREW == 1) {
                    // accumulate mode: wg(s, g) = exp(scales - factor) alphaminus beta = wj * ws / D
                    const double sw = (D > 0.0) ? ws / D : 0.0;
                    double*      wp = p.wbuf + ((size_t)job * p.wstride + (m - first)) * 512;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        double2 v2 = make_double2(sw != 0.0 ? sw * wj[2 * k] : 0.0, sw != 0.0 ? sw * wj[2 * k + 1] : 0.0);
                        *(double2*)(wp + k * 128 + lane * 2) = v2;
                    }
                }
                n_tot = chain_sum(n_tot);
                n_a1  = chain_sum(n_a1);
                n_b1  = chain_sum(n_b1);
                n_2   = chain_sum(n_2);
                const double scale = (D > 0.0) ? ws / D : 0.0;
                // (inclusion-exclusion of non-negative sums: a true zero can come out as -1e-16; clamp)
                double d2 = fmax(across_chains_sum(scale * n_2), 0.0);
                double d1 = fmax(across_chains_sum(scale * (n_a1 + n_b1 - 2.0 * n_2)), 0.0);
                double d0 = fmax(across_chains_sum(scale * (n_tot - n_a1 - n_b1 + n_2)), 0.0);
                if (lane == 0) {
                    if (!(p.flags & KP_RAW_DOSAGE)) {
                        double tsum = d0 + d1 + d2;
                        double inv  = tsum > 0.0 ? 1.0 / tsum : 0.0;
                        d0 *= inv;
                        d1 *= inv;
                        d2 *= inv;
                    }
                    double* out = p.dosage + ((size_t)jb.ind * p.n_markers + m) * 3;
                    out[0] = d0;
                    out[1] = d1;
                    out[2] = d2;
                }
            }

            // beta_{m-1} = T (e_m . beta_m), rescaled (cnF2freq.cpp:2238 with d = -1, then 2273-2367)
            if (m > first) {
                if (p.flags & KP_FLUSH_TINY) {
#pragma unroll
                    for (int j = 0; j < 8; j++) b[j] = b[j] < 1e-300 ? 0.0 : b[j];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) b[j] *= e[j];
                double sum = scale_chain(b, &bmant, &bexpo, &bdead);
                if (DEBUG_STORE) {
                    if (bdead) dbg_bfactor = (double)CNF2_MINFACTOR_F;
                    else dbg_bfactor += log(sum);
                }
                const double2 r = p.rho[m - 1];
                transition(b, r.x, r.y);
            }
            wave_lds_fence();
        }
    }
}


// =====================================================================================
// Fast kernel: windows without an active tie (the common case; F2, outbred, ...).
// Same lane/register layout and the same sweep arithmetic as fb_kernel; the emission tables
// are produced per tile of 8 markers by cnf2_emtab.h (lane = part x marker, division-free)
// into LDS, so the per-marker critical path only reads them.
// LDS per wave: 8 markers x TAB_STRIDE doubles: [0,64) tot, [64,68) root weights c[f][s0],
// and for the backward pass [72,136) restricted totals, [136,200) class-2 parts.
// =====================================================================================
#ifndef CNF2_RESCALE_MASK
#define CNF2_RESCALE_MASK 7   /* half-spill sweep: rescale alpha / beta at markers whose local index has these bits clear
                                 (7 = once per tile of 8 markers; measured +3.3 % over rescaling at every second marker) */
#endif
#define CNF2_RESCALE_GUARD 1e150 /* a normaliser below 1 / this makes the wave rescale at every second marker from there on */
#define TAB_STRIDE 202   /* doubles per marker row: 16-B aligned rows, conflict-free producer stores */
#define TAB_C 64
#define TAB_T 68     /* double2: r/(1-r) of the gap carried by this row */
#define TAB_R 72
#define TAB_2 136

typedef double d2v __attribute__((ext_vector_type(2)));     // 16-byte operand of the nontemporal builtins

struct FastCtx {
    PartCfg pc;
    int     part, mi;
    int     idx_base, idx_k01, idx_k10;   // where this lane's entries go in a table row (produce_tile)
    int32_t row_root, row_par, row_tr, row_ot;
    int     slot_a;                       // two-phase producer: window slot this lane prepares (lane >> 3; 7 = none)
    int32_t row_a;                        //                     and its genotype row
    int     s0, s1, s2, lo;
    bool    active;
};

// Raw inputs of one lane of the tile producer (4 window members at one marker), kept packed so
// that they can be requested a whole tile ahead.
struct RawSlots {
    uint8_t ap[4];
    double2 su[4];
    double  hw[4];
    double2 tq;      // scaled recombination odds of the gap this marker's row carries (see produce_tile)
};

// TQ_SHIFT: the forward pass needs the gap after marker m in row m, the backward pass the gap before it
template <int TQ_SHIFT>
__device__ __forceinline__ void load_raw_at(const KernelParams& p, const FastCtx& c, int m, int lo_m, int hi_m, RawSlots* r)
{
    m = m < lo_m ? lo_m : (m > hi_m ? hi_m : m);          // clamp: lanes beyond the chromosome load a valid marker
    const int32_t rows[4] = {c.row_root, c.row_par, c.row_tr, c.row_ot};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t i = (size_t)rows[k] * p.n_markers + m;
        r->ap[k] = p.allele8[i];
        r->su[k] = p.sure[i];
        r->hw[k] = p.hw[i];
    }
    const int mt = m + TQ_SHIFT;
    r->tq        = p.tq[mt < 0 ? 0 : mt];
}
template <int TQ_SHIFT>
__device__ __forceinline__ void load_raw(const KernelParams& p, const FastCtx& c, int m0, int lo_m, int hi_m, RawSlots* r)
{
    load_raw_at<TQ_SHIFT>(p, c, m0 + c.mi, lo_m, hi_m, r);
}

// the table row c.mi of the tile from this lane's raw inputs (valid: the row's marker exists)
// STRIDE: doubles per table row; KOFF > 0 (tie combinations, second of a pair): only the restricted tables are stored,
// KOFF doubles behind the first combination's
template <bool CLASSES, bool HOMPAR = false, bool HOMLEAF = false, bool NORESTR = false, bool TIES = false, int STRIDE = TAB_STRIDE,
          int KOFF = 0>
__device__ __forceinline__ void produce_row(const FastCtx& c, double* tab, bool valid, const RawSlots& raw)
{
    if (valid) {
        const Slot root = unpack_slot(raw.ap[0], raw.su[0].x, raw.su[0].y, raw.hw[0]);
        const Slot par  = unpack_slot(raw.ap[1], raw.su[1].x, raw.su[1].y, raw.hw[1]);
        const Slot tr   = unpack_slot(raw.ap[2], raw.su[2].x, raw.su[2].y, raw.hw[2]);
        const Slot ot   = unpack_slot(raw.ap[3], raw.su[3].x, raw.su[3].y, raw.hw[3]);
        double  cw[2];
        double* row = tab + c.mi * STRIDE;
        // entry e = sp*4 + bit_ot*2 + bit_tr goes to table index base + sp*8 + k: k = 0 / 6 for the two
        // diagonal entries, and the two off-diagonal ones swap places with firstpar (part_entry_index); the
        // two lane-dependent offsets are per-job constants, so no value is ever selected.  Entries are stored
        // as they are formed: the producer holds no output array in registers.
        double* rb = row + c.idx_base;
        emtab_part_to<CLASSES, HOMPAR, HOMLEAF, NORESTR, TIES>(c.pc, root, par, tr, ot,
                               [&](int kind, int e, double v) {
                                   if (KOFF > 0 && kind == 0) return;
                                   const int b  = e & 3;
                                   const int k  = b == 0 ? 0 : (b == 1 ? c.idx_k01 : (b == 2 ? c.idx_k10 : 6));
                                   rb[(kind == 0 ? 0 : (kind == 1 ? TAB_R : TAB_2) + KOFF) + (e >> 2) * 8 + k] = v;
                               },
                               cw);
        if (KOFF > 0) return;
        if ((c.part & 5) == 0) {                    // P == 0, firstpar == 0: one writer per f
            row[TAB_C + c.pc.f * 2 + 0] = cw[0];
            row[TAB_C + c.pc.f * 2 + 1] = cw[1];
        }
        // the gap's butterfly factors ride in the row: the marker loop never touches vmcnt for them
        // (a vector load there would make every s_waitcnt drain the spill stores as well)
        if (c.part == 0) *(double2*)(row + TAB_T) = raw.tq;
    }
}
// hom (wave-uniform): 1 = both parents of the window are homozygous with equal sure at every marker (SLOT_HOM),
// so each part evaluates one allele index of its parent (cnf2_emtab.h HOMPAR); 2 = the four grandparents are
// present and homozygous everywhere as well (HOMLEAF) -- the F2 with empty F1 parents and inbred founders;
// 3 = no slot of the window is restricted (flag2ignore == 0: a complete window): the restricted table is a
// copy of the unrestricted one (NORESTR)
// TIES: the restricted tables of the tie combination whose forces are in c.pc (general form of the producer)
template <bool CLASSES, bool TIES = false, int STRIDE = TAB_STRIDE, int KOFF = 0>
__device__ __forceinline__ void produce_tile(const KernelParams& p, const FastCtx& c, double* tab, int m0, int last,
                                             const RawSlots& raw, int hom)
{
    if (TIES) {
        produce_row<CLASSES, false, false, false, CLASSES, STRIDE, KOFF>(c, tab, m0 + c.mi <= last, raw);
        return;
    }
    if (hom == 2) produce_row<CLASSES, true, true, false, false, STRIDE>(c, tab, m0 + c.mi <= last, raw);
    else if (hom == 1) produce_row<CLASSES, true, false, false, false, STRIDE>(c, tab, m0 + c.mi <= last, raw);
    else if (hom == 3) produce_row<CLASSES, false, false, true, false, STRIDE>(c, tab, m0 + c.mi <= last, raw);
    else produce_row<CLASSES, false, false, false, false, STRIDE>(c, tab, m0 + c.mi <= last, raw);
}

// Raw inputs of one lane of the tile producer's first phase (ONE window member at one marker), requested a
// whole tile ahead.
struct RawOne {
    uint8_t ap;
    double2 su;
    double  hw;
    double2 tq;      // scaled recombination odds of the gap this marker's row carries (lanes of slot 0)
};

// TQ_SHIFT: the forward pass needs the gap after marker m in row m, the backward pass the gap before it
template <int TQ_SHIFT>
__device__ __forceinline__ void load_raw1_at(const KernelParams& p, const FastCtx& c, int m, int lo_m, int hi_m, RawOne* r)
{
    m = m < lo_m ? lo_m : (m > hi_m ? hi_m : m);          // clamp: lanes beyond the chromosome load a valid marker
    const size_t i = (size_t)c.row_a * p.n_markers + m;
    r->ap = p.allele8[i];
    r->su = p.sure[i];
    r->hw = p.hw[i];
    const int mt = m + TQ_SHIFT;
    r->tq        = p.tq[mt < 0 ? 0 : mt];
}
// Two-phase form of produce_row (used by the packed kernel, where the producer runs every second marker and
// its register footprint decides what spills; for the ordinary kernel the extra LDS round trips cost more than
// the shorter match logic saves: measured 657 against 629 ms).  The table row c.mi in two phases: lanes (slot, row) turn the raw
// data of one window member into its match record (cnf2_emtab.h slot_table: 20 doubles in the row itself); then
// lanes (part, row) combine the records of their line into their 8 entries per table kind and, once every
// lane has read what it needs, store them over the records.
template <bool CLASSES, bool HOMPAR = false, bool HOMLEAF = false>
__device__ __forceinline__ void produce_row2(const FastCtx& c, double* tab, bool valid, const RawOne& raw)
{
    double* row = tab + c.mi * TAB_STRIDE;
    if (valid && c.slot_a < 7)
        slot_table(unpack_slot(raw.ap, raw.su.x, raw.su.y, raw.hw), row + c.slot_a * SLOTTAB_DOUBLES);
    wave_lds_fence();
    double tot[8], rtot[8], two[8], cw[2];
    if (valid)
        emtab_part_tables<CLASSES, HOMPAR, HOMLEAF>(c.pc, row,
                                   [&](int kind, int e, double v) { (kind == 0 ? tot : (kind == 1 ? rtot : two))[e] = v; }, cw);
    wave_lds_fence();
    if (valid) {
        // entry e = sp*4 + bit_ot*2 + bit_tr goes to table index base + sp*8 + k: k = 0 / 6 for the two
        // diagonal entries, and the two off-diagonal ones swap places with firstpar (part_entry_index); the
        // two lane-dependent offsets are per-job constants, so no value is ever selected
        double* rb = row + c.idx_base;
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
            double* r8 = rb + sp * 8;
            r8[0]         = tot[sp * 4 + 0];
            r8[c.idx_k01] = tot[sp * 4 + 1];
            r8[c.idx_k10] = tot[sp * 4 + 2];
            r8[6]         = tot[sp * 4 + 3];
            if (CLASSES) {
                r8[TAB_R + 0]         = rtot[sp * 4 + 0];
                r8[TAB_R + c.idx_k01] = rtot[sp * 4 + 1];
                r8[TAB_R + c.idx_k10] = rtot[sp * 4 + 2];
                r8[TAB_R + 6]         = rtot[sp * 4 + 3];
                r8[TAB_2 + 0]         = two[sp * 4 + 0];
                r8[TAB_2 + c.idx_k01] = two[sp * 4 + 1];
                r8[TAB_2 + c.idx_k10] = two[sp * 4 + 2];
                r8[TAB_2 + 6]         = two[sp * 4 + 3];
            }
        }
        if ((c.part & 5) == 0) {                    // P == 0, firstpar == 0: one writer per f
            row[TAB_C + c.pc.f * 2 + 0] = cw[0];
            row[TAB_C + c.pc.f * 2 + 1] = cw[1];
        }
        // the gap's butterfly factors ride in the row: the marker loop never touches vmcnt for them
        // (a vector load there would make every s_waitcnt drain the spill stores as well)
        if (c.part == 0) *(double2*)(row + TAB_T) = raw.tq;
    }
}
__device__ __forceinline__ void emission_from_row(const double* row, const FastCtx& c, double (&e)[8], double k = 1.0)
{
    const double cA0 = row[TAB_C + 0 + c.s0] * row[(0 << 5) | (0 << 4) | (c.s1 << 3) | c.lo] * k;
    const double cA1 = row[TAB_C + 2 + c.s0] * row[(0 << 5) | (1 << 4) | (c.s1 << 3) | c.lo] * k;
    const double* B0 = row + ((1 << 5) | (0 << 4) | (c.s2 << 3));
    const double* B1 = row + ((1 << 5) | (1 << 4) | (c.s2 << 3));
#pragma unroll
    for (int j = 0; j < 8; j++) e[j] = cA0 * B0[j] + cA1 * B1[j];
}

// The same in the transposed layout of an odd marker (XPOSE variant): the lane holds state bits 3-5 (its low lane bits
// index the B half of the table), the registers state bits 0-2 (the A half).
__device__ __forceinline__ void emission_from_row_t(const double* row, const FastCtx& c, double (&e)[8], double k = 1.0)
{
    const double cB0 = row[TAB_C + 0 + c.s0] * row[(1 << 5) | (0 << 4) | (c.s2 << 3) | c.lo] * k;
    const double cB1 = row[TAB_C + 2 + c.s0] * row[(1 << 5) | (1 << 4) | (c.s2 << 3) | c.lo] * k;
    const double* A0 = row + ((0 << 5) | (0 << 4) | (c.s1 << 3));
    const double* A1 = row + ((0 << 5) | (1 << 4) | (c.s1 << 3));
#pragma unroll
    for (int j = 0; j < 8; j++) e[j] = cB0 * A0[j] + cB1 * A1[j];
}

// Running state of the backward pass of one lane (kept in one struct so that the per-marker body can be
// instantiated for even and odd markers without a merge of differently-defined values between them).
#define CNF2_LI_K __attribute__((always_inline))
struct BwdState {
    double b[8];          // beta
    double am[8];         // alpha-minus of the row in flight (HALF: of the even marker of the pair)
    double inv_even;      // reciprocal normaliser stored with that row
    double inv_odd;       // HALF: the one of an unrescaled odd last marker (else 1)
    double bmant, fmant;
    int    bexpo, fexpo;
    bool   bdead;
    double ec[8];         // emission of the even marker of the pair: formed for the odd marker's rebuild, reused when
                          // the even marker itself is reached (it is the next one; measured +2.2 %)
};

// HALF: alpha-minus is spilled for every second marker only; the backward pass rebuilds the odd ones
// with one forward step from the stored even neighbour (same arithmetic, same bits).  Halves the
// spill traffic for ~15 % more arithmetic.
// TIED: windows with tie groups (ignoreflag2's all-or-none rule, cnF2freq.cpp:3484-3486).  The forward pass and the beta
// recursion do not see the rule; the per-locus rows are sums over the tie combinations (2, 4 or 8 of them) of class sums
// taken with that combination's restricted tables.  A table row holds the restricted tables of TWO combinations; a marker's
// posterior weights are contracted with both (everything else at the marker is the same for every combination), so a
// window with one tie group -- most of them -- is swept in one backward pass, one with two groups in two.  Every further
// pass over a tile starts from the tile's state (beta, scales: kept in LDS; spill row and tile inputs: asked for again)
// and the tile epilogue adds the passes up before it writes the rows.  The tables are dynamic LDS (84 KB a block, see the
// declaration): a block shares a CU with a block of the untied windows' kernel, not with another of its own.
template <bool HALF, int STOREW = 0, bool XPOSE = false, bool TIED = false>
__global__ __launch_bounds__(CNF2_BLOCK, 2) void fb_fast_kernel(KernelParams p)
{
    static_assert(!XPOSE || (HALF && STOREW == 0), "the transposing variant exists for the plain half-spill sweep");
    // STOREW: 0 plain sweep; 1 accumulate mode (also stores the posterior weights wg); 2 turn-scan mode (stores alpha e, beta
    // and their scales; no rows); 3 accumulate mode of a call that did not ask for the per-locus rows (wg only)
    constexpr bool ROWS = STOREW == 0 || STOREW == 1;      // class sums, restricted tables, tile epilogue, p.dosage
    constexpr bool WG   = STOREW == 1 || STOREW == 3;
    static_assert(!TIED || (!XPOSE && ROWS), "tie combinations only matter to the rows");
    // Spill row (528 doubles): [k = 0..3][lane][2] = registers 2k, 2k+1 of every lane (one 16-byte access
    // per lane and k), then [chain][2] = reciprocal normaliser of the (even) marker and, HALF only, of
    // an odd last marker.
    constexpr int ROW = 528;
    // TIED: the restricted tables of TWO tie combinations per row (the second pair TIE_KOFF doubles behind the first)
    constexpr int TIE_KOFF = 128;
    constexpr int TS = TIED ? TAB_STRIDE + TIE_KOFF : TAB_STRIDE;
    // (TIED: the tables are dynamic LDS, CNF2_TIED_LDS_BYTES at the launch.  With 105 KB of static LDS the compiler concludes
    // that only one block fits a CU and hands every wave 400 registers -- which keeps the block from sharing a CU with a
    // block of the untied windows' kernel; what it cannot see it does not count, and the launch bounds' 256 registers stand)
    __shared__ __attribute__((aligned(16))) double lds[TIED ? 1 : CNF2_WAVES_PER_BLOCK][TIED ? 2 : 8 * TS];
    extern __shared__ __attribute__((aligned(16))) double lds_tied[];
    __shared__ __attribute__((aligned(16))) double xlds[XPOSE ? CNF2_WAVES_PER_BLOCK : 1][XPOSE ? 64 * XPOSE_RS : 2];
    __shared__ double tsum[TIED ? CNF2_WAVES_PER_BLOCK : 1][TIED ? 24 : 1];     // TIED: class sums of the tile's markers over the combinations
    // TIED: beta and the two scale mantissas as they stand at the tile's start ([k][lane]: conflict-free), read back for
    // every further combination -- 20 registers per lane that would otherwise live (in scratch) across the whole tile
    __shared__ double tsave[TIED ? CNF2_WAVES_PER_BLOCK : 1][TIED ? 11 * 64 : 1];

    const int wib   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave  = blockIdx.x * CNF2_WAVES_PER_BLOCK + wib;
    const int nwave = gridDim.x * CNF2_WAVES_PER_BLOCK;
    double*   tab   = TIED ? lds_tied + wib * (8 * TS) : lds[TIED ? 0 : wib];
    double*   xb    = xlds[XPOSE ? wib : 0];
    double*   spill = p.spill + (size_t)wave * p.spill_stride;
    // shader-clock and wall-clock ticks of the first wave (the bench's effective clock of this very kernel)
    const bool stamp = p.clock_out != nullptr && wave == 0;
    unsigned long long t_shader = 0, t_wall = 0;
    if (stamp) {
        t_shader = __builtin_readcyclecounter();
        t_wall   = wall_clock64();
    }

    // the next job of this wave: from the launch's counter (one atomic per job, by lane 0) or, without one, by striding
    auto take_job = [&](int strided) {
        if (!p.job_next) return strided;
        int j = 0;
        if (fresh_lane() == 0) j = atomicAdd(p.job_next, 1);
        return __builtin_amdgcn_readfirstlane(j);
    };
    for (int job = take_job(wave); job < p.n_jobs; job = take_job(job + nwave)) {
        // the lane number read afresh, as a value the compiler cannot see through: what is derived from it is formed per
        // job, where it is used, instead of being held in registers (or spilled) for the whole kernel
        const int lane = fresh_lane();
        const Job    jb = p.jobs[job];
        const Window w  = p.windows[jb.ind];
        FastCtx      c;
        c.part = lane >> 3;
        c.mi   = lane & 7;
        make_part(w, c.part, &c.pc, &c.row_par, &c.row_tr, &c.row_ot);
        c.idx_base  = part_entry_index(c.part, 0);          // (sp, bit_ot, bit_tr) = (0, 0, 0)
        c.idx_k01   = part_entry_index(c.part, 1) - c.idx_base;
        c.idx_k10   = part_entry_index(c.part, 2) - c.idx_base;
        c.row_root  = w.row[0];
        int hom = 0;
        if (w.flags[1] & w.flags[4] & SLOT_HOM) {
            hom = 1;
            const int gp = w.flags[2] & w.flags[3] & w.flags[5] & w.flags[6];
            if ((gp & SLOT_HOM) && (gp & SLOT_PRESENT)) hom = 2;
        }
        if (hom == 0 && w.flag2ignore == 0) hom = 3;
        if (TIED) hom = 0;
        hom = __builtin_amdgcn_readfirstlane(hom);
        const int n_combo = TIED ? __builtin_amdgcn_readfirstlane(1 << w.n_groups) : 1;
        // TIED: the tie groups of this lane's three slots (part_forces), one byte each, so that the window itself need not
        // stay live for the combinations' loop
        int my_ties = 0;
        if (TIED) {
            const bool   hi = (c.part >> 2) != 0, fp = (c.part & 1) != 0;
            const int8_t tp = hi ? w.tie[4] : w.tie[1];
            const int8_t ta = hi ? w.tie[5] : w.tie[2], tb = hi ? w.tie[6] : w.tie[3];
            const int8_t tt = fp ? tb : ta, to = fp ? ta : tb;
            my_ties = (tp & 255) | ((tt & 255) << 8) | ((to & 255) << 16);
        }
        if (p.path_log && lane == 0) p.path_log[(size_t)jb.ind * p.n_chrom + jb.chrom] = TIED ? PATH_TIED : hom;
        const int s = lane >> 3;
        c.s0 = s & 1;
        c.s1 = (s >> 1) & 1;
        c.s2 = (s >> 2) & 1;
        c.lo = XPOSE ? (lane & 7) : state_lo(lane);   // no DPP butterflies in the transposing variant: plain order
        c.active = !(s & w.shiftignore) && s < w.shiftend;
        const int first = jb.first, last = jb.last;
        const int ntile = (last - first + 8) >> 3;

        using odd_t  = std::integral_constant<bool, true>;
        using even_t = std::integral_constant<bool, false>;

        // ---------------------------------------------------------------- forward
        double a[8];
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = 1.0 / 64.0;
        double mant = 1.0;
        int    expo = 0;
        bool   dead = false;
        double pend = 1.0;     // reciprocal normaliser not yet applied to a[] (folded into the next emission)
        int    rmask = CNF2_RESCALE_MASK, bmask = CNF2_RESCALE_MASK;   // rescaling pattern of the forward / backward pass
        // One marker of the forward pass.  ODD (HALF only): nothing is spilled and, except at the last
        // marker of the chromosome, nothing is rescaled: two emission products in a row cannot underflow a
        // double.  A skipped step has normaliser 1 (the reference rescales at every marker,
        // cnF2freq.cpp:1664-1668; only the bookkeeping of the scale differs, not the normalised values).
        auto fwd_step = [&](auto odd_tag, const double* row, int m) {
            constexpr bool ODD = decltype(odd_tag)::value;
            double         e[8];
            if (XPOSE && ODD) emission_from_row_t(row, c, e, pend);
            else emission_from_row(row, c, e, pend);
            const double2 r  = *(const double2*)(row + TAB_T);
            const int     ml = m - first;
            double*       sp = spill + (size_t)(HALF ? (ml >> 1) : ml) * ROW;
            if (!ODD) {
#ifndef CNF2_X_NOSTORE   /* timing ablation only: results are wrong */
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // written once, read once a whole chromosome later: streaming (nt) accesses keep the rows
                    // from churning L2 (measured: -2 %)
                    const d2v v = {a[2 * k], a[2 * k + 1]};
                    __builtin_nontemporal_store(v, (d2v*)(sp + k * 128 + lane * 2));
                }
#endif
            }
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] *= e[j];
            pend = 1.0;
            // HALF: the vectors are rescaled at every CNF2_RESCALE-th marker only (and at the last one): a few emission
            // products in a row cannot underflow a double, and a skipped step simply has normaliser 1
            const bool norm_here = !HALF || m == last || (!ODD && (ml & rmask) == 0);
            if (norm_here) {
                // reciprocal of this step's normaliser, per chain: stored so that the backward pass can
                // rebuild the forward scale before each marker without a reduction (and redo the forward step)
                double inv;
                if (HALF) {
                    inv  = chain_normaliser(a, &mant, &expo, &dead);
                    pend = inv;
                    // data that loses > 150 decades in one stretch: keep the vectors in range by rescaling densely
                    if (__ballot(inv > CNF2_RESCALE_GUARD)) rmask = 1;
                } else {
                    scale_chain(a, &mant, &expo, &dead, &inv);     // full spill: the stored rows are normalised at once
                }
                if (c.lo == 0) sp[512 + 2 * s + (ODD ? 1 : 0)] = inv;
            } else if (!ODD) {
                if (c.lo == 0) sp[512 + 2 * s] = 1.0;
            }
            if (m < last) {
                if (XPOSE) transition_xpose(a, r.x, r.y, xb, lane);
                else transition_scaled(a, r.x, r.y);
            }
        };
        RawSlots raw;
        load_raw<0>(p, c, first, first, last, &raw);
        for (int t = 0; t < ntile; t++) {
            const int m0 = first + t * 8;
#ifdef CNF2_X_NOPRODUCE  /* timing ablation only: results are wrong */
            if (t == 0)
#endif
            {
                produce_tile<false, false, TS>(p, c, tab, m0, last, raw, hom);
                if (t + 1 < ntile) load_raw<0>(p, c, m0 + 8, first, last, &raw);   // next tile's inputs, a tile ahead
            }
            wave_lds_fence();
            const int mend = (m0 + 7 < last) ? m0 + 7 : last;
            if (HALF) {
                const int itop = mend - m0;
                for (int i = 0; i <= itop; i += 2) {
                    fwd_step(even_t(), tab + i * TS, m0 + i);
                    if (i < itop) fwd_step(odd_t(), tab + (i + 1) * TS, m0 + i + 1);
                }
            } else {
                for (int m = m0; m <= mend; m++) fwd_step(even_t(), tab + (m - m0) * TS, m);
            }
            wave_lds_fence();
        }

        // ---------------------------------------------------------------- likelihoods
        // P(data, mode s) = mant 2^expo per chain.  Everything the backward pass needs of them is exact binary scaling:
        // with T = sum over the live chains of mant 2^(expo - emax), exp(-factor) = (1 / T) 2^-emax, and the reference's
        // "40 below the total" rule (cnF2freq.cpp:5420-5421) compares a chain's term with T.  The logarithms the
        // outputs hold (and the chromosome's dropped butterfly constants) are taken by likelihood_logs_kernel after the
        // sweep: the sweep kernel has no transcendental and none of their constants in its registers.
        const bool   alive  = c.active && !dead;
        const double emaxd  = across_chains_max(alive ? (double)expo : -1e300);
        const bool   any_alive = emaxd > -1e299;
        const int    emax   = any_alive ? (int)emaxd : 0;
        int          dsh    = expo - emax;
        dsh                 = dsh < -2000 ? -2000 : dsh;
        const double tmine  = alive ? ldexp(mant, dsh) : 0.0;
        const double T      = across_chains_sum(tmine);
        {
            const size_t e = (size_t)jb.ind * p.n_chrom + jb.chrom;
            const int    n_active = __builtin_popcountll(__ballot(c.active)) >> 3;
            if (c.lo == 0) {
                p.factors[e * 8 + s] = mant;
                p.fexp[e * 8 + s]    = alive ? expo : (c.active ? CNF2_LEXP_DEAD : CNF2_LEXP_IGNORED);
            }
            if (lane == 0) {
                // no live chain: every analysed mode sits at the floor, the total is the floor + log(their number)
                p.loglik[e] = any_alive ? T : (double)n_active;
                p.lexp[e]   = any_alive ? emax : CNF2_LEXP_DEAD;
            }
        }
        if (p.flags & KP_NO_DOSAGE) continue;

        // ---------------------------------------------------------------- backward + rows
        // Row of the reference: sum over (g, s, path) of val = exp(query - factor) by class
        // (cnF2freq.cpp:5499-5508, 3536).  With Fpre(m) = prod_{k<m} sum_k (forward normalisers) and
        // Bsuf(m) the backward ones, val summed over paths of class d and states is
        //     exp(-factor) * Fpre_s(m) * Bsuf_s(m) * sum_g alphaminus_s(g) beta_s(g) e^{(d)}_s(g).
        // Scales are carried as mantissa * 2^exponent; Fpre is rebuilt from the stored reciprocals.
        BwdState S;
#pragma unroll
        for (int j = 0; j < 8; j++) S.b[j] = 1.0;
        S.bmant = 1.0;
        S.bexpo = 0;
        S.bdead = false;
        S.fmant = mant;          // becomes Fpre(m) after multiplying the reciprocals of k >= m
        S.fexpo = expo;
        // exp(-factor) = xm * 2^xe; a job without a live chain has no rows (cnF2freq.cpp:5403)
        const double xm  = any_alive ? 1.0 / T : 0.0;
        const int    xe  = -emax;
        const bool   chain_on = alive && !(tmine * 2.3538526683701998e17 < T);        // factor - fs > 40: cnF2freq.cpp:5420-5421
        // software pipeline: the spill row (and its reciprocals) is requested one row ahead, straight into
        // the registers it is used from; nothing else in the marker loop is a vector memory operation
        auto load_row = [&](int idx) {
            const double* sp = spill + (size_t)idx * ROW;
#ifdef CNF2_X_NOLOAD     /* timing ablation only: results are wrong */
            if (idx >= 0) return;
#endif
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const d2v v = __builtin_nontemporal_load((const d2v*)(sp + k * 128 + lane * 2));
                S.am[2 * k]     = v.x;
                S.am[2 * k + 1] = v.y;
            }
            const double2 iv = *(const double2*)(sp + 512 + 2 * s);
            S.inv_even       = iv.x;
            S.inv_odd        = iv.y;
        };
        S.inv_odd = 1.0;
        load_row(HALF ? ((last - first) >> 1) : (last - first));

        int cur_combo = 0;     // TIED: tie combination of the pass over the tile that is running
        // One marker of the backward pass.  ODD (HALF only): the marker's alpha-minus is rebuilt from the
        // row of its even neighbour.  LOADS: request the next spill row once this one has been used.
        auto marker = [&](auto odd_tag, double* row, int m, bool carried = false) {
            constexpr bool ODD = decltype(odd_tag)::value;
            const int      ml  = m - first;
            double         wj[8];
            double         aw[STOREW == 2 ? 8 : 1];                 // turn-scan mode: alpha-minus of this marker, unscaled
            const double2 r_m = *(const double2*)(row + TAB_T);     // gap m-1 -> m
            double        inv_m;
            if (ODD) {
                // odd marker: alpha-minus(m) = T( alpha-minus(m-1) * e(m-1) * inv(m-1) ), exactly the
                // forward step (cnF2freq.cpp:2238-2367); marker m-1 is the previous row of this tile
                double ep[8];
                emission_from_row(row - TS, c, ep);
                if (!TIED) {                       // (TIED: the 16 registers are needed elsewhere; the even marker forms its own)
#pragma unroll
                    for (int j = 0; j < 8; j++) S.ec[j] = ep[j];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) wj[j] = S.am[j] * ep[j];
                // the normaliser inv(m-1) is a per-chain scalar and everything below is linear in wj:
                // it is applied to the three class sums (`scale`) instead of to the eight states
                if (XPOSE) transition_xpose(wj, r_m.x, r_m.y, xb, lane);
                else transition_scaled(wj, r_m.x, r_m.y);
                if (STOREW == 2) {
#pragma unroll
                    for (int j = 0; j < 8; j++) aw[j] = wj[j];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) wj[j] *= S.b[j];
                inv_m = S.inv_odd;
            } else {
                if (STOREW == 2) {
#pragma unroll
                    for (int j = 0; j < 8; j++) aw[j] = S.am[j];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) wj[j] = S.am[j] * S.b[j];
                inv_m = S.inv_even;
                // the row is used up: request the one below it (clamped at the chromosome start, where the
                // reload is harmless) so that a whole marker, or two, of arithmetic covers the latency.
                // The empty asm pins "last use, then reload" in that order: otherwise the loads are hoisted
                // above the products, land in fresh registers and are copied (and waited for) at once.
                asm volatile("" : "+v"(wj[0]), "+v"(wj[1]), "+v"(wj[2]), "+v"(wj[3]), "+v"(wj[4]), "+v"(wj[5]),
                             "+v"(wj[6]), "+v"(wj[7]), "+v"(inv_m) : : "memory");
                const int idx = HALF ? (ml >> 1) - 1 : ml - 1;
                load_row(idx < 0 ? 0 : idx);
            }
            if (!HALF || !ODD || m == last) {     // skipped (odd) steps have normaliser 1
                // (so do the even markers between two rescalings: their stored reciprocal is exactly 1.  Updating the
                // scale only where a chain really was rescaled -- a wave-uniform branch -- was measured 2.4 % SLOWER
                // than this unconditional straight-line form)
                int ex;
                S.fmant = frexp(S.fmant * inv_m, &ex);
                S.fexpo += ex;
            }
            double n_tot = 0.0, n_a1 = 0.0, n_b1 = 0.0, n_2 = 0.0;
#ifdef CNF2_X_FUSEDACC   /* timing ablation only (tools/ablate_fused_acc.sh): results are wrong */
            double x_av[2], x_sb[2], x_t0[2], x_t1[2], x_cf[2];
#endif
            // TIED: the sums of the pair of tie combinations whose restricted tables the row holds (everything else at this
            // marker -- the posterior weights wj, the scale -- is the same for every combination)
            // (no rows asked for: no class sums, no restricted tables, no epilogue)
#pragma unroll 1
            for (int ko = 0; ko < (!ROWS ? 0 : (TIED ? 2 * TIE_KOFF : 1)); ko += TIE_KOFF)
#pragma unroll
            for (int f = 0; f < 2; f++) {
                // the lane's own line (its low bits index that half of the tables) and the line held in the registers:
                // A / B as printed, B / A in the transposed layout of an odd marker (n_a1 and n_b1 then swap roles,
                // which the class sums below do not notice: they use n_a1 + n_b1)
                constexpr bool T = XPOSE && ODD;
                const int     ia = ((T ? 1 : 0) << 5) | (f << 4) | ((T ? c.s2 : c.s1) << 3) | c.lo;
                const double* Br = row + ko + TAB_R + (((T ? 0 : 1) << 5) | (f << 4) | ((T ? c.s1 : c.s2) << 3));
                const double* B1 = row + ko + TAB_2 + (((T ? 0 : 1) << 5) | (f << 4) | ((T ? c.s1 : c.s2) << 3));
                const double  cf = row[TAB_C + f * 2 + c.s0];
                const double  av = cf * row[ko + TAB_R + ia], a1 = cf * row[ko + TAB_2 + ia];
                double        sb = 0.0, sb1 = 0.0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    sb += wj[j] * Br[j];
                    sb1 += wj[j] * B1[j];
                }
                n_tot += av * sb;
                n_a1 += a1 * sb;
                n_b1 += av * sb1;
                n_2 += a1 * sb1;
#ifdef CNF2_X_FUSEDACC
                if (WG) {
                    // the two HOMOZYGOUS probe contractions of phase B: stand-in tables of the same shape in the same row
                    const double* H0 = row + (((T ? 0 : 1) << 5) | (f << 4) | ((T ? c.s1 : c.s2) << 3));
                    double        t0 = 0.0, t1 = 0.0;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        t0 += wj[j] * H0[j];
                        t1 += wj[j] * H0[j ^ 1];
                    }
                    x_av[f] = av;
                    x_sb[f] = sb;
                    x_t0[f] = t0;
                    x_t1[f] = t1;
                    x_cf[f] = cf;
                }
#endif
            }
            const double sc0   = ODD ? xm * S.inv_even : xm;
            const double scale = chain_on ? ldexp(sc0 * S.fmant * S.bmant, xe + S.fexpo + S.bexpo) : 0.0;
            double q2 = scale * n_2;
            double q1 = scale * (n_a1 + n_b1 - 2.0 * n_2);
            double q0 = scale * (n_tot - n_a1 - n_b1 + n_2);
#ifdef CNF2_X_FUSEDACC
            if (WG && (!TIED || cur_combo == 0)) {
                // phase B of acc_tile_kernel formed here: v (sum over s2), z (over s0 and s2), u (over the 32 lanes of a half)
                double*   wp = p.wbuf + ((size_t)job * p.wstride + ml) * 160;
                const int e0 = (c.s1 << 3) | c.lo;
                double    uu[16];
#pragma unroll
                for (int f = 0; f < 2; f++) {
                    const double k = scale * x_cf[f];
                    double       v = k * x_sb[f], z0 = k * x_t0[f], z1 = k * x_t1[f];
                    v += lane_xor32(v);
                    z0 += lane_xor32(z0);
                    z1 += lane_xor32(z1);
                    z0 += lane_xor8(z0);
                    z1 += lane_xor8(z1);
                    if (c.s2 == 0) wp[(f * 2 + c.s0) * 16 + e0] = v;
                    if (c.s2 == 0 && c.s0 == 0) {
                        wp[96 + (f * 2 + 0) * 16 + e0] = z0;
                        wp[96 + (f * 2 + 1) * 16 + e0] = z1;
                    }
                    const double ka = scale * x_av[f];
#pragma unroll
                    for (int j = 0; j < 8; j++) uu[f * 8 + j] = wj[j] * ka;
                }
                auto halve = [&](double a, double b, bool bit, int which) CNF2_LI_K {
                    const double keep = bit ? b : a, send = bit ? a : b;
                    return keep + (which == 0 ? lane_xor1(send) : (which == 1 ? lane_xor2(send) : (which == 2 ? lane_xor4(send) : lane_xor8(send))));
                };
                double h8[8], h4[4], h2[2];
#pragma unroll
                for (int k = 0; k < 8; k++) h8[k] = halve(uu[2 * k], uu[2 * k + 1], (lane & 1) != 0, 0);
#pragma unroll
                for (int k = 0; k < 4; k++) h4[k] = halve(h8[2 * k], h8[2 * k + 1], (lane & 2) != 0, 1);
#pragma unroll
                for (int k = 0; k < 2; k++) h2[k] = halve(h4[2 * k], h4[2 * k + 1], (lane & 4) != 0, 2);
                double h1v = halve(h2[0], h2[1], (lane & 8) != 0, 3);
                h1v += lane_xor16(h1v);
                if ((lane & 16) == 0) wp[64 + ((lane >> 3) & 1) * 16 + (c.s2 << 3) + (lane & 7)] = h1v;
            }
#else
            if (WG && (!TIED || cur_combo == 0)) {
                // accumulate mode: wg(s, g) = exp(scales - factor) alphaminus beta for the batched HOT LOOP 2 kernel
                // (the same in every tie combination)
                double* wp = p.wbuf + ((size_t)job * p.wstride + ml) * 512;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const d2v v = {scale != 0.0 ? scale * wj[2 * k] : 0.0, scale != 0.0 ? scale * wj[2 * k + 1] : 0.0};
                    __builtin_nontemporal_store(v, (d2v*)(wp + k * 128 + lane * 2));
                }
            }
#endif
            // this marker's own emission is only needed for the beta step: formed here, after the sums, so
            // that it does not occupy registers across them
            double e[8];
            if (XPOSE && ODD) emission_from_row_t(row, c, e);
            else if (!ODD && carried) {
#pragma unroll
                for (int j = 0; j < 8; j++) e[j] = S.ec[j];
            }
            else emission_from_row(row, c, e);
            if (STOREW == 2) {
                // turn-scan mode: A = alphaminus e and B = beta as held here, with the log2 of the scales that make them
                // absolute (alphaminus: Fpre, and the stored normaliser of the even neighbour for a rebuilt odd marker)
                double* wp = p.wbuf + ((size_t)job * p.wstride + ml) * CNF2_TURN_ROW;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const d2v va = {aw[2 * k] * e[2 * k], aw[2 * k + 1] * e[2 * k + 1]};
                    const d2v vb = {S.b[2 * k], S.b[2 * k + 1]};
                    __builtin_nontemporal_store(va, (d2v*)(wp + k * 128 + lane * 2));
                    __builtin_nontemporal_store(vb, (d2v*)(wp + 512 + k * 128 + lane * 2));
                }
                if (c.lo == 0) {
                    *(double2*)(wp + 1024 + 4 * s)     = make_double2((ODD ? S.inv_even : 1.0) * S.fmant, (double)S.fexpo);
                    *(double2*)(wp + 1024 + 4 * s + 2) = make_double2(S.bmant, (double)S.bexpo);
                }
            }
            // every lane parks its three class partials in this marker's row (all of it is dead by now:
            // tables, root weights and gap factors have been read); the tile epilogue sums them
            if (ROWS) {
                wave_lds_fence();
                row[lane]       = q0;
                row[64 + lane]  = q1;
                row[128 + lane] = q2;
            }
            // beta(m-1) = T( beta(m) * e(m) ); at the first marker the result is never used
#pragma unroll
            for (int j = 0; j < 8; j++) S.b[j] *= e[j];
            // (deferring this normaliser like the forward one was measured slower here: the row scale and the
            // emission of the next step would both wait for the reciprocal)
            if (!HALF || (!ODD && (ml & bmask) == 0)) {
                const double bsum = scale_chain(S.b, &S.bmant, &S.bexpo, &S.bdead);
                if (HALF && __ballot(bsum > 0.0 && bsum * CNF2_RESCALE_GUARD < 1.0)) bmask = 1;
            }
            if (XPOSE) transition_xpose(S.b, r_m.x, r_m.y, xb, lane);
            else transition_scaled(S.b, r_m.x, r_m.y);
        };
        load_raw<-1>(p, c, first + (ntile - 1) * 8, first, last, &raw);
        for (int t = ntile - 1; t >= 0; t--) {
            const int m0 = first + t * 8;
            // TIED: the state at the tile's start, restored for every combination: beta, the mantissas and the exponents from
            // LDS, the spill row in flight by asking for it again
            const int bmask0 = bmask;
            const bool bdead0 = S.bdead;
            if (TIED) {
                double* sv = tsave[wib] + lane;
#pragma unroll
                for (int j = 0; j < 8; j++) sv[j * 64] = S.b[j];
                sv[8 * 64] = S.bmant;
                sv[9 * 64] = S.fmant;
                sv[10 * 64] = __hiloint2double(S.bexpo, S.fexpo);
                double zero = 0.0;
                asm volatile("" : "+v"(zero));          // formed here: the compiler would keep (and spill) a 0.0 for the whole kernel
                if (lane < 24) tsum[wib][lane] = zero;
            }
          // TIED: the tie combinations two at a time (there are 2, 4 or 8): one pass over the tile per pair
          for (int combo = 0; combo < n_combo; combo += (TIED ? 2 : 1)) {
            const int tp = (int8_t)(my_ties & 255), tt = (int8_t)((my_ties >> 8) & 255), to = (int8_t)((my_ties >> 16) & 255);
            auto forces = [&](int k) {                                       // = part_forces(w, c.part, k, &c.pc)
                c.pc.force_par = tp < 0 ? -1 : ((k >> tp) & 1);
                c.pc.force_tr  = tt < 0 ? -1 : ((k >> tt) & 1);
                c.pc.force_ot  = to < 0 ? -1 : ((k >> to) & 1);
            };
            if (TIED) {
                cur_combo = combo;
                // the spill row of the tile's top marker once more (the first pass has it in flight from the tile
                // above): asked for here, so that the producer's work covers the latency
                if (combo > 0) load_row(HALF ? ((((m0 + 7 < last) ? m0 + 7 : last) - first) >> 1) : (((m0 + 7 < last) ? m0 + 7 : last) - first));
                forces(combo);
            }
#ifdef CNF2_X_NOPRODUCE  /* timing ablation only: results are wrong */
            if (t == ntile - 1)
#endif
            {
                produce_tile<ROWS, TIED, TS>(p, c, tab, m0, last, raw, hom);
                if (TIED) {
                    forces(combo + 1);
                    produce_tile<true, TIED, TS, TIE_KOFF>(p, c, tab, m0, last, raw, hom);
                }
            }
            wave_lds_fence();
            const int mend = (m0 + 7 < last) ? m0 + 7 : last;
            if (TIED) {
                // (for the first combination too: nothing of the backward state is then live across the producer, whose tie
                // form needs the registers)
                const double* sv = tsave[wib] + lane;
#pragma unroll
                for (int j = 0; j < 8; j++) S.b[j] = sv[j * 64];
                S.bmant = sv[8 * 64];
                S.fmant = sv[9 * 64];
                S.bexpo = __double2hiint(sv[10 * 64]);
                S.fexpo = __double2loint(sv[10 * 64]);
                S.bdead = bdead0;
                bmask   = bmask0;
            }
            int       i    = mend - m0;                 // local index; its parity is the parity of m - first
            if (HALF) {
                if (!(i & 1)) {                         // an even top marker: only the last tile of a chromosome
                    marker(even_t(), tab + i * TS, m0 + i);
                    i--;
                }
                for (; i >= 1; i -= 2) {
                    marker(odd_t(), tab + i * TS, m0 + i);
                    marker(even_t(), tab + (i - 1) * TS, m0 + i - 1, !TIED);
                }
            } else {
                for (; i >= 0; i--) marker(even_t(), tab + i * TS, m0 + i);
            }
            wave_lds_fence();
            // next tile's inputs: requested here, before the epilogue's LDS work (holding them across the
            // whole marker loop costs more in registers than the extra latency it hides; measured)
#ifndef CNF2_X_NOPRODUCE
            // (TIED: the tile's own inputs once more for its next combination, rather than 29 registers held across the
            // marker loop; the tile's start goes through an empty asm so that the eight addresses are formed here, not kept)
            if (TIED) {
                int m0x = (combo + 2 < n_combo) ? m0 : m0 - 8;
                asm volatile("" : "+s"(m0x));
                if (t > 0 || combo + 2 < n_combo) load_raw<-1>(p, c, m0x, first, last, &raw);
            } else if (t > 0) load_raw<-1>(p, c, m0 - 8, first, last, &raw);
#endif
            // tile epilogue: lanes (marker mi = lane >> 3, eighth sub = lane & 7) add up the 3 x 64 partials
            // of the tile's markers, lane sub == 0 normalises and stores the row
            if (ROWS) {
                const int     mi2 = lane >> 3, sub = lane & 7;
                const double* red = tab + mi2 * TS + sub * 8;
                double        d0 = 0.0, d1 = 0.0, d2 = 0.0;
#pragma unroll
                for (int i2 = 0; i2 < 8; i2++) {
                    d0 += red[i2];
                    d1 += red[64 + i2];
                    d2 += red[128 + i2];
                }
                d0 = chain_sum(d0);
                d1 = chain_sum(d1);
                d2 = chain_sum(d2);
                if (TIED) {
                    // add this combination's class sums to the tile's; the rows are written after the last one
                    if (sub == 0) {
                        d0 += tsum[wib][mi2 * 3 + 0];
                        d1 += tsum[wib][mi2 * 3 + 1];
                        d2 += tsum[wib][mi2 * 3 + 2];
                        tsum[wib][mi2 * 3 + 0] = d0;
                        tsum[wib][mi2 * 3 + 1] = d1;
                        tsum[wib][mi2 * 3 + 2] = d2;
                    }
                }
                // class 0 and 1 are formed by inclusion-exclusion of non-negative sums: a true zero can come
                // out as -1e-16; the reference adds non-negative terms (cnF2freq.cpp:3536), so clamp
                d0 = fmax(d0, 0.0);
                d1 = fmax(d1, 0.0);
                d2 = fmax(d2, 0.0);
                if (sub == 0 && m0 + mi2 <= last && combo + (TIED ? 2 : 1) >= n_combo) {
                    if (!(p.flags & KP_RAW_DOSAGE)) {
                        const double tsum = d0 + d1 + d2;
                        const double inv  = tsum > 0.0 ? 1.0 / tsum : 0.0;
                        d0 *= inv;
                        d1 *= inv;
                        d2 *= inv;
                    }
                    double* out = p.dosage + ((size_t)jb.ind * p.n_markers + (m0 + mi2)) * 3;
                    out[0] = d0;
                    out[1] = d1;
                    out[2] = d2;
                }
            }
            wave_lds_fence();
          }
        }
    }
    if (stamp && fresh_lane() == 0) {
        p.clock_out[0] = __builtin_readcyclecounter() - t_shader;
        p.clock_out[1] = wall_clock64() - t_wall;
    }
}

// The logarithms of the likelihoods a fast-kernel launch left as mantissa and binary exponent: one thread per (job, shift
// mode), thread 0 of a job also the job's total.  Adds the chromosome's dropped butterfly constants (chrom_logk).
__global__ __launch_bounds__(256) void likelihood_logs_kernel(KernelParams p)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.n_jobs * 8) return;
    const Job    jb   = p.jobs[t >> 3];
    const int    s    = t & 7;
    const size_t e    = (size_t)jb.ind * p.n_chrom + jb.chrom;
    const double logk = p.chrom_logk[jb.chrom];
    const double ln2  = 0.69314718055994530942;
    const int    fe   = p.fexp[e * 8 + s];
    double       fs;
    if (fe == CNF2_LEXP_IGNORED) fs = CNF2_IGNORED_D;
    else if (fe == CNF2_LEXP_DEAD) fs = (double)CNF2_MINFACTOR_F;
    else fs = log(p.factors[e * 8 + s]) + (double)fe * ln2 + logk;
    p.factors[e * 8 + s] = fs;
    if (s == 0) {
        const int    le = p.lexp[e];
        const double T  = p.loglik[e];
        p.loglik[e] = (le == CNF2_LEXP_DEAD) ? (double)CNF2_MINFACTOR_F + log(T) : log(T) + (double)le * ln2 + logk;
    }
}
static void launch_likelihood_logs(const KernelParams& p, hipStream_t stream)
{
    if (p.n_jobs > 0) hipLaunchKernelGGL(likelihood_logs_kernel, dim3((p.n_jobs * 8 + 255) / 256), dim3(256), 0, stream, p);
}

// =====================================================================================
// Packed kernel: four jobs per wavefront for windows whose BOTH parents are homozygous with equal sure
// at every marker (e.g. the private empty F1 parents of an F2, cnF2freq.cpp:6515-6527).  For such a
// parent the two allele indices are interchangeable, its phase weight is 0/1 (cnF2freq.cpp:1235-1239)
// and the unrestricted table does not depend on its shift bit: shift modes that differ in shift bits
// 1, 2 carry bit-identical alpha and beta.  Only the two modes s0 = 0, 1 are swept; the per-locus row
// sums the restricted / class-2 tables over both values of each parent's shift bit, which is exactly the
// sum of the four copies.  Lane layout: chain = lane >> 3 = (job in wave) << 1 | s0, the rest as in
// fb_fast_kernel; a tile is 2 markers x 4 jobs (same 8 table rows of LDS, row = marker << 2 | job).
// Always half-spill.  Outputs are written for all 8 shift modes.
// =====================================================================================
__global__ __launch_bounds__(CNF2_BLOCK, 2) void fb_packed_kernel(KernelParams p)
{
    constexpr int ROW = 528;
    __shared__ __attribute__((aligned(16))) double lds[CNF2_WAVES_PER_BLOCK][8 * TAB_STRIDE];

    const int lane  = threadIdx.x & 63;
    const int wib   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave  = blockIdx.x * CNF2_WAVES_PER_BLOCK + wib;
    const int nwave = gridDim.x * CNF2_WAVES_PER_BLOCK;
    double*   tab   = lds[wib];
    double*   spill = p.spill + (size_t)wave * p.spill_stride;

    // (jobs from the launch's counter, as in fb_fast_kernel)
    auto take_job = [&](int strided) {
        if (!p.job_next) return strided;
        int j = 0;
        if (lane == 0) j = atomicAdd(p.job_next, 1);
        return __builtin_amdgcn_readfirstlane(j);
    };
    for (int job = take_job(wave); job < p.n_pjobs; job = take_job(job + nwave)) {
        const PackedJob pj = p.pjobs[job];
        // producer role: part x (marker in tile, job)
        FastCtx c;
        c.part = lane >> 3;
        c.mi   = lane & 7;                                   // table row = marker_in_tile << 2 | job
        const int pmt = c.mi >> 2;
        {
            const Window wp = p.windows[pj.ind[c.mi & 3]];
            int32_t      rp, rt, ro;
            make_part(wp, c.part, &c.pc, &rp, &rt, &ro);
            c.slot_a = lane >> 3;
            c.row_a  = (c.slot_a < 7 && wp.row[c.slot_a] >= 0) ? wp.row[c.slot_a] : 0;
        }
        c.idx_base = part_entry_index(c.part, 0);
        c.idx_k01  = part_entry_index(c.part, 1) - c.idx_base;
        c.idx_k10  = part_entry_index(c.part, 2) - c.idx_base;
        // consumer role: chain = job << 1 | s0; shift bits 1, 2 are represented by 0
        const int s    = lane >> 3;
        const int cjob = s >> 1;
        const int ind  = pj.ind[cjob];
        c.s0 = s & 1;
        c.s1 = 0;
        c.s2 = 0;
        c.lo = state_lo(lane);
        c.active = true;                                     // eligibility: shiftignore == 0, shiftend == 8
        const int first = pj.first, last = pj.last;
        const int ntile = (last - first + 2) >> 1;
        if (p.path_log && (lane & 15) == 0) p.path_log[(size_t)ind * p.n_chrom + pj.chrom] = PATH_PACKED | pj.homleaf;
        double*   myrow0 = tab + cjob * TAB_STRIDE;          // this lane's job, marker 0 of the tile
        double*   myrow1 = myrow0 + 4 * TAB_STRIDE;

        using odd_t  = std::integral_constant<bool, true>;
        using even_t = std::integral_constant<bool, false>;

        // ---------------------------------------------------------------- forward
        double a[8];
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = 1.0 / 64.0;
        double mant = 1.0;
        int    expo = 0;
        bool   dead = false;
        double pend = 1.0;
        auto fwd_step = [&](auto odd_tag, const double* row, int m) {
            constexpr bool ODD = decltype(odd_tag)::value;
            double         e[8];
            emission_from_row(row, c, e, pend);
            const double2 r  = *(const double2*)(row + TAB_T);
            const int     ml = m - first;
            double*       sp = spill + (size_t)(ml >> 1) * ROW;
            if (!ODD) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const d2v v = {a[2 * k], a[2 * k + 1]};
                    __builtin_nontemporal_store(v, (d2v*)(sp + k * 128 + lane * 2));
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] *= e[j];
            pend = 1.0;
            if (!ODD || m == last) {
                const double inv = chain_normaliser(a, &mant, &expo, &dead);
                pend             = inv;
                if (c.lo == 0) sp[512 + 2 * s + (ODD ? 1 : 0)] = inv;
            }
            if (m < last) transition_scaled(a, r.x, r.y);
        };
        RawOne raw;
        load_raw1_at<0>(p, c, first + pmt, first, last, &raw);
        for (int t = 0; t < ntile; t++) {
            const int m0 = first + t * 2;
            if (pj.homleaf) produce_row2<false, true, true>(c, tab, m0 + pmt <= last, raw);
            else produce_row2<false, true, false>(c, tab, m0 + pmt <= last, raw);
            if (t + 1 < ntile) load_raw1_at<0>(p, c, m0 + 2 + pmt, first, last, &raw);
            wave_lds_fence();
            fwd_step(even_t(), myrow0, m0);
            if (m0 < last) fwd_step(odd_t(), myrow1, m0 + 1);
            wave_lds_fence();
        }

        // ---------------------------------------------------------------- likelihoods
        // every mode s has the likelihood of mode s & 1; the log-sum-exp over 8 modes adds the same two
        // terms four times (the pairwise tree of fb_fast_kernel doubles exactly, so the bits agree)
        const double logk = p.chrom_logk[pj.chrom];
        const double fs   = dead ? (double)CNF2_MINFACTOR_F : (log(mant) + (double)expo * 0.69314718055994530942);
        double fmaxv = fmax(fs, lane_xor8(fs));
        fmaxv        = fmax(fmaxv, -1e15);
        const double term = exp(fs - fmaxv);
        const double real = 4.0 * (term + lane_xor8(term));
        const double factor = fmaxv + log(real);
        if (c.lo == 0) {
            double* fo = p.factors + ((size_t)ind * p.n_chrom + pj.chrom) * 8 + c.s0;
            const double v = !dead ? fs + logk : fs;
            fo[0] = v;
            fo[2] = v;
            fo[4] = v;
            fo[6] = v;
        }
        if ((lane & 15) == 0) p.loglik[(size_t)ind * p.n_chrom + pj.chrom] = (fmaxv > -1e14) ? factor + logk : factor;
        const bool skip = isnan(factor) || factor < (double)CNF2_MINFACTOR_F;     // cnF2freq.cpp:5403
        if (p.flags & KP_NO_DOSAGE) continue;

        // ---------------------------------------------------------------- backward + rows
        BwdState S;
#pragma unroll
        for (int j = 0; j < 8; j++) S.b[j] = 1.0;
        S.bmant = 1.0;
        S.bexpo = 0;
        S.bdead = false;
        S.fmant = mant;
        S.fexpo = expo;
        const double nf  = -factor * 1.4426950408889634074;
        const double nfk = floor(nf);
        const double xm  = exp2(nf - nfk);
        const int    xe  = (int)nfk;
        const bool   chain_on = !skip && !dead && !(factor - fs > 40.0);          // cnF2freq.cpp:5420-5421
        auto load_row = [&](int idx) {
            const double* sp = spill + (size_t)idx * ROW;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const d2v v = __builtin_nontemporal_load((const d2v*)(sp + k * 128 + lane * 2));
                S.am[2 * k]     = v.x;
                S.am[2 * k + 1] = v.y;
            }
            const double2 iv = *(const double2*)(sp + 512 + 2 * s);
            S.inv_even       = iv.x;
            S.inv_odd        = iv.y;
        };
        S.inv_odd = 1.0;
        load_row((last - first) >> 1);

        auto marker = [&](auto odd_tag, double* row, int m) {
            constexpr bool ODD = decltype(odd_tag)::value;
            const int      ml  = m - first;
            double         wj[8];
            const double2 r_m = *(const double2*)(row + TAB_T);     // gap m-1 -> m
            double        inv_m;
            if (ODD) {
                double ep[8];
                emission_from_row(row - 4 * TAB_STRIDE, c, ep);     // the even marker of the pair: same job, marker 0
#pragma unroll
                for (int j = 0; j < 8; j++) wj[j] = S.am[j] * ep[j];
                transition_scaled(wj, r_m.x, r_m.y);
#pragma unroll
                for (int j = 0; j < 8; j++) wj[j] *= S.b[j];
                inv_m = S.inv_odd;
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) wj[j] = S.am[j] * S.b[j];
                inv_m = S.inv_even;
                asm volatile("" : "+v"(wj[0]), "+v"(wj[1]), "+v"(wj[2]), "+v"(wj[3]), "+v"(wj[4]), "+v"(wj[5]),
                             "+v"(wj[6]), "+v"(wj[7]), "+v"(inv_m) : : "memory");
                const int idx = (ml >> 1) - 1;
                load_row(idx < 0 ? 0 : idx);
            }
            if (!ODD || m == last) {
                int ex;
                S.fmant = frexp(S.fmant * inv_m, &ex);
                S.fexpo += ex;
            }
            // class sums over the four copies (s1, s2) of this mode: tables summed over the parents' shift bits
            double n_tot = 0.0, n_a1 = 0.0, n_b1 = 0.0, n_2 = 0.0;
#pragma unroll
            for (int f = 0; f < 2; f++) {
                const int     ia = (0 << 5) | (f << 4) | c.lo;
                const double* Br = row + TAB_R + ((1 << 5) | (f << 4));
                const double* B1 = row + TAB_2 + ((1 << 5) | (f << 4));
                const double  cf = row[TAB_C + f * 2 + c.s0];
                const double  av = cf * (row[TAB_R + ia] + row[TAB_R + ia + 8]);
                const double  a1 = cf * (row[TAB_2 + ia] + row[TAB_2 + ia + 8]);
                double        sb = 0.0, sb1 = 0.0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    sb += wj[j] * (Br[j] + Br[8 + j]);
                    sb1 += wj[j] * (B1[j] + B1[8 + j]);
                }
                n_tot += av * sb;
                n_a1 += a1 * sb;
                n_b1 += av * sb1;
                n_2 += a1 * sb1;
            }
            const double sc0   = ODD ? xm * S.inv_even : xm;
            const double scale = chain_on ? ldexp(sc0 * S.fmant * S.bmant, xe + S.fexpo + S.bexpo) : 0.0;
            const double q2 = scale * n_2;
            const double q1 = scale * (n_a1 + n_b1 - 2.0 * n_2);
            const double q0 = scale * (n_tot - n_a1 - n_b1 + n_2);
            double e[8];
            emission_from_row(row, c, e);
            // the 16 lanes of this job park their partials in the job's own row of this marker
            wave_lds_fence();
            row[lane & 15]         = q0;
            row[64 + (lane & 15)]  = q1;
            row[128 + (lane & 15)] = q2;
#pragma unroll
            for (int j = 0; j < 8; j++) S.b[j] *= e[j];
            if (!ODD) scale_chain(S.b, &S.bmant, &S.bexpo, &S.bdead);
            transition_scaled(S.b, r_m.x, r_m.y);
        };
        load_raw1_at<-1>(p, c, first + (ntile - 1) * 2 + pmt, first, last, &raw);
        for (int t = ntile - 1; t >= 0; t--) {
            const int m0 = first + t * 2;
            if (pj.homleaf) produce_row2<true, true, true>(c, tab, m0 + pmt <= last, raw);
            else produce_row2<true, true, false>(c, tab, m0 + pmt <= last, raw);
            wave_lds_fence();
            if (m0 < last) marker(odd_t(), myrow1, m0 + 1);
            marker(even_t(), myrow0, m0);
            wave_lds_fence();
            if (t > 0) load_raw1_at<-1>(p, c, m0 - 2 + pmt, first, last, &raw);
            // tile epilogue: lane = table row (marker << 2 | job) x eighth: 3 x 16 partials per row
            {
                const int     r8 = lane >> 3, sub = lane & 7;
                const double* red = tab + r8 * TAB_STRIDE + sub * 2;
                double d0 = red[0] + red[1], d1 = red[64] + red[65], d2 = red[128] + red[129];
                d0 = fmax(chain_sum(d0), 0.0);
                d1 = fmax(chain_sum(d1), 0.0);
                d2 = fmax(chain_sum(d2), 0.0);
                const int mm = m0 + (r8 >> 2);
                if (sub == 0 && mm <= last) {
                    if (!(p.flags & KP_RAW_DOSAGE)) {
                        const double tsum = d0 + d1 + d2;
                        const double inv  = tsum > 0.0 ? 1.0 / tsum : 0.0;
                        d0 *= inv;
                        d1 *= inv;
                        d2 *= inv;
                    }
                    double* out = p.dosage + ((size_t)pj.ind[r8 & 3] * p.n_markers + mm) * 3;
                    out[0] = d0;
                    out[1] = d1;
                    out[2] = d2;
                }
            }
            wave_lds_fence();
        }
    }
}

// per-row flag: every marker homozygous (or doubly unknown) with equal sure -- such an ancestor
// can never make a tie group active (cnF2freq.cpp:3488: its zero-weight paths are the pruned ones)
__global__ __launch_bounds__(256) void row_flags_kernel(const uint8_t* allele8, const double2* sure, int n_markers,
                                                        uint8_t* flags)
{
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * n_markers;
    int          mine = 0;
    for (int m = threadIdx.x; m < n_markers; m += blockDim.x) {
        const uint8_t ap = allele8[base + m];
        const double2 s  = sure[base + m];
        if ((ap & 15) != (ap >> 4) || s.x != s.y) mine = 1;
    }
    if (mine) bad = 1;
    __syncthreads();
    if (threadIdx.x == 0) flags[blockIdx.x] = bad ? 0 : 1;
}

// Parity hook: path-free emission e_s(g) of one (individual, marker) for the 8 shift modes,
// through exactly the producer/consumer code of the sweep.
__global__ __launch_bounds__(64) void emission_kernel(KernelParams p, int ind, int marker, double* out)
{
    __shared__ double tab[64];
    const int    lane = threadIdx.x;
    const Window w    = p.windows[ind];
    LaneCtx      c;
    make_lane(w, lane, &c.L);
    c.row_root   = w.row[0];
    c.root_attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    const int s  = lane >> 3;
    c.s0 = s & 1;
    c.s1 = (s >> 1) & 1;
    c.s2 = (s >> 2) & 1;
    c.lo = state_lo(lane);
    c.active = true;
    c.n_combo = 1;
    const Slot root = load_slot(p, c.row_root, marker);
    LineTerms  T;
    tab[lane] = produce_entry(p, c, root, marker, &T);
    wave_lds_fence();
    double c0, c1, e[8];
    root_weights(c, root, &c0, &c1);
    emission_from_table(tab, c, c0, c1, e);
#pragma unroll
    for (int j = 0; j < 8; j++) out[s * 64 + j * 8 + c.lo] = e[j];
}

// Self-test of the lane-exchange helpers (tests/test_gpu_parity.py): out[k][lane] = value
// received by `lane` for xor distance 1<<k.
__global__ __launch_bounds__(64) void xor_selftest_kernel(double* out)
{
    const int lane = threadIdx.x;
    double    v    = 1000.0 + lane;
    out[0 * 64 + lane] = lane_xor1(v);
    out[1 * 64 + lane] = lane_xor2(v);
    out[2 * 64 + lane] = lane_xor4(v);
    out[3 * 64 + lane] = lane_xor8(v);
    out[4 * 64 + lane] = lane_xor16(v);
    out[5 * 64 + lane] = lane_xor32(v);
}


// =====================================================================================
// Stage-2 consumers of the alpha/beta store, parity level:
// they read the reference-layout store that fb_kernel<true> leaves for ONE individual and
// chromosome and answer the queries of HOT LOOP 2 / 3 in bulk.  Not tuned; one thread per answer.
// =====================================================================================
__device__ __forceinline__ double s2_fw(const Stage2Params& q, int s, int ml, int slot, int g)
{
    return q.fwbw[(((size_t)s * q.len + ml) * 3 + slot) * 64 + g];
}
__device__ __forceinline__ double s2_ff(const Stage2Params& q, int s, int ml, int slot)
{
    return q.fwbwfactors[((size_t)s * q.len + ml) * 3 + slot];
}

// Terms of the two lines of state g under shift mode s for root allele f.
__device__ __forceinline__ void s2_lines(const Stage2Params& q, const Window& w, int m, int g, int s, int f,
                                         LaneJob* L0, LaneJob* L1, LineTerms* T0, LineTerms* T1, double* cf, bool* attop)
{
    const KernelParams& p = q.kp;
    const Slot root = load_slot(p, w.row[0], m);
    *attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    RootTerms R;
    root_terms(root, *attop, f, &R);
    *cf = R.cbase * phase_weight(root, f ^ (s & 1));
    make_lane(w, (0 << 5) | (f << 4) | (((s >> 1) & 1) << 3) | (g & 7), L0);
    make_lane(w, (1 << 5) | (f << 4) | (((s >> 2) & 1) << 3) | (g >> 3), L1);
    line_terms(L0->cfg, load_slot(p, L0->row_par, m), load_slot(p, L0->row_tr, m), load_slot(p, L0->row_ot, m),
               R.inmv0, R.sv0, R.inmv0 == 2, T0);
    line_terms(L1->cfg, load_slot(p, L1->row_par, m), load_slot(p, L1->row_tr, m), load_slot(p, L1->row_ot, m),
               R.inmv1, R.sv1, false, T1);
}

// One allele assignment (path) of a line: parent allele fp, traced / other grandparent alleles.
__device__ __forceinline__ double s2_path_term(const LineCfg& c, const LineTerms& T, int fp, int fg_tr, int fg_ot)
{
    const bool par_line = !(c.par & SLOT_PRESENT) || (c.par & SLOT_FOUNDER);
    if (!(c.par & SLOT_PRESENT)) fp = 0;                 // 1 + secondval, whatever the path bits say
    if (par_line || !(c.tr & SLOT_PRESENT)) fg_tr = 0;
    if (par_line || !(c.ot & SLOT_PRESENT)) fg_ot = 0;
    return (T.base[fp] * T.ot[fp][fg_ot]) * T.tr[fp][fg_tr];
}

// val(s, g, flag2) = exp(doanalyze(classicstop(q, g), flag2) - factor) of cnF2freq.cpp:5499-5508 for one
// marker: out[s][g][flag2]; 0 where the reference would not count it (not finite or <= -200).
__global__ __launch_bounds__(128) void locked_query_kernel(Stage2Params q, int marker, double* out)
{
    const int flag2 = threadIdx.x;            // 0..127
    const int g     = blockIdx.x & 63;
    const int s     = blockIdx.x >> 6;
    const Window w  = q.kp.windows[0];
    const int ml    = marker - q.first;
    const int f     = flag2 & 1;
    LaneJob   L0, L1;
    LineTerms T0, T1;
    double    cf;
    bool      attop;
    s2_lines(q, w, marker, g, s, f, &L0, &L1, &T0, &T1, &cf, &attop);
    double e;
    if (attop) e = cf;
    else {
        const int fp0 = (flag2 >> 1) & 1, fp1 = (flag2 >> 4) & 1;
        const int a0 = (flag2 >> (2 + L0.cfg.firstpar)) & 1, o0 = (flag2 >> (2 + (L0.cfg.firstpar ^ 1))) & 1;
        const int a1 = (flag2 >> (5 + L1.cfg.firstpar)) & 1, o1 = (flag2 >> (5 + (L1.cfg.firstpar ^ 1))) & 1;
        e = (cf * s2_path_term(L1.cfg, T1, fp1, a1, o1)) * s2_path_term(L0.cfg, T0, fp0, a0, o0);
    }
    const double factor = q.loglik[0];
    const double am = s2_fw(q, s, ml, 0, g), be = s2_fw(q, s, ml, 1, g);
    // cnF2freq.cpp:1960-2016: alpha-minus filtered to g, emission of the path, beta of g
    double lv = s2_ff(q, s, ml, 0) + log(am * e) + s2_ff(q, s, ml, 1) + log(be) - factor;
    double v  = (isfinite(lv) && lv > -200.0) ? exp(lv) : 0.0;
    out[((size_t)s * 64 + g) * 128 + flag2] = v;
}

// Parity hook: path-resolved emission e_s(g, flag2) = trackpossible<0,0>(..., 2g, flag2, s) of cnF2freq.cpp:1380-1385
// for one (individual, marker): out[s][g][flag2], through the line terms the stage-2 kernels use.
__global__ __launch_bounds__(128) void emission_paths_kernel(Stage2Params q, int marker, double* out)
{
    const int flag2 = threadIdx.x;            // 0..127
    const int g     = blockIdx.x & 63;
    const int s     = blockIdx.x >> 6;
    const Window w  = q.kp.windows[0];
    const int f     = flag2 & 1;
    LaneJob   L0, L1;
    LineTerms T0, T1;
    double    cf;
    bool      attop;
    s2_lines(q, w, marker, g, s, f, &L0, &L1, &T0, &T1, &cf, &attop);
    double e;
    if (attop) e = cf;
    else {
        const int fp0 = (flag2 >> 1) & 1, fp1 = (flag2 >> 4) & 1;
        const int a0 = (flag2 >> (2 + L0.cfg.firstpar)) & 1, o0 = (flag2 >> (2 + (L0.cfg.firstpar ^ 1))) & 1;
        const int a1 = (flag2 >> (5 + L1.cfg.firstpar)) & 1, o1 = (flag2 >> (5 + (L1.cfg.firstpar ^ 1))) & 1;
        e = (cf * s2_path_term(L1.cfg, T1, fp1, a1, o1)) * s2_path_term(L0.cfg, T0, fp0, a0, o0);
    }
    out[((size_t)s * 64 + g) * 128 + flag2] = e;
}
void launch_emission_paths(const KernelParams& p, int marker, double* out, hipStream_t stream)
{
    Stage2Params q;
    memset(&q, 0, sizeof(q));
    q.kp = p;
    hipLaunchKernelGGL(emission_paths_kernel, dim3(512), dim3(128), 0, stream, q, marker, out);
}

// rawervals[turn][s] of cnF2freq.cpp:5686-5752 for one marker (aroundturner, cnF2freq.cpp:498-554):
// alpha (after emission) of mode s, states XOR-permuted by turn & 54, times beta of the turned mode.
__global__ __launch_bounds__(64) void turn_scan_kernel(Stage2Params q, int marker, double* out)
{
    const int k    = threadIdx.x;
    const int s    = blockIdx.x & 7;
    const int turn = blockIdx.x >> 3;
    const int ml   = marker - q.first;
    const int xorturn = turn & 54;                                                      // cnF2freq.cpp:508
    const int shiftx  = (turn >> 6) | ((turn & 1) ? 2 : 0) | ((turn & 8) ? 4 : 0);     // cnF2freq.cpp:509-510
    const int s2      = s ^ shiftx;
    double v = s2_fw(q, s, ml, 2, k ^ xorturn) * s2_fw(q, s2, ml, 1, k);
    v += lane_xor1(v);
    v += lane_xor2(v);
    v += dpp_mov_all<0x141>(v);
    v += lane_xor8(v);
    v += lane_xor16(v);
    v += lane_xor32(v);
    if (k == 0) {
        double r = (v > 0.0) ? s2_ff(q, s, ml, 2) + s2_ff(q, s2, ml, 1) + log(v) : (double)CNF2_MINFACTOR_F;
        out[turn * 8 + s] = r - q.loglik[0];
    }
}

// statereporter row (cnF2freq.cpp:3540-3546): sum of val over shift modes and paths, by state g,
// for every marker of the chromosome: out[len][64] (un-normalised like the reference's reporter).
__global__ __launch_bounds__(64) void state_rows_kernel(Stage2Params q, uint32_t flags, double* out)
{
    const int g  = threadIdx.x;
    const int ml = blockIdx.x;
    const int m  = q.first + ml;
    const Window w = q.kp.windows[0];
    const double factor = q.loglik[0];
    const int n_combo = (flags & KP_NO_TIES) ? 1 : (1 << w.n_groups);
    double acc = 0.0;
    for (int s = 0; s < 8; s++) {
        if ((s & w.shiftignore) || s >= w.shiftend) continue;
        if (factor - q.factors[s] > 40.0) continue;                                    // cnF2freq.cpp:5421
        double e = 0.0;
        for (int f = 0; f < 2; f++) {
            LaneJob   L0, L1;
            LineTerms T0, T1;
            double    cf;
            bool      attop;
            s2_lines(q, w, m, g, s, f, &L0, &L1, &T0, &T1, &cf, &attop);
            if (attop) {
                e += cf;
                continue;
            }
            for (int combo = 0; combo < n_combo; combo++) {
                const bool nt = (flags & KP_NO_TIES) != 0;
                double r0, r1, t2;
                line_restricted(L0.cfg, T0, nt ? -1 : tie_force(L0.tie_par, combo), nt ? -1 : tie_force(L0.tie_tr, combo),
                                nt ? -1 : tie_force(L0.tie_ot, combo), &r0, &t2);
                line_restricted(L1.cfg, T1, nt ? -1 : tie_force(L1.tie_par, combo), nt ? -1 : tie_force(L1.tie_tr, combo),
                                nt ? -1 : tie_force(L1.tie_ot, combo), &r1, &t2);
                e += (cf * r1) * r0;
            }
        }
        const double sc = exp(s2_ff(q, s, ml, 0) + s2_ff(q, s, ml, 1) - factor);
        acc += s2_fw(q, s, ml, 0, g) * s2_fw(q, s, ml, 1, g) * sc * e;
    }
    out[(size_t)ml * 64 + g] = acc;
}


// HAPLOS accumulators of HOT LOOP 2 (cnF2freq.cpp:5554-5556 -> updatehaplo 1561-1575 -> trackpossible<HAPLOS>
// 1347-1350), per marker and window slot: out[len][7][2], out[..][k][phase] = sum of val over states,
// shift modes and admissible paths on which the individual in slot k is used with that phase
// (allele index ^ firstpar ^ localshift).  A slot that is homozygous with equal sure at the marker,
// or that the recursion does not reach, stays 0.  (movehaplos, cnF2freq.cpp:3601-3616, then turns the
// pair into haplobase/haplocount on the host, summed per individual.)
__global__ __launch_bounds__(64) void haplos_rows_kernel(Stage2Params q, uint32_t flags, double* out)
{
    const int g  = threadIdx.x;
    const int ml = blockIdx.x;
    const int m  = q.first + ml;
    const Window w = q.kp.windows[0];
    const double factor = q.loglik[0];
    const bool   nt = (flags & KP_NO_TIES) != 0;
    const int    n_combo = nt ? 1 : (1 << w.n_groups);
    double acc[7][2];
#pragma unroll
    for (int k = 0; k < 7; k++) acc[k][0] = acc[k][1] = 0.0;
    bool upd[7];                                   // doupdatehaplo of the slot at this marker
#pragma unroll
    for (int k = 0; k < 7; k++) {
        upd[k] = false;
        if (w.flags[k] & SLOT_PRESENT) {
            const Slot d = load_slot(q.kp, w.row[k], m);
            upd[k] = !(d.a0 == d.a1 && d.s0 == d.s1);          // cnF2freq.cpp:1235-1239
        }
    }
    const bool skip = isnan(factor) || factor < (double)CNF2_MINFACTOR_F;
    for (int s = 0; s < 8 && !skip; s++) {
        if ((s & w.shiftignore) || s >= w.shiftend) continue;
        if (factor - q.factors[s] > 40.0) continue;
        const double wg = s2_fw(q, s, ml, 0, g) * s2_fw(q, s, ml, 1, g) *
                          exp(s2_ff(q, s, ml, 0) + s2_ff(q, s, ml, 1) - factor);
        for (int f = 0; f < 2; f++) {
            LaneJob   L[2];
            LineTerms T[2];
            double    cf;
            bool      attop;
            s2_lines(q, w, m, g, s, f, &L[0], &L[1], &T[0], &T[1], &cf, &attop);
            // root: phase = f ^ shift bit 0 (firstpar is 0 at the root, cnF2freq.cpp:1156,1227)
            if (attop) {
                if (upd[0]) acc[0][f ^ (s & 1)] += wg * cf;
                continue;
            }
            for (int combo = 0; combo < n_combo; combo++) {
                int    force[2][3];                // [line][par, tr, ot]
                double r[2], t2;
                for (int P = 0; P < 2; P++) {
                    force[P][0] = nt ? -1 : tie_force(L[P].tie_par, combo);
                    force[P][1] = nt ? -1 : tie_force(L[P].tie_tr, combo);
                    force[P][2] = nt ? -1 : tie_force(L[P].tie_ot, combo);
                    line_restricted(L[P].cfg, T[P], force[P][0], force[P][1], force[P][2], &r[P], &t2);
                }
                const double e = (cf * r[1]) * r[0];
                if (upd[0]) acc[0][f ^ (s & 1)] += wg * e;
                for (int P = 0; P < 2; P++) {
                    const LineCfg& c = L[P].cfg;
                    if (!(c.par & SLOT_PRESENT)) continue;
                    const int  slot_par = 1 + 3 * P;
                    const int  ls = P ? (s >> 2) & 1 : (s >> 1) & 1;      // the parent's localshift
                    const bool interior = !(c.par & SLOT_FOUNDER);
                    for (int x = 0; x < 3; x++) {                          // 0 parent, 1 traced gp, 2 other gp
                        int slot;
                        if (x == 0) slot = slot_par;
                        else {
                            if (!interior) continue;
                            if (!((x == 1 ? c.tr : c.ot) & SLOT_PRESENT)) continue;
                            slot = slot_par + 1 + (x == 1 ? c.firstpar : (c.firstpar ^ 1));
                        }
                        if (!upd[slot]) continue;
                        for (int psi = 0; psi < 2; psi++) {
                            if (force[P][x] >= 0 && force[P][x] != psi) continue;   // tied slot: this combination fixes it
                            int fo[3] = {force[P][0], force[P][1], force[P][2]};
                            fo[x] = psi;
                            double rr;
                            line_restricted(c, T[P], fo[0], fo[1], fo[2], &rr, &t2);
                            const double ev = P ? (cf * rr) * r[0] : (cf * r[1]) * rr;
                            const int phase = (x == 0) ? (psi ^ ls) : psi;         // grandparents have localshift 0
                            acc[slot][phase] += wg * ev;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 7; k++)
#pragma unroll
        for (int ph = 0; ph < 2; ph++) {
            double v = acc[k][ph];
            v += lane_xor1(v);
            v += lane_xor2(v);
            v += dpp_mov_all<0x141>(v);
            v += lane_xor8(v);
            v += lane_xor16(v);
            v += lane_xor32(v);
            if (g == 0) out[((size_t)ml * 7 + k) * 2 + ph] = v;
        }
}

// ---------------------------------------------------------------------------------------------------
// Update-mode trackpossible for ONE path (flag99 >= 0), restated for the device: cnF2freq.cpp:1075-1359 with
// update in {GENOS = 2, HOMOZYGOUS = 4, GENOSPROBE = 8} (cnF2freq.cpp:792-795), zeropropagate = false,
// CORRECTIONINFERENCE = false.  A node is a window slot; the recursion depth is the template parameter
// (genwidth 4 -> 2 -> 1), so it unrolls into straight-line code.  GENOS adds updateval to
// inf[slot][allele index][markerval - 1] of every visited node of the traced line (cnF2freq.cpp:1351-1354).
// ---------------------------------------------------------------------------------------------------
#define TP_GENOS 2
#define TP_HOMOZYGOUS 4
#define TP_GENOSPROBE 8

__device__ __forceinline__ int tp_upflagit(int flag, int parnum, int genwidth)      // cnF2freq.cpp:321-329
{
    if (flag < 0) return flag;
    flag >>= parnum * (genwidth - 1);
    return flag & ((1 << (genwidth - 1)) - 1);
}

// NOEQ: zeropropagate = NO_EQUIVALENCE (cnF2freq.cpp:42): alleles are matched but an unknown incoming value is
// not bound (cnF2freq.cpp:311), every level weighs 0.5 instead of its phase weight (cnF2freq.cpp:1229-1233) and
// below the root only the traced line is followed (cnF2freq.cpp:1291); used by addvariance.
// CI: CORRECTIONINFERENCE is set (postmarkerdata, cnF2freq.cpp:8083-8085): a pair of equal alleles weighs 0 / 1 whatever
// its sure values are (cnF2freq.cpp:1235).
template <int GW, bool NOEQ = false, bool CI = false>
__device__ double tp_path(const KernelParams& p, const Window& w, int m, int slot, int inmv, double secondval,
                          unsigned flag, int flag99, int localshift, int update, double updateval, double* inf)
{
    const bool attopnow = !(update & TP_HOMOZYGOUS) && (GW == 1 || (w.flags[slot] & SLOT_FOUNDER));   // cpp:1120
    const Slot d        = load_slot(p, w.row[slot], m);
    const int  upflag2  = flag99 >> 1;                                   // cpp:1141-1146 (flag99 != -1)
    const int  upflag   = (int)(flag >> 1);
    const int  upshift  = localshift >> 1;
    const int  firstpar = flag & 1;                                      // cpp:1156
    const int  realf2n  = flag99 & 1;
    int        f2n      = realf2n;
    const int    mf = realf2n ? d.a1 : d.a0, mo = realf2n ? d.a0 : d.a1;
    const double sf = realf2n ? d.s1 : d.s0, so = realf2n ? d.s0 : d.s1;
    int    markerval;
    double baseval, mainsecondval = 0.0;
    const bool miss = markermiss(inmv, mf, &markerval);
    if (NOEQ && inmv == 0) markerval = 0;                                // cpp:311: no binding
    if (miss) {                                                          // cpp:1198-1202
        baseval = sf;
        if (sf != 0.0 && secondval != 0.0) mainsecondval = (1.0 - sf) * secondval;
    } else {                                                             // cpp:1203-1210
        const double esv = (inmv == 0 && markerval != 0) ? 1.0 : secondval;
        baseval          = 1.0 - sf;
        mainsecondval    = (mf == 0 ? 1.0 : sf) * esv;
    }
    if (attopnow) {                                                      // cpp:1213 (`update & 1` is false here)
        baseval += mainsecondval;
        mainsecondval = 0.0;
    } else if (mainsecondval != 0.0) mainsecondval /= baseval;
    f2n ^= (firstpar ^ localshift) & 1;                                  // cpp:1227
    if (NOEQ) baseval *= 0.5;                                            // cpp:1229-1233
    else if (d.a0 == d.a1 && (CI || d.s0 == d.s1)) baseval *= (f2n ? 1.0 : 0.0); // cpp:1235-1239
    else baseval *= fabs((f2n ? 1.0 : 0.0) - d.hw);                      // cpp:1245
    if constexpr (GW > 1) {
        if (baseval != 0.0 && !attopnow) {                               // cpp:1271
            const int down = update & ~TP_HOMOZYGOUS;                    // cpp:1280, 1322
            auto recurse = [&](int fp, int mv, double sv) -> double {    // recursetrackpossible, cpp:984-986, 1035-1057
                const int child = slot == 0 ? (fp ? 4 : 1) : slot + 1 + fp;
                if (!(w.flags[child] & SLOT_PRESENT)) return 1.0 + sv;   // cpp:1043-1046
                return tp_path<GW / 2, NOEQ, CI>(p, w, m, child, mv, sv, (unsigned)tp_upflagit(upflag, fp, GW),
                                       tp_upflagit(upflag2, fp, GW), tp_upflagit(upshift, fp, GW >> 1), down,
                                       updateval, inf);
            };
            if (!(update & TP_GENOS) && (!NOEQ || GW == 4)) {            // cpp:1291
                double secsecondval = 0.0;
                int    secmark      = mo;
                if (!(update & TP_HOMOZYGOUS)) {
                    if (so != 0.0) {                                     // cpp:1298-1302
                        baseval *= (1.0 - so);
                        secsecondval = so / (1.0 - so);
                    }
                } else if (markerval != secmark) {                       // cpp:1304-1313
                    if (secmark != 0) baseval *= so;
                    secmark = markerval;
                } else {                                                 // cpp:1314-1318
                    baseval *= (1.0 - so);
                }
                baseval *= recurse(firstpar ^ 1, secmark, secsecondval);
            }
            if (baseval != 0.0) baseval *= recurse(firstpar, markerval, mainsecondval);      // cpp:1336-1340
        }
    }
    if (baseval != 0.0 && (update & TP_GENOS) && (markerval == 1 || markerval == 2))        // cpp:1351-1354
        atomicAdd(inf + (slot * 2 + realf2n) * 2 + (markerval - 1), updateval);
    return baseval;
}

// infprobs / homozyg accumulators of HOT LOOP 2 at one marker (cnF2freq.cpp:5513-5577, DOINFPROBS), brute
// force like the reference: one thread per (shift mode, state, path).  out[0..27] = infprobs[slot][allele
// index][markerval - 1] before moveinfprobs (cnF2freq.cpp:3577-3597), out[28..29] = what is added to the
// individual's homozyg[marker].  Paths are masked as ignoreflag2 does where that changes a sum
// (flag2ignore and the all-or-none rule, cnF2freq.cpp:3478-3486; its third rule only drops zero terms).
__global__ __launch_bounds__(128) void infprobs_kernel(Stage2Params q, int marker, uint32_t flags, double* out)
{
    const int flag2 = threadIdx.x;            // 0..127
    const int g     = blockIdx.x & 63;
    const int s     = blockIdx.x >> 6;
    const Window w  = q.kp.windows[0];
    const double factor = q.loglik[0];
    if ((s & w.shiftignore) || s >= w.shiftend) return;
    if (isnan(factor) || factor < (double)CNF2_MINFACTOR_F) return;                   // cnF2freq.cpp:5403
    if (factor - q.factors[s] > 40.0) return;                                         // cnF2freq.cpp:5421
    if (flag2 & w.flag2ignore) return;                                                // cnF2freq.cpp:3478
    if (!(flags & KP_NO_TIES)) {
        for (int t = 0; t < w.n_groups; t++) {                                        // cnF2freq.cpp:3483-3486
            int mask = 0;
            for (int k = 0; k < 7; k++)
                if (w.tie[k] == t) mask |= 1 << k;
            const int filtered = (flag2 ^ (g * 2)) & mask;
            if (filtered && filtered != mask) return;
        }
    }
    // val of the path (as locked_query_kernel)
    const int ml = marker - q.first;
    const int f  = flag2 & 1;
    LaneJob   L0, L1;
    LineTerms T0, T1;
    double    cf;
    bool      attop;
    s2_lines(q, w, marker, g, s, f, &L0, &L1, &T0, &T1, &cf, &attop);
    double e;
    if (attop) e = cf;
    else {
        const int fp0 = (flag2 >> 1) & 1, fp1 = (flag2 >> 4) & 1;
        const int a0 = (flag2 >> (2 + L0.cfg.firstpar)) & 1, o0 = (flag2 >> (2 + (L0.cfg.firstpar ^ 1))) & 1;
        const int a1 = (flag2 >> (5 + L1.cfg.firstpar)) & 1, o1 = (flag2 >> (5 + (L1.cfg.firstpar ^ 1))) & 1;
        e = (cf * s2_path_term(L1.cfg, T1, fp1, a1, o1)) * s2_path_term(L0.cfg, T0, fp0, a0, o0);
    }
    const double am = s2_fw(q, s, ml, 0, g), be = s2_fw(q, s, ml, 1, g);
    const double lv = s2_ff(q, s, ml, 0) + log(am * e) + s2_ff(q, s, ml, 1) + log(be) - factor;
    if (!(isfinite(lv) && lv > -200.0)) return;                                       // cnF2freq.cpp:5502
    const double val = exp(lv);

    const KernelParams& p = q.kp;
    double sidevals[2][2], sums[2] = {0.0, 0.0}, homozyg[2];
    for (int side = 0; side < 2; side++)                                              // cnF2freq.cpp:5519-5528
        for (int i = 1; i <= 2; i++) {
            const double sv = tp_path<4>(p, w, marker, 0, i, 0.0, (unsigned)(g * 2 + side), flag2 ^ side, s,
                                         TP_GENOSPROBE, 0.0, nullptr);
            sidevals[side][i - 1] = sv;
            sums[side] += sv;
        }
    for (int i = 1; i <= 2; i++)                                                      // cnF2freq.cpp:5531-5537
        homozyg[i - 1] = tp_path<4>(p, w, marker, 0, i, 0.0, (unsigned)(g * 2), flag2, s, TP_HOMOZYGOUS, 0.0, nullptr);
    for (int side = 0; side < 2; side++)                                              // cnF2freq.cpp:5560-5568
        for (int i = 1; i <= 2; i++)
            tp_path<4>(p, w, marker, 0, i, 0.0, (unsigned)(g * 2 + side), flag2 ^ side, s, TP_GENOS,
                       val * sidevals[side][i - 1] / sums[side], out);
    for (int i = 1; i <= 2; i++) atomicAdd(out + 28 + (i - 1), val * homozyg[i - 1] / sums[0]);   // cnF2freq.cpp:5571-5575
}

// The same accumulators for every marker of the chromosome through the closed form of cnf2_accum.h (one
// line at a time instead of 128 paths per state and mode): out[len][30] = infprobs[7][2][2] then homozyg[2].
__global__ __launch_bounds__(64) void infprobs_rows_kernel(Stage2Params q, uint32_t flags, double* out)
{
    const int g  = threadIdx.x;
    const int ml = blockIdx.x;
    const int m  = q.first + ml;
    const Window w = q.kp.windows[0];
    const double factor = q.loglik[0];
    double acc[30];
#pragma unroll
    for (int k = 0; k < 30; k++) acc[k] = 0.0;
    Slot slot[7];
#pragma unroll
    for (int k = 0; k < 7; k++) slot[k] = load_slot(q.kp, w.row[k] < 0 ? 0 : w.row[k], m);
    const bool skip = isnan(factor) || factor < (double)CNF2_MINFACTOR_F;              // cnF2freq.cpp:5403
    for (int s = 0; s < 8 && !skip; s++) {
        if ((s & w.shiftignore) || s >= w.shiftend) continue;
        if (factor - q.factors[s] > 40.0) continue;                                    // cnF2freq.cpp:5421
        const double wg = s2_fw(q, s, ml, 0, g) * s2_fw(q, s, ml, 1, g) *
                          exp(s2_ff(q, s, ml, 0) + s2_ff(q, s, ml, 1) - factor);
        if (wg != 0.0) accum_infprobs(w, slot, g, s, wg, (flags & KP_NO_TIES) != 0, acc, acc + 28);
    }
    for (int k = 0; k < 30; k++) {
        double v = acc[k];
        v += lane_xor1(v);
        v += lane_xor2(v);
        v += dpp_mov_all<0x141>(v);
        v += lane_xor8(v);
        v += lane_xor16(v);
        v += lane_xor32(v);
        if (g == 0) out[(size_t)ml * 30 + k] = v;
    }
}

// individ::addvariance (cnF2freq.cpp:1489-1558) for the analysed individual of windows[0] and every marker in
// [first, first + len): trackpossible<false, NO_EQUIVALENCE> fed the individual's own two alleles (their sure
// as error odds) over shift modes 0-1, the 128 flags i = 2 g + firstpar and the admissible paths; per
// (shift, i & 1, flag2 & 1) the signed sum over the two alleles is squared.  out[ml] = variance, NaN when every
// term is zero (the reference leaves variances[marker] alone).  One block per marker, brute force.
__global__ __launch_bounds__(256) void addvariance_kernel(KernelParams p, int first, double* out)
{
    __shared__ double red_ok[8][256], red_full[256];
    const int    m  = first + blockIdx.x;
    const Window w  = p.windows[blockIdx.y];
    const Slot   me = load_slot(p, w.row[0], m);
    double ok[8], full = 0.0;
#pragma unroll
    for (int k = 0; k < 8; k++) ok[k] = 0.0;
    // 2 shifts x 128 flags x 128 paths = 32768 (shift, i, flag2) triples, 128 per thread
    for (int t = threadIdx.x; t < 2 * 128 * 128; t += 256) {
        const int flag2 = t & 127, i = (t >> 7) & 127, shift = t >> 14;
        if (flag2 & w.flag2ignore) continue;
        double d = 0.0;
        for (int allele = 0; allele < 2; allele++) {
            const double term = tp_path<4, true>(p, w, m, 0, allele ? me.a1 : me.a0, allele ? me.s1 : me.s0, (unsigned)i,
                                                 flag2, shift, 0, 0.0, nullptr);
            d += allele ? term : -term;
            full += term;
        }
        ok[(shift << 2) | ((i & 1) << 1) | (flag2 & 1)] += d;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) red_ok[k][threadIdx.x] = ok[k];
    red_full[threadIdx.x] = full;
    __syncthreads();
    if (threadIdx.x < 9) {
        const double* src = threadIdx.x < 8 ? red_ok[threadIdx.x] : red_full;
        double acc = 0.0;
        for (int k = 0; k < 256; k++) acc += src[k];
        if (threadIdx.x < 8) red_ok[threadIdx.x][0] = acc;
        else red_full[0] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sq = 0.0;
        for (int k = 0; k < 8; k++) sq += red_ok[k][0] * red_ok[k][0];
        out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red_full[0] != 0.0 ? sq : nan("");
    }
}

// fixparents' admissibility test (cnF2freq.cpp:1411-1431) for window y of p.windows at every marker: is there a state i
// and a path flag2 of parity b with a non-zero emission under shift mode 0, CORRECTIONINFERENCE set?  out[y][m][b].
// One thread per (marker, parity) tries the first paths of state 0, where nearly every admissible genotype is found at
// once; an item that found nothing there may need all 64 x 64 (state, path) pairs -- as a thread of its own it kept the
// other 63 lanes of its wavefront waiting for 4 096 evaluations (160 ms of a 5 000-individual set-up) -- so the wavefront
// takes such items one at a time, lane = state, and stops at the first path for which any state is non-zero.
__global__ __launch_bounds__(256) void okvals_kernel(KernelParams p, uint8_t* out)
{
    const int    t = blockIdx.x * blockDim.x + threadIdx.x;
    const int    lane = threadIdx.x & 63;
    const bool   live = t < p.n_markers * 2;
    const int    m = live ? t >> 1 : 0, b = t & 1;
    const Window w = p.windows[blockIdx.y];
    bool         ok = false;
    if (live)
        for (int flag2 = b; flag2 < 16 && !ok; flag2 += 2)
            ok = tp_path<4, false, true>(p, w, m, 0, 0, 0.0, 0u, flag2, 0, 0, 0.0, nullptr) != 0.0;
    unsigned long long pend = __ballot(live && !ok);
    while (pend) {
        const int L = __ffsll((long long)pend) - 1;
        pend &= pend - 1;
        const int mL = __shfl(m, L), bL = __shfl(b, L);
        bool      found = false;
        for (int flag2 = bL; flag2 < 128; flag2 += 2) {
            const double v = tp_path<4, false, true>(p, w, mL, 0, 0, 0.0, (unsigned)(lane * 2), flag2, 0, 0, 0.0, nullptr);
            if (__ballot(v != 0.0)) {
                found = true;
                break;
            }
        }
        if (lane == L) ok = found;
    }
    if (live) out[((size_t)blockIdx.y * p.n_markers + m) * 2 + b] = ok ? 1 : 0;
}
void launch_okvals(const KernelParams& p, int n_windows, uint8_t* out, hipStream_t stream)
{
    hipLaunchKernelGGL(okvals_kernel, dim3((p.n_markers * 2 + 255) / 256, n_windows), dim3(256), 0, stream, p, out);
}
void launch_addvariance_batch(const KernelParams& p, int n_windows, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(addvariance_kernel, dim3(p.n_markers, n_windows), dim3(256), 0, stream, p, 0, out);
}

// addvariance for window y of p.windows at every marker through its closed form (cnf2_variance.h): one thread per
// (window, marker) instead of 65 536 emission calls.  out[y][m], NaN where the reference leaves the entry alone.
__global__ __launch_bounds__(256) void variance_closed_kernel(KernelParams p, double* out)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= p.n_markers) return;
    const Window w = p.windows[blockIdx.y];
    Slot slot[7];
#pragma unroll
    for (int k = 0; k < 7; k++) slot[k] = load_slot(p, w.row[k] < 0 ? 0 : w.row[k], m);
    bool         valid;
    const double v = variance_closed(w, slot, &valid);
    out[(size_t)blockIdx.y * p.n_markers + m] = valid ? v : nan("");
}
void launch_variance_closed(const KernelParams& p, int n_windows, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(variance_closed_kernel, dim3((p.n_markers + 255) / 256, n_windows), dim3(256), 0, stream, p, out);
}

// addvariance for (window q of p.windows, markers[q]) with the reference's own rounding (variance_exact, cnf2_variance.h): four
// lanes per entry, one per class of the reference's loops -- 32 768 additions each, in the reference's order --, the first of
// them squares and adds the four.  out[q], NaN where the reference leaves the entry alone.  Called for the handful of markers
// per (record, chromosome) that can win lockhaplos' comparison, never for all.
__global__ __launch_bounds__(256) void variance_exact_kernel(KernelParams p, const int32_t* markers, int n, double* out)
{
    const int  t = blockIdx.x * blockDim.x + threadIdx.x;
    const int  k = t & 3;
    const bool live = (t >> 2) < n;
    const int  q = live ? t >> 2 : n - 1;
    const Window w = p.windows[q];
    const int    m = markers[q];
    Slot slot[7];
#pragma unroll
    for (int j = 0; j < 7; j++) slot[j] = load_slot(p, w.row[j] < 0 ? 0 : w.row[j], m);
    double ok, full;
    variance_exact_class(w, slot, k >> 1, k & 1, &ok, &full);
    double oks[4], fulls[4];
    const int base = (threadIdx.x & 63) & ~3;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        oks[j] = __shfl(ok, base + j);
        fulls[j] = __shfl(full, base + j);
    }
    if (live && k == 0) {
        bool         valid;
        const double v = variance_exact_finish(oks, fulls, &valid);
        out[q] = valid ? v : nan("");
    }
}
void launch_variance_exact(const KernelParams& p, const int32_t* markers, int n, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(variance_exact_kernel, dim3((n * 4 + 255) / 256), dim3(256), 0, stream, p, markers, n, out);
}

// =====================================================================================
// Batched HOT LOOP 2: every accumulator of cnF2freq.cpp:5416-5577 for one (job, marker)
// per wavefront, from the posterior weights a STOREW sweep left behind, through the table form of cnf2_acctab.h:
//   1. lane = table entry (P, f, sp, k): its AK_COUNT per-line sums (acc_entry) into LDS;
//   2. lane = (shift mode, low state bits) as in the sweep, registers = high state bits: the partial contractions
//      v (by f, s0), u (by f), z (by f, allele value) of wg with the restricted totals / HOMOZYGOUS probe sums;
//   3. lane = accumulator: 16-term dot products;
//   4. the per-locus reductions of doit (homozyg scale, moveinfprobs, movehaplos; cnF2freq.cpp:5876-5902,
//      3577-3616) with f64 atomics into the per-record arrays.
// Not tuned beyond the algebra (the table entries are evaluated by the plain host/device code of cnf2_accum.h).
// =====================================================================================
// What a window member receives at a locus (moveinfprobs / movehaplos, cnF2freq.cpp:3577-3616).  Default: f64 atomics on
// the per-record slabs (ranks and jobs add in whatever order they arrive).  CNF2_DETERMINISTIC (q.part != null): the six
// values go to the job's own row part[ind][m][slot][6] instead, and acc_gather_kernel adds the rows of a record in
// ascending order of the analysed individual -- one writer per element, the same sum to the bit on every run.
__device__ __forceinline__ void acc_emit(const AccParams& q, int ind, int slot, int rec, int m, const double inf[4], double norm,
                                         bool hap, double hbv, double hcv)
{
    const size_t M = (size_t)q.kp.n_markers;
    if (q.part) {
        double* d = q.part + (((size_t)ind * M + m) * 7 + slot) * 6;
#pragma unroll
        for (int t = 0; t < 4; t++) d[t] = inf[t] * norm;
        d[4] = hap ? hbv : 0.0;
        d[5] = hap ? hcv : 0.0;
        return;
    }
    double* dst = q.acc_inf + ((size_t)rec * M + m) * 4;
#pragma unroll
    for (int t = 0; t < 4; t++) atomicAdd(dst + t, inf[t] * norm);
    if (hap) {
        atomicAdd(q.acc_hb + (size_t)rec * M + m, hbv);
        atomicAdd(q.acc_hc + (size_t)rec * M + m, hcv);
    }
}

// CNF2_DETERMINISTIC: one thread per (record, marker) adds the rows its record received, in the order of list
// (ascending analysed individual, then slot): list[rec_start[r] .. rec_start[r + 1]) holds ind * 8 + slot.
__global__ __launch_bounds__(256) void acc_gather_kernel(AccParams q, const int32_t* rec_start, const int32_t* list, int n_rec)
{
    const size_t M = (size_t)q.kp.n_markers;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n_rec * M) return;
    const int    r = (int)(t / M), m = (int)(t % M);
    double       v[6] = {0, 0, 0, 0, 0, 0};
    for (int e = rec_start[r]; e < rec_start[r + 1]; e++) {
        const int     ind = list[e] >> 3, slot = list[e] & 7;
        const double* d = q.part + (((size_t)ind * M + m) * 7 + slot) * 6;
#pragma unroll
        for (int i = 0; i < 6; i++) v[i] += d[i];
    }
    double* dst = q.acc_inf + t * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) dst[i] += v[i];
    q.acc_hb[t] += v[4];
    q.acc_hc[t] += v[5];
}
void launch_acc_gather(const AccParams& q, const int32_t* rec_start, const int32_t* list, int n_rec, hipStream_t stream)
{
    const size_t n = (size_t)n_rec * q.kp.n_markers;
    hipLaunchKernelGGL(acc_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, q, rec_start, list, n_rec);
}

// The per-locus reductions of doit for one (job, marker) (cnF2freq.cpp:5876-5902, 3577-3616): out[44] = inf[7][2][2],
// homozyg[2], haplos[7][2] of the window's slots -> homozyg scale, moveinfprobs, movehaplos with f64 atomics.
__device__ __forceinline__ void acc_reduce_locus(const AccParams& q, const Job& jb, int m, int lane, double* out)
{
    const KernelParams& p = q.kp;
    const int32_t* srec = q.slot_rec + (size_t)jb.ind * 7;
    // this lane's own window member (lanes 0-6), read with the lane as index straight from memory
    const Window* wg = p.windows + jb.ind;
    const int     kk = lane < 7 ? lane : 0;
    const int     myrow = wg->row[kk];
    const Slot    mine = load_slot(p, myrow < 0 ? 0 : myrow, m);
    if (lane < 7) {
        // doupdatehaplo (cnF2freq.cpp:1224-1239): nothing for a slot that is homozygous with equal sure here
        const bool upd = (wg->flags[kk] & SLOT_PRESENT) && !(mine.a0 == mine.a1 && mine.s0 == mine.s1);
        if (!upd) out[30 + kk * 2] = out[30 + kk * 2 + 1] = 0.0;
    }
    wave_lds_fence();
    double self0 = 0.0;
    for (int k = 0; k < 7; k++)
        if (srec[k] == srec[0]) self0 += out[(k * 2 + 0) * 2 + 0] + out[(k * 2 + 0) * 2 + 1];
    const double sum = 1.0 / self0;                                                            // cnF2freq.cpp:5880-5885
    if (lane < 2) q.acc_hz[((size_t)jb.ind * p.n_markers + m) * 2 + lane] = out[28 + lane] * sum;
    if (lane < 7) {
        const int k = lane, r = srec[k];
        bool first = r >= 0;
        for (int k2 = 0; k2 < k; k2++) first = first && (srec[k2] != r);                      // reltree: unique members
        if (first) {
            double inf[4] = {0, 0, 0, 0}, h0 = 0.0, h1 = 0.0;
            int    occ = 0;
            for (int k2 = k; k2 < 7; k2++) {
                if (srec[k2] != r) continue;
                for (int t = 0; t < 4; t++) inf[t] += out[k2 * 4 + t];
                h0 += out[30 + k2 * 2];
                h1 += out[30 + k2 * 2 + 1];
                // reltreeordered: the individual itself always, ancestors only when non-empty (cnF2freq.cpp:3111-3152)
                if (k2 == 0 || !q.rec_empty[r]) occ++;
            }
            const double descf = (double)q.desc[srec[0]];
            double       norm = sum * 2;                                                       // cnF2freq.cpp:3582-3587
            for (int t = 0; t < occ; t++) norm /= 2;
            norm *= descf;
            const bool   hap = (h0 != 0.0 || h1 != 0.0) && fabs(mine.hw - 0.5) < 0.5 - 1e-12;      // cnF2freq.cpp:3601-3616
            const double md = (double)0.000005f;
            const double b1 = h0 + exp(-400.0) * md * md * 0.5;
            const double b2 = h1 + exp(-400.0) * md * md * 0.5;
            acc_emit(q, jb.ind, k, r, m, inf, norm, hap, b1 / (b1 + b2) * descf, descf);
        }
    }
}

#define ACC_TAB (64 * AK_COUNT)
#define ACC_LDS (ACC_TAB + 64 + 32 + 64 + 48)
#ifndef CNF2_ACC_MINBLOCKS
#define CNF2_ACC_MINBLOCKS 2   /* 2 blocks per CU (<= 256 VGPRs): measured 1.8 x faster than 1 block at 280 VGPRs */
#endif
__global__ __launch_bounds__(CNF2_BLOCK, CNF2_ACC_MINBLOCKS) void acc_rows_kernel(AccParams q)
{
    __shared__ double lds[CNF2_WAVES_PER_BLOCK][ACC_LDS];
    const int lane = threadIdx.x & 63;
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int job  = blockIdx.x;
    const int ml   = blockIdx.y * CNF2_WAVES_PER_BLOCK + wib;
    const KernelParams& p = q.kp;
    const Job jb = p.jobs[job];
    const int len = jb.last - jb.first + 1;
    if (ml >= len) return;
    const int    m  = jb.first + ml;
    const double factor = p.loglik[(size_t)jb.ind * p.n_chrom + jb.chrom];
    if (isnan(factor) || factor < (double)CNF2_MINFACTOR_F) return;                     // cnF2freq.cpp:5403
    const Window w = p.windows[jb.ind];
    if (!(q.flags & KP_ACC_TABLE) && !(w.flags[0] & SLOT_FOUNDER)) return;               // acc_paths_kernel's
    double* tab = lds[wib];                    // [64 entries][AK_COUNT]
    double* vt  = tab + ACC_TAB;               // [f][s0][16]
    double* ut  = vt + 64;                     // [f][16]
    double* zt  = ut + 32;                     // [f][i][16]
    double* out = zt + 64;                     // [44]: inf 28, hz 2, hap 14

    // posterior weights of this lane's 8 states
    double x[8];
    {
        const double* wp = p.wbuf + ((size_t)job * p.wstride + ml) * 512;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double2 v2 = *(const double2*)(wp + k * 128 + lane * 2);
            x[2 * k]     = v2.x;
            x[2 * k + 1] = v2.y;
        }
    }
    Slot slot[7];
#pragma unroll
    for (int k = 0; k < 7; k++) slot[k] = load_slot(p, w.row[k] < 0 ? 0 : w.row[k], m);
    const bool no_ties    = (q.flags & KP_NO_TIES) != 0;
    const bool root_attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    const int  n_combo    = (no_ties || root_attop) ? 1 : (1 << w.n_groups);
    const int  s = lane >> 3, s0 = s & 1, s1 = (s >> 1) & 1, s2 = (s >> 2) & 1, lo = state_lo(lane);
    // per root allele f (no run-time indexed arrays: everything the later steps need is selected from ar0 / ar1)
    AccRoot ar0, ar1;
    acc_root(slot[0], root_attop, 0, &ar0);
    acc_root(slot[0], root_attop, 1, &ar1);
    if (lane < 44) out[lane] = 0.0;

    for (int combo = 0; combo < n_combo; combo++) {
        // ---- 1. table entries
        wave_lds_fence();
        {
            double e[AK_COUNT];
            const bool f_e = ((lane >> 4) & 1) != 0;
            if (f_e ? ar1.live : ar0.live) {
                if (f_e) acc_entry(w, slot, lane, combo, no_ties, ar1, e);
                else acc_entry(w, slot, lane, combo, no_ties, ar0, e);
            } else {
#pragma unroll
                for (int k = 0; k < AK_COUNT; k++) e[k] = 0.0;
            }
#pragma unroll
            for (int k = 0; k < AK_COUNT; k++) tab[lane * AK_COUNT + k] = e[k];
        }
        wave_lds_fence();
        // ---- 2. partial contractions
#pragma unroll
        for (int f = 0; f < 2; f++) {
            const AccRoot& af = f ? ar1 : ar0;
            const double cf = af.live ? (s0 ? af.cf[1] : af.cf[0]) : 0.0;
            const double* t1 = tab + (size_t)((1 << 5) | (f << 4) | (s2 << 3)) * AK_COUNT;    // line 1 entries of this chain
            double tr = 0.0, th0 = 0.0, th1 = 0.0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const double cx = cf * x[j];
                if (cx != 0.0) {
                    tr += cx * t1[j * AK_COUNT + AK_R];
                    th0 += cx * t1[j * AK_COUNT + AK_HZ + 0];
                    th1 += cx * t1[j * AK_COUNT + AK_HZ + 1];
                }
            }
            // v: sum over s2 (lane bit 5); z: over s0 and s2 (lane bits 3, 5)
            tr += lane_xor32(tr);
            th0 += lane_xor32(th0);
            th1 += lane_xor32(th1);
            th0 += lane_xor8(th0);
            th1 += lane_xor8(th1);
            const int e0 = (s1 << 3) | lo;
            if (s2 == 0) vt[(f * 2 + s0) * 16 + e0] = tr;
            if (s2 == 0 && s0 == 0) {
                zt[(f * 2 + 0) * 16 + e0] = th0;
                zt[(f * 2 + 1) * 16 + e0] = th1;
            }
            // u: per high state j, sum over the chain's 8 lanes and over s0, s1 (lane bits 0-4)
            const double r0 = tab[(size_t)((0 << 5) | (f << 4) | (s1 << 3) | lo) * AK_COUNT + AK_R];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const double cx = cf * x[j];
                double       pj = (cx != 0.0) ? cx * r0 : 0.0;
                pj = chain_sum(pj);
                pj += lane_xor8(pj);
                pj += lane_xor16(pj);
                if ((lane & 31) == 0) ut[f * 16 + (s2 << 3) + j] = pj;
            }
        }
        wave_lds_fence();
        // ---- 3. lane = accumulator
        if (lane < 44) {
            double acc = 0.0;
            if (lane < 28) {                                   // inf[slot][allele index][i]
                const int slotk = lane >> 2, ax = (lane >> 1) & 1, i = lane & 1;
                if (slotk == 0) {
#pragma unroll
                    for (int f = 0; f < 2; f++) {
                        const AccRoot& af = f ? ar1 : ar0;
                        if (!af.live) continue;
                        if (root_attop) {
                            const bool side = (f != ax);
                            double     vsum = 0.0;
                            for (int e = 0; e < 16; e++) vsum += vt[(f * 2 + 0) * 16 + e] + vt[(f * 2 + 1) * 16 + e];
                            const double r0 = side ? af.Rs[1][0] : af.Rs[0][0], r1 = side ? af.Rs[1][1] : af.Rs[0][1];
                            acc += vsum * ((i ? r1 : r0) / (r0 + r1));
                        } else {
                            const int P = (f == ax) ? 0 : 1;                                   // fr = P ? f ^ 1 : f
                            for (int e = 0; e < 16; e++) {
                                const double wt = P ? ut[f * 16 + e] : vt[(f * 2 + 0) * 16 + e] + vt[(f * 2 + 1) * 16 + e];
                                acc += mul0(tab[(size_t)((P << 5) | (f << 4) | e) * AK_COUNT + AK_WROOT + i], wt);
                            }
                        }
                    }
                } else if (!root_attop) {
                    const int P = slotk >= 4, rel = slotk - (1 + 3 * P);                       // 0 parent, 1 / 2 grandparents
                    const int kind = rel == 0 ? AK_WPAR + ax * 2 + i : AK_WGP + ((rel - 1) * 2 + ax) * 2 + i;
#pragma unroll
                    for (int f = 0; f < 2; f++) {
                        if (!(f ? ar1.live : ar0.live)) continue;
                        for (int e = 0; e < 16; e++) {
                            const double wt = P ? ut[f * 16 + e] : vt[(f * 2 + 0) * 16 + e] + vt[(f * 2 + 1) * 16 + e];
                            acc += mul0(tab[(size_t)((P << 5) | (f << 4) | e) * AK_COUNT + kind], wt);
                        }
                    }
                }
            } else if (lane < 30) {                            // homozyg[i]
                const int i = lane - 28;
#pragma unroll
                for (int f = 0; f < 2; f++) {
                    const AccRoot& af = f ? ar1 : ar0;
                    if (!af.live) continue;
                    double t = 0.0;
                    for (int e = 0; e < 16; e++)
                        t += mul0(tab[(size_t)((0 << 5) | (f << 4) | e) * AK_COUNT + AK_HZ + i], zt[(f * 2 + i) * 16 + e]);
                    acc += (i ? af.hzscale[1] : af.hzscale[0]) * t;
                }
            } else {                                           // haplos[slot][phase]
                const int slotk = (lane - 30) >> 1, ph = (lane - 30) & 1;
                if (slotk == 0) {
#pragma unroll
                    for (int f = 0; f < 2; f++) {
                        if (!(f ? ar1.live : ar0.live)) continue;
                        const int s0v = f ^ ph;                                                // phase = f ^ s0
                        for (int e = 0; e < 16; e++)
                            acc += mul0(tab[(size_t)((0 << 5) | (f << 4) | e) * AK_COUNT + AK_R], vt[(f * 2 + s0v) * 16 + e]);
                    }
                } else if (!root_attop) {
                    const int P = slotk >= 4, rel = slotk - (1 + 3 * P);
#pragma unroll
                    for (int f = 0; f < 2; f++) {
                        if (!(f ? ar1.live : ar0.live)) continue;
                        for (int e = 0; e < 16; e++) {
                            const double wt = P ? ut[f * 16 + e] : vt[(f * 2 + 0) * 16 + e] + vt[(f * 2 + 1) * 16 + e];
                            acc += mul0(tab[(size_t)((P << 5) | (f << 4) | e) * AK_COUNT + AK_HAP + rel * 2 + ph], wt);
                        }
                    }
                }
            }
            out[lane] += acc;
        }
    }
    wave_lds_fence();
    acc_reduce_locus(q, jb, m, lane, out);
}

// ---------------------------------------------------------------------------------------------------
// The same accumulators in the path form of cnf2_accpath.h, for every window whose root is not the top of its own
// lines (the others stay with acc_rows_kernel).  One wavefront per (job, marker):
//   1. lane = path (P, f, fp, t, g1, g0): the line's three match terms, its GENOS weights and probe products -- once;
//   2. three butterfly stages (DPP on lane bits 0, 1, 3) turn paths into the 64 restricted totals and the
//      HOMOZYGOUS probe sums of line 1; into LDS in the emission-table numbering;
//   3. lane = (shift mode, low state bits): partial contractions v, u, z of the posterior weights (as acc_rows_kernel);
//   4. lane = entry: its weight from v / u / z; the transposed butterflies give every path the sum Omega of the
//      weights of the entries it belongs to, with the last stage kept split for each slot (HAPLOS phases);
//   5. lane = path: products, 8-lane sums, and a last gather by the 44 accumulators through LDS;
//   6. the per-locus reductions of doit (as acc_rows_kernel).
// ---------------------------------------------------------------------------------------------------
#define APL_RT   0                 /* [64] restricted totals, emission-table numbering */
#define APL_HT   64                /* [i][32] HOMOZYGOUS probe sums of the entries of line 1 */
#define APL_VT   128               /* [f][s0][s2][16] partial sums of v */
#define APL_ZT   256               /* [f][i][s0][s2][16] partial sums of z */
#define APL_UT   512               /* [f][16] */
#define APL_RED  544               /* [12 kinds][8 groups] */
#define APL_TW   640               /* [i][64] GENOS terms per path lane */
#define APL_OUT  768               /* [44] accumulators; [44..47] stay 0 (padding target of the gather lists) */
#define APL_OFF  816               /* [64][8] int: gather list of every accumulator lane */
#define APL_LDS  (816 + 256)
#define APL_ZERO (APL_OUT + 44)

template <int BIT>
__device__ __forceinline__ double path_xchg(double v)
{
    if (BIT == 0) return lane_xor1(v);
    if (BIT == 1) return lane_xor2(v);
    if (BIT == 2) return lane_xor4(v);
    return lane_xor8(v);
}
template <int BIT>
__device__ __forceinline__ double path_stage(double v, double own, double other)
{
    return own * v + other * path_xchg<BIT>(v);
}
// One halving step of a sum over lanes: of the pair (a, c) the lane keeps a (its bit BIT clear) or c (set) and adds
// the partner's copy of the same one; N values become N / 2, each summed over the lane pair.
template <int BIT>
__device__ __forceinline__ double halve_pair(double a, double c, bool bit)
{
    const double keep = bit ? c : a, send = bit ? a : c;
    return keep + path_xchg<BIT>(send);
}

#ifndef CNF2_APL_TILE
#define CNF2_APL_TILE 8            /* consecutive markers per wavefront (what depends on the window only is formed once) */
#endif
#ifndef CNF2_APL_MINBLOCKS
#define CNF2_APL_MINBLOCKS 2   /* 3 needs <= 168 VGPRs: 92 B of scratch and no faster */
#endif
__global__ __launch_bounds__(CNF2_BLOCK, CNF2_APL_MINBLOCKS) void acc_paths_kernel(AccParams q)
{
    __shared__ double lds[CNF2_WAVES_PER_BLOCK][APL_LDS];
    const int lane = threadIdx.x & 63;
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int job  = blockIdx.x;
    const int ml0  = (blockIdx.y * CNF2_WAVES_PER_BLOCK + wib) * CNF2_APL_TILE;
    const KernelParams& p = q.kp;
    const Job jb = p.jobs[job];
    const int len = jb.last - jb.first + 1;
    if (ml0 >= len) return;
    const double factor = p.loglik[(size_t)jb.ind * p.n_chrom + jb.chrom];
    if (isnan(factor) || factor < (double)CNF2_MINFACTOR_F) return;                     // cnF2freq.cpp:5403
    const Window w = p.windows[jb.ind];
    if (w.flags[0] & SLOT_FOUNDER) return;                                              // acc_rows_kernel's
    double* L = lds[wib];
    const bool no_ties = (q.flags & KP_NO_TIES) != 0;
    const int  n_combo = no_ties ? 1 : (1 << w.n_groups);
    // consumer role of this lane (step 3): shift mode bits and low state bits
    const int  s = lane >> 3, s0 = s & 1, s1 = (s >> 1) & 1, s2 = (s >> 2) & 1, lo = state_lo(lane);
    // path / entry role (steps 1, 2, 4, 5): this lane's line and what it reads of the window
    const int  P = lane >> 5, f = (lane >> 4) & 1, t = (lane >> 2) & 1;
    const int  eidx = path_entry_index(lane), e4 = eidx & 15;
    PathLine   ln;
    ln.fl_par  = P ? w.flags[4] : w.flags[1];
    ln.fl_a    = P ? w.flags[5] : w.flags[2];
    ln.fl_b    = P ? w.flags[6] : w.flags[3];
    ln.tie_par = P ? w.tie[4] : w.tie[1];
    ln.tie_a   = P ? w.tie[5] : w.tie[2];
    ln.tie_b   = P ? w.tie[6] : w.tie[3];
    const int row_root = w.row[0] < 0 ? 0 : w.row[0];
    int       row_par = P ? w.row[4] : w.row[1], row_a = P ? w.row[5] : w.row[2], row_b = P ? w.row[6] : w.row[3];
    row_par = row_par < 0 ? 0 : row_par;
    row_a   = row_a < 0 ? 0 : row_a;
    row_b   = row_b < 0 ? 0 : row_b;
    const uint32_t plan1 = path_coef_plan(ln, lane, 0, no_ties);
    const uint32_t outflags = ((ln.fl_par & SLOT_PRESENT) ? PO_PAR : 0) |
                              (((ln.fl_par & (SLOT_PRESENT | SLOT_FOUNDER)) == SLOT_PRESENT && (ln.fl_a & SLOT_PRESENT)) ? PO_G0 : 0) |
                              (((ln.fl_par & (SLOT_PRESENT | SLOT_FOUNDER)) == SLOT_PRESENT && (ln.fl_b & SLOT_PRESENT)) ? PO_G1 : 0) |
                              (((ln.fl_par & (SLOT_PRESENT | SLOT_FOUNDER)) == SLOT_PRESENT && ((t ? ln.fl_b : ln.fl_a) & SLOT_PRESENT)) ? PO_TR : 0);
    // gather list of this lane as an accumulator (step 5): which 8-lane sums / GENOS terms it adds up
    {
        int* off = (int*)(L + APL_OFF) + lane * 8;
        int  o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) o[k] = APL_ZERO;
        if (lane < 28) {
            const int slotk = lane >> 2, ax = (lane >> 1) & 1, i = lane & 1;
            const int PP = slotk >= 4, rel = slotk == 0 ? -1 : slotk - (1 + 3 * PP);
            if (slotk == 0) {                                             // root: allele index f ^ P; group = P<<2 | f<<1 | fp
                o[0] = APL_RED + i * 8 + (0 << 2 | ax << 1);
                o[1] = APL_RED + i * 8 + (0 << 2 | ax << 1 | 1);
                o[2] = APL_RED + i * 8 + (1 << 2 | (ax ^ 1) << 1);
                o[3] = APL_RED + i * 8 + (1 << 2 | (ax ^ 1) << 1 | 1);
            } else if (rel == 0) {                                        // parent: allele index fp
                if ((PP ? w.flags[4] : w.flags[1]) & SLOT_PRESENT) {
                    o[0] = APL_RED + i * 8 + (PP << 2 | 0 << 1 | ax);
                    o[1] = APL_RED + i * 8 + (PP << 2 | 1 << 1 | ax);
                }
            } else {                                                      // grandparent rel - 1 as the traced one: allele index g
                const int tt = rel - 1;
#pragma unroll
                for (int k = 0; k < 8; k++) {                             // over f, fp and the other grandparent's allele
                    const int ff = k & 1, pp = (k >> 1) & 1, go = k >> 2;
                    const int gg0 = tt ? go : ax, gg1 = tt ? ax : go;
                    o[k] = APL_TW + i * 64 + (PP << 5 | ff << 4 | pp << 3 | tt << 2 | gg1 << 1 | gg0);
                }
            }
        } else if (lane < 30) {
            const int i = lane - 28;
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = APL_RED + (2 + i) * 8 + k;
        } else if (lane < 44) {
            const int slotk = (lane - 30) >> 1, ph = (lane - 30) & 1;
            const int PP = slotk >= 4, rel = slotk == 0 ? -1 : slotk - (1 + 3 * PP);
            const int kind = 4 + (rel + 1) * 2 + ph;                      // root, parent, grandparent 0, grandparent 1
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = APL_RED + kind * 8 + PP * 4 + k;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) off[k] = o[k];
        if (lane < 4) L[APL_ZERO + lane] = 0.0;
    }
    // per-locus reductions (step 6): what lanes 0-6 need of the window's members
    const int32_t* srec = q.slot_rec + (size_t)jb.ind * 7;
    const int      kk = lane < 7 ? lane : 0;
    const int      myrec = srec[kk];
    int            mymask = 0, mask0 = 0;                                 // slots holding the same record as mine / as the root
    {
        const int r0 = srec[0];
        for (int k2 = 0; k2 < 7; k2++) {
            const int r2 = srec[k2];
            if (r2 == myrec) mymask |= 1 << k2;
            if (r2 == r0) mask0 |= 1 << k2;
        }
    }
    const bool   myfirst = lane < 7 && myrec >= 0 && (mymask & ((1 << kk) - 1)) == 0;              // reltree: unique members
    const Window* wmem = p.windows + jb.ind;                              // own member: read with the lane as index from memory
    const int    myrow_raw = wmem->row[kk];
    const int    myrow = myrow_raw < 0 ? 0 : myrow_raw;
    const bool   mypresent = (wmem->flags[kk] & SLOT_PRESENT) != 0;
    double       mynorm = 0.0;                                            // 2 / 2^occ * descendants (cnF2freq.cpp:3582-3587)
    const double descf = (double)q.desc[srec[0]];
    if (myfirst) {
        // reltreeordered: the individual itself always, ancestors only when non-empty (cnF2freq.cpp:3111-3152)
        const int occ = q.rec_empty[myrec] ? (mymask & 1) : __popc(mymask);
        mynorm = 2.0;
        for (int k2 = 0; k2 < occ; k2++) mynorm *= 0.5;
        mynorm *= descf;
    }

  for (int mi = 0; mi < CNF2_APL_TILE; mi++) {
    const int ml = ml0 + mi;
    if (ml >= len) break;
    const int m = jb.first + ml;
    double x[8];
    {
        const double* wp = p.wbuf + ((size_t)job * p.wstride + ml) * 512;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double2 v2 = *(const double2*)(wp + k * 128 + lane * 2);
            x[2 * k]     = v2.x;
            x[2 * k + 1] = v2.y;
        }
    }
    const Slot root = load_slot(p, row_root, m);
    ln.par = load_slot(p, row_par, m);
    ln.gpa = load_slot(p, row_a, m);
    ln.gpb = load_slot(p, row_b, m);
    const Slot mine = load_slot(p, myrow, m);
    double     cf0, cf1, hzs0, hzs1;            // c_f(s0 of the consumer role) for f = 0, 1; HOMOZYGOUS scale of this lane's f
    bool       live;
    PathTerms  T;
    {
        const double c0 = path_root_cbase(root, 0), c1 = path_root_cbase(root, 1);
        const double p0 = phase_weight(root, 0), p1 = phase_weight(root, 1);
        const bool   live0 = (c0 * p0 != 0.0) || (c0 * p1 != 0.0), live1 = (c1 * p0 != 0.0) || (c1 * p1 != 0.0);
        cf0  = live0 ? c0 * (s0 ? p1 : p0) : 0.0;              // c_0(s0) = cbase_0 * phase weight(0 ^ s0)
        cf1  = live1 ? c1 * (s0 ? p0 : p1) : 0.0;
        live = f ? live1 : live0;
        PathRoot pr;
        path_root(root, f, P, &pr);
        hzs0 = pr.hzscale0;
        hzs1 = pr.hzscale1;
        path_terms(ln, lane, pr, &T);
    }
    if (!live) T.term0 = T.k0 = T.k1 = 0.0;
    wave_lds_fence();
    if (lane < 44) L[APL_OUT + lane] = 0.0;

    for (int combo = 0; combo < n_combo; combo++) {
        PathCoef C;
        uint32_t plan = plan1;
        if (__builtin_amdgcn_readfirstlane(n_combo) != 1) plan = path_coef_plan(ln, lane, combo, no_ties);
        path_coef_apply(plan, ln, &C);
        // ---- 2. paths -> entries
        wave_lds_fence();
        {
            double R = T.term0, H0 = T.k0, H1 = T.k1;
            R  = path_stage<0>(R, C.g0_s, C.g0_f);
            H0 = path_stage<0>(H0, C.g0_s, C.g0_f);
            H1 = path_stage<0>(H1, C.g0_s, C.g0_f);
            R  = path_stage<1>(R, C.g1_s, C.g1_f);
            H0 = path_stage<1>(H0, C.g1_s, C.g1_f);
            H1 = path_stage<1>(H1, C.g1_s, C.g1_f);
            R  = path_stage<3>(R, C.par_s, C.par_f);
            H0 = path_stage<3>(H0, C.par_s, C.par_f);
            H1 = path_stage<3>(H1, C.par_s, C.par_f);
            L[APL_RT + eidx] = R;
            if (P) {
                L[APL_HT + (eidx & 31)]      = H0;
                L[APL_HT + 32 + (eidx & 31)] = H1;
            }
        }
        wave_lds_fence();
        const double Rmine = L[APL_RT + eidx];
        // ---- 3. partial contractions (lane = shift mode and low state bits, registers = high state bits)
        {
            const int e0 = (s1 << 3) | lo;
            double    uu[16];                                             // [f][j]: this lane's terms of u
#pragma unroll
            for (int ff = 0; ff < 2; ff++) {
                const double  cf = ff ? cf1 : cf0;
                const double* t1 = L + APL_RT + ((1 << 5) | (ff << 4) | (s2 << 3));      // line 1 entries of this chain
                const double* h0 = L + APL_HT + ((ff << 4) | (s2 << 3));
                const double* h1 = h0 + 32;
                const double  r0 = cf * L[APL_RT + ((0 << 5) | (ff << 4) | (s1 << 3) | lo)];
                double tr = 0.0, th0 = 0.0, th1 = 0.0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    tr  = fma(x[j], t1[j], tr);
                    th0 = fma(x[j], h0[j], th0);
                    th1 = fma(x[j], h1[j], th1);
                    uu[ff * 8 + j] = x[j] * r0;
                }
                // v: still to be summed over s2; z: over s0 and s2 -- by the entry lanes that read them
                L[APL_VT + ((ff * 2 + s0) * 2 + s2) * 16 + e0]            = cf * tr;
                L[APL_ZT + (((ff * 2 + 0) * 2 + s0) * 2 + s2) * 16 + e0] = cf * th0;
                L[APL_ZT + (((ff * 2 + 1) * 2 + s0) * 2 + s2) * 16 + e0] = cf * th1;
            }
            // u[f][s2][j]: sum over the 32 lanes of this half (s0, s1, low bits), 16 values -> one per lane
            double h8[8], h4[4], h2[2];
#pragma unroll
            for (int k = 0; k < 8; k++) h8[k] = halve_pair<0>(uu[2 * k], uu[2 * k + 1], (lane & 1) != 0);
#pragma unroll
            for (int k = 0; k < 4; k++) h4[k] = halve_pair<1>(h8[2 * k], h8[2 * k + 1], (lane & 2) != 0);
#pragma unroll
            for (int k = 0; k < 2; k++) h2[k] = halve_pair<2>(h4[2 * k], h4[2 * k + 1], (lane & 4) != 0);
            double h1v = halve_pair<3>(h2[0], h2[1], (lane & 8) != 0);
            h1v += lane_xor16(h1v);
            // the lane holds index (f<<3 | j) = its low four lane bits
            if ((lane & 16) == 0) L[APL_UT + ((lane >> 3) & 1) * 16 + (s2 << 3) + (lane & 7)] = h1v;
        }
        wave_lds_fence();
        // ---- 4. lane = entry: weights, then entries -> paths
        double v0, v1, wt, z0 = 0.0, z1 = 0.0;
        {
            const double* vt = L + APL_VT + (f * 2) * 2 * 16 + e4;
            v0 = vt[0] + vt[16];
            v1 = vt[32] + vt[48];
            if (P == 0) {
                const double* zt = L + APL_ZT + (f * 2) * 4 * 16 + e4;
                z0 = (zt[0] + zt[16]) + (zt[32] + zt[48]);
                z1 = (zt[64] + zt[80]) + (zt[96] + zt[112]);
            }
            wt = P ? L[APL_UT + f * 16 + e4] : v0 + v1;
        }
        // HAPLOS of the root straight from the entries: phase f ^ s0 (cnF2freq.cpp:1227)
        double hr0 = 0.0, hr1 = 0.0;
        if (P == 0 && live) {
            const double a = Rmine * v0, b = Rmine * v1;               // totals and weights are finite: plain products
            hr0 = f ? b : a;
            hr1 = f ? a : b;
        }
        const double X1  = path_stage<1>(wt, C.g1_s, C.g1_r);
        const double X01 = path_stage<0>(X1, C.g0_s, C.g0_r);
        const double Y   = path_stage<3>(X1, C.par_s, C.par_r);
        const double X0  = path_stage<0>(wt, C.g0_s, C.g0_r);
        const double Z   = path_stage<3>(X0, C.par_s, C.par_r);
        z0 = path_stage<1>(z0, C.g1_s, C.g1_r);
        z1 = path_stage<1>(z1, C.g1_s, C.g1_r);
        z0 = path_stage<0>(z0, C.g0_s, C.g0_r);
        z1 = path_stage<0>(z1, C.g0_s, C.g0_r);
        z0 = path_stage<3>(z0, C.par_s, C.par_r);
        z1 = path_stage<3>(z1, C.par_s, C.par_r);
        const double o_par_self = C.par_s * X01, o_par_part = C.par_r * path_xchg<3>(X01);      // phase t / !t
        const double o_g0_self = C.g0_s * Y, o_g0_part = C.g0_r * path_xchg<0>(Y);              // phase 0 / 1
        const double o_g1_self = C.g1_s * Z, o_g1_part = C.g1_r * path_xchg<1>(Z);
        // ---- 5. lane = path
        double val[12];
        {
            const double c = (o_par_self + o_par_part) * T.term0;
            const bool   ok = c != 0.0 && T.den_ok;
            val[0] = ok ? c * T.w0 : 0.0;                               // GENOS terms
            val[1] = ok ? c * T.w1 : 0.0;
            const double c0 = z0 * T.term0, c1 = z1 * T.term0;          // HOMOZYGOUS (line 0)
            val[2] = (P == 0 && c0 != 0.0) ? hzs0 * (c0 * T.w0) : 0.0;
            val[3] = (P == 0 && c1 != 0.0) ? hzs1 * (c1 * T.w1) : 0.0;
            val[4] = hr0;
            val[5] = hr1;
            const double tp = (outflags & PO_PAR) ? T.term0 : 0.0, ta = (outflags & PO_G0) ? T.term0 : 0.0, tb = (outflags & PO_G1) ? T.term0 : 0.0;
            const double ps = tp * o_par_self, pp = tp * o_par_part;
            val[6] = t ? pp : ps;
            val[7] = t ? ps : pp;
            val[8]  = ta * o_g0_self;
            val[9]  = ta * o_g0_part;
            val[10] = tb * o_g1_self;
            val[11] = tb * o_g1_part;
        }
        L[APL_TW + lane]      = (outflags & PO_TR) ? val[0] : 0.0;
        L[APL_TW + 64 + lane] = (outflags & PO_TR) ? val[1] : 0.0;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            // sum over lane bits 0-2 (g0, g1, t): xor1, xor2, then the pairing i <-> 7 - i finishes it
            double v = val[k];
            v += lane_xor1(v);
            v += lane_xor2(v);
            v += dpp_mov_all<0x141>(v);
            if ((lane & 7) == 0) L[APL_RED + k * 8 + (lane >> 3)] = v;
        }
        wave_lds_fence();
        if (lane < 44) {
            const int* off = (const int*)(L + APL_OFF) + lane * 8;
            double     acc = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++) acc += L[off[k]];
            L[APL_OUT + lane] += acc;
        }
    }
    // ---- 6. per-locus reductions (cnF2freq.cpp:5876-5902, 3577-3616)
    double* out = L + APL_OUT;
    if (lane < 7) {
        // doupdatehaplo (cnF2freq.cpp:1224-1239): nothing for a slot that is homozygous with equal sure here
        const bool upd = mypresent && !(mine.a0 == mine.a1 && mine.s0 == mine.s1);
        if (!upd) out[30 + kk * 2] = out[30 + kk * 2 + 1] = 0.0;
    }
    wave_lds_fence();
    double self0 = 0.0;
    for (int mm = mask0; mm; mm &= mm - 1) {
        const int k = __ffs(mm) - 1;
        self0 += out[k * 4 + 0] + out[k * 4 + 1];
    }
    const double sum = 1.0 / self0;                                                                // cnF2freq.cpp:5880-5885
    if (lane < 2) q.acc_hz[((size_t)jb.ind * p.n_markers + m) * 2 + lane] = out[28 + lane] * sum;
    if (myfirst) {
        double inf[4] = {0, 0, 0, 0}, h0 = 0.0, h1 = 0.0;
        for (int mm = mymask; mm; mm &= mm - 1) {
            const int k2 = __ffs(mm) - 1;
#pragma unroll
            for (int i = 0; i < 4; i++) inf[i] += out[k2 * 4 + i];
            h0 += out[30 + k2 * 2];
            h1 += out[30 + k2 * 2 + 1];
        }
        const double norm = sum * mynorm;
        const bool   hap = (h0 != 0.0 || h1 != 0.0) && fabs(mine.hw - 0.5) < 0.5 - 1e-12;         // cnF2freq.cpp:3601-3616
        const double md = (double)0.000005f;
        const double b1 = h0 + exp(-400.0) * md * md * 0.5;
        const double b2 = h1 + exp(-400.0) * md * md * 0.5;
        acc_emit(q, jb.ind, kk, myrec, m, inf, norm, hap, b1 / (b1 + b2) * descf, descf);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The path form once more, laid out like the sweep's tile producer (cnf2_accpath.h, "tile form"): a wavefront takes
// 8 consecutive markers of a job at a time and
//   A. lane = part x marker (part = P, f, traced grandparent): the match terms of its 8 paths, butterflies in
//      registers, its 8 restricted totals (and HOMOZYGOUS probe sums) into the marker's table in LDS;
//   B. marker by marker, lane = (shift mode, low state bits): the contractions v, u, z of the 512 posterior weights;
//   C. lane = part x marker again: the weights of its 8 entries back to paths, the sums over its paths;
//   D. lane = accumulator x marker: gathers over the parts; lane = window member x marker: the per-locus
//      reductions of doit with f64 atomics.
// A and C run once per 8 markers instead of once per marker: the per-path work of acc_paths_kernel divided by 8.
// Two wavefronts per block (25 KB of LDS each: three blocks per CU).  Windows with tie groups (a loop over phase
// combinations) stay with acc_paths_kernel.
// ---------------------------------------------------------------------------------------------------
#define APT_WAVES 2
#ifndef CNF2_APT_TILES
#define CNF2_APT_TILES 4           /* tiles (of 8 markers) per wavefront: amortises the per-job set-up */
#endif
#define APT_SSTRIDE 110            /* staging block of one marker: 4 (P, f) x 17 (16 sums over t, odd stride), then 8 parts x 5 (the
                                      traced grandparent's 4 infprobs sums) */
/* strides are padded so that the 8 markers of a tile (lane bits 0-2) fall into different LDS banks */
#define APT_TSTRIDE 138            /* table of one marker: R of line 0 [0, 32), of line 1 [34, 66), H[0] [68, 100), H[1] [102, 134) */
#define APT_R1    34
#define APT_H0    68
#define APT_H1    102
#define APT_VSTRIDE 162            /* V[f][s0][16] at 0, U[f][16] at 64, Z[f][i][16] at 96 */
#define APT_OSTRIDE 50
#define APT_TAB   0                /* [mi][APT_TSTRIDE]; later the staging blocks [mi][APT_SSTRIDE] */
#define APT_CF    (8 * APT_TSTRIDE) /* [mi][f][s0] */
#define APT_VUZ   (APT_CF + 32)    /* [mi][APT_VSTRIDE] */
#define APT_OUT   APT_VUZ           /* [mi][APT_OSTRIDE]: over the weights, once they have been read */
#define APT_LIST  (APT_VUZ + 8 * APT_VSTRIDE) /* int [44][2]: the two staged sums every accumulator adds (offsets inside a staging block) */
#define APT_LDS   (APT_LIST + 44)

__global__ __launch_bounds__(64 * APT_WAVES, 2) void acc_tile_kernel(AccParams q)
{
    __shared__ double lds[APT_WAVES][APT_LDS];
    const int lane = threadIdx.x & 63;
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int job  = blockIdx.x;
    const KernelParams& p = q.kp;
    const Job jb = p.jobs[job];
    const int len = jb.last - jb.first + 1;
    const int ml_wave = (blockIdx.y * APT_WAVES + wib) * (8 * CNF2_APT_TILES);
    if (ml_wave >= len) return;
    const double factor = p.loglik[(size_t)jb.ind * p.n_chrom + jb.chrom];
    if (isnan(factor) || factor < (double)CNF2_MINFACTOR_F) return;                     // cnF2freq.cpp:5403
    const Window w = p.windows[jb.ind];
    if (w.flags[0] & SLOT_FOUNDER) return;                                              // acc_rows_kernel's
    double* L = lds[wib];
    const bool no_ties = (q.flags & KP_NO_TIES) != 0;
    if (!no_ties && w.n_groups > 0) return;                                              // tie combinations: acc_paths_kernel's
    // role B: shift mode bits and low state bits
    const int  s = lane >> 3, s0 = s & 1, s1 = (s >> 1) & 1, s2 = (s >> 2) & 1, lo = state_lo(lane);
    // roles A, C: part and marker of the tile
    const int  part = lane >> 3, mi = lane & 7;
    const int  P = part >> 2, f = (part >> 1) & 1, t = part & 1;
    PathLine   ln;
    ln.fl_par  = P ? w.flags[4] : w.flags[1];
    ln.fl_a    = P ? w.flags[5] : w.flags[2];
    ln.fl_b    = P ? w.flags[6] : w.flags[3];
    ln.tie_par = P ? w.tie[4] : w.tie[1];
    ln.tie_a   = P ? w.tie[5] : w.tie[2];
    ln.tie_b   = P ? w.tie[6] : w.tie[3];
    const int row_root = w.row[0] < 0 ? 0 : w.row[0];
    int       row_par = P ? w.row[4] : w.row[1], row_a = P ? w.row[5] : w.row[2], row_b = P ? w.row[6] : w.row[3];
    row_par = row_par < 0 ? 0 : row_par;
    row_a   = row_a < 0 ? 0 : row_a;
    row_b   = row_b < 0 ? 0 : row_b;
    const uint32_t plan1 = path_mats_plan(ln, t, 0, no_ties);
    const bool     general = (ln.fl_par & (SLOT_PRESENT | SLOT_FOUNDER)) == SLOT_PRESENT;
    const bool     do_par = (ln.fl_par & SLOT_PRESENT) != 0;
    const bool     do_tr = general && ((t ? ln.fl_b : ln.fl_a) & SLOT_PRESENT), do_ot = general && ((t ? ln.fl_a : ln.fl_b) & SLOT_PRESENT);
    // where entry r = sp<<2 | b_ot<<1 | b_tr of this lane sits in the table: sp<<3 | u1<<2 | u0<<1 | t, the roles mapped
    // to grandparents 0 / 1 by t (two per-job offsets, as in the sweep's tile producer)
    const int      e_tr = t ? 4 : 2, e_ot = t ? 2 : 4;
    // gather lists (role D): the two staged sums every accumulator adds up; offsets inside a marker's staging block
    // ((P, f) block pf = P*2 + f at pf*17: 0-1 root infprobs [i], 2-5 parent [a][i], 6-7 homozyg [i], 8-9 HAPLOS of the root
    // [phase], 10-11 of the parent, 12-13 / 14-15 of grandparent 0 / 1; part block at 68 + part*5: traced grandparent [a][i])
    if (lane < 44) {
        int o2[2];
        if (lane < 28) {
            const int slotk = lane >> 2, ax = (lane >> 1) & 1, i = lane & 1;
            const int PP = slotk >= 4, rel = slotk == 0 ? -1 : slotk - (1 + 3 * PP);
            if (slotk == 0) {                                             // root: allele index f ^ P
                o2[0] = (0 * 2 + ax) * 17 + i;
                o2[1] = (1 * 2 + (ax ^ 1)) * 17 + i;
            } else if (rel == 0) {                                        // parent: allele index fp
                o2[0] = (PP * 2 + 0) * 17 + 2 + ax * 2 + i;
                o2[1] = (PP * 2 + 1) * 17 + 2 + ax * 2 + i;
            } else {                                                      // grandparent rel - 1 where it is the traced one
                o2[0] = 68 + (PP << 2 | 0 << 1 | (rel - 1)) * 5 + ax * 2 + i;
                o2[1] = 68 + (PP << 2 | 1 << 1 | (rel - 1)) * 5 + ax * 2 + i;
            }
        } else if (lane < 30) {
            o2[0] = 0 * 17 + 6 + (lane - 28);
            o2[1] = 1 * 17 + 6 + (lane - 28);
        } else {
            const int slotk = (lane - 30) >> 1, ph = (lane - 30) & 1;
            const int PP = slotk >= 4, rel = slotk == 0 ? -1 : slotk - (1 + 3 * PP);
            o2[0] = (PP * 2 + 0) * 17 + 8 + (rel + 1) * 2 + ph;
            o2[1] = (PP * 2 + 1) * 17 + 8 + (rel + 1) * 2 + ph;
        }
        int* list = (int*)(L + APT_LIST) + lane * 2;
        list[0] = o2[0];
        list[1] = o2[1];
    }
    // role D, per-locus reductions: lane = member k x marker
    const int      dk = lane >> 3;                                        // window slot 0-6 (7: idle)
    const int32_t* srec = q.slot_rec + (size_t)jb.ind * 7;
    const int      kk = dk < 7 ? dk : 0;
    const int      myrec = srec[kk];
    int            mymask = 0, mask0 = 0;                                 // slots holding the same record as mine / as the root
    {
        const int r0 = srec[0];
        for (int k2 = 0; k2 < 7; k2++) {
            const int r2 = srec[k2];
            if (r2 == myrec) mymask |= 1 << k2;
            if (r2 == r0) mask0 |= 1 << k2;
        }
    }
    const bool    myfirst = dk < 7 && myrec >= 0 && (mymask & ((1 << kk) - 1)) == 0;             // reltree: unique members
    const Window* wmem = p.windows + jb.ind;
    const int     myrow_raw = wmem->row[kk];
    const int     myrow = myrow_raw < 0 ? 0 : myrow_raw;
    const bool    mypresent = (wmem->flags[kk] & SLOT_PRESENT) != 0;
    const double  descf = (double)q.desc[srec[0]];
    double        mynorm = 0.0;                                           // 2 / 2^occ * descendants (cnF2freq.cpp:3582-3587)
    if (myfirst) {
        // reltreeordered: the individual itself always, ancestors only when non-empty (cnF2freq.cpp:3111-3152)
        const int occ = q.rec_empty[myrec] ? (mymask & 1) : __popc(mymask);
        mynorm = 2.0;
        for (int k2 = 0; k2 < occ; k2++) mynorm *= 0.5;
        mynorm *= descf;
    }

    for (int tile = 0; tile < CNF2_APT_TILES; tile++) {
        const int ml0 = ml_wave + tile * 8;
        if (ml0 >= len) break;
        const int nvalid = (len - ml0 < 8) ? len - ml0 : 8;
        const int mlc = ml0 + (mi < nvalid ? mi : nvalid - 1);            // lanes beyond the chromosome redo its last marker
        const int m = jb.first + mlc;
        // ---- A. paths of this lane's part at its marker
        const Slot root = load_slot(p, row_root, m);
        ln.par = load_slot(p, row_par, m);
        ln.gpa = load_slot(p, row_a, m);
        ln.gpb = load_slot(p, row_b, m);
        const Slot mine = load_slot(p, myrow, m);
        PathTile T;
        double   hzs0, hzs1;
        {
            const double cb = path_root_cbase(root, f);
            const double c0 = cb * phase_weight(root, f ^ 0), c1 = cb * phase_weight(root, f ^ 1);      // c_f(s0 = 0), c_f(1)
            const bool   live = (c0 != 0.0) || (c1 != 0.0);
            PathRoot pr;
            path_root(root, f, P, &pr);
            hzs0 = pr.hzscale0;
            hzs1 = pr.hzscale1;
            path_terms8(ln, t, pr, &T);
            if (!live) {
#pragma unroll
                for (int r = 0; r < 8; r++) T.term0[r] = T.k0[r] = T.k1[r] = 0.0;
            }
            wave_lds_fence();                                             // the previous tile's readers are done
            if ((part & 5) == 0) {                                        // P == 0, t == 0: one writer per f
                L[APT_CF + mi * 4 + f * 2 + 0] = c0;
                L[APT_CF + mi * 4 + f * 2 + 1] = c1;
            }
        }
        {
            PathMats M;
            path_mats_apply(plan1, ln, t, &M);
            {
                double R[8], H0[8], H1[8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    R[r]  = T.term0[r];
                    H0[r] = T.k0[r];
                    H1[r] = T.k1[r];
                }
                tile_fwd<0>(R, M.tr);
                tile_fwd<1>(R, M.ot);
                tile_fwd<2>(R, M.par);
                double* tab = L + APT_TAB + mi * APT_TSTRIDE + ((f << 4) | t);
#pragma unroll
                for (int r = 0; r < 8; r++)
                    tab[(P ? APT_R1 : 0) + ((r >> 2) << 3) + ((r & 2) ? e_ot : 0) + ((r & 1) ? e_tr : 0)] = R[r];
                if (P) {
                    tile_fwd<0>(H0, M.tr);
                    tile_fwd<0>(H1, M.tr);
                    tile_fwd<1>(H0, M.ot);
                    tile_fwd<1>(H1, M.ot);
                    tile_fwd<2>(H0, M.par);
                    tile_fwd<2>(H1, M.par);
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        const int e3 = ((r >> 2) << 3) + ((r & 2) ? e_ot : 0) + ((r & 1) ? e_tr : 0);
                        tab[APT_H0 + e3] = H0[r];
                        tab[APT_H1 + e3] = H1[r];
                    }
                }
            }
            wave_lds_fence();
            // ---- B. contractions, one marker at a time (every lane in the consumer role)
            double xn[8];                                                 // posterior weights of the next marker, in flight
            {
                const double* wp = p.wbuf + ((size_t)job * p.wstride + ml0) * 512;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double2 v2 = *(const double2*)(wp + k * 128 + lane * 2);
                    xn[2 * k]     = v2.x;
                    xn[2 * k + 1] = v2.y;
                }
            }
#ifdef CNF2_X_FUSEDACC   /* timing ablation only: the contractions come from the sweep (160 doubles per marker) */
            for (int mj = 0; mj < nvalid; mj++) {
                const double* wp = p.wbuf + ((size_t)job * p.wstride + (ml0 + mj)) * 160;
                double*       vuz = L + APT_VUZ + mj * APT_VSTRIDE;
                vuz[lane] = wp[lane];
                vuz[64 + lane] = wp[64 + lane];
                if (lane < 32) vuz[128 + lane] = wp[128 + lane];
            }
            (void)xn;
            for (int mj = 0; mj < 0; mj++) {
#else
            for (int mj = 0; mj < nvalid; mj++) {
#endif
                double x[8];
#pragma unroll
                for (int k = 0; k < 8; k++) x[k] = xn[k];
                if (mj + 1 < nvalid) {
                    const double* wp = p.wbuf + ((size_t)job * p.wstride + (ml0 + mj + 1)) * 512;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const double2 v2 = *(const double2*)(wp + k * 128 + lane * 2);
                        xn[2 * k]     = v2.x;
                        xn[2 * k + 1] = v2.y;
                    }
                }
                const double* tab = L + APT_TAB + mj * APT_TSTRIDE;
                double*       vuz = L + APT_VUZ + mj * APT_VSTRIDE;
                const int     e0 = (s1 << 3) | lo;
                double        uu[16];
#pragma unroll
                for (int ff = 0; ff < 2; ff++) {
                    const double  cf = L[APT_CF + mj * 4 + ff * 2 + s0];
                    const double* t1 = tab + APT_R1 + ((ff << 4) | (s2 << 3));            // line 1 entries of this chain
                    const double* h0 = tab + APT_H0 + ((ff << 4) | (s2 << 3));
                    const double* h1 = tab + APT_H1 + ((ff << 4) | (s2 << 3));
                    const double  r0 = cf * tab[(ff << 4) | (s1 << 3) | lo];
                    double tr = 0.0, th0 = 0.0, th1 = 0.0;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        tr  = fma(x[j], t1[j], tr);
                        th0 = fma(x[j], h0[j], th0);
                        th1 = fma(x[j], h1[j], th1);
                        uu[ff * 8 + j] = x[j] * r0;
                    }
                    tr *= cf;
                    th0 *= cf;
                    th1 *= cf;
                    // v: sum over s2 (lane bit 5); z: over s0 and s2 (lane bits 3, 5)
                    tr += lane_xor32(tr);
                    th0 += lane_xor32(th0);
                    th1 += lane_xor32(th1);
                    th0 += lane_xor8(th0);
                    th1 += lane_xor8(th1);
                    if (s2 == 0) vuz[(ff * 2 + s0) * 16 + e0] = tr;
                    if (s2 == 0 && s0 == 0) {
                        vuz[96 + (ff * 2 + 0) * 16 + e0] = th0;
                        vuz[96 + (ff * 2 + 1) * 16 + e0] = th1;
                    }
                }
                // u[f][s2][j]: sum over the 32 lanes of this half, 16 values -> one per lane (index = low four lane bits).
                // (Measured alternatives: the weights read a second time in the transposed role, from memory or through
                // LDS, so that u sums in registers: 20 % fewer VALU instructions, 3-9 % slower -- the kernel waits on
                // dependent LDS / memory round trips at 1.5 waves per SIMD, not on the VALU.)
                double h8[8], h4[4], h2[2];
#pragma unroll
                for (int k = 0; k < 8; k++) h8[k] = halve_pair<0>(uu[2 * k], uu[2 * k + 1], (lane & 1) != 0);
#pragma unroll
                for (int k = 0; k < 4; k++) h4[k] = halve_pair<1>(h8[2 * k], h8[2 * k + 1], (lane & 2) != 0);
#pragma unroll
                for (int k = 0; k < 2; k++) h2[k] = halve_pair<2>(h4[2 * k], h4[2 * k + 1], (lane & 4) != 0);
                double h1v = halve_pair<3>(h2[0], h2[1], (lane & 8) != 0);
                h1v += lane_xor16(h1v);
                if ((lane & 16) == 0) vuz[64 + ((lane >> 3) & 1) * 16 + (s2 << 3) + (lane & 7)] = h1v;
            }
            wave_lds_fence();
            // ---- C. the weights of this lane's entries, back to its paths
            TileSums S;
            double   hroot0 = 0.0, hroot1 = 0.0;
            {
                const double* vuz = L + APT_VUZ + mi * APT_VSTRIDE;
                const double* tab = L + APT_TAB + mi * APT_TSTRIDE;
                double wt[8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int e4 = ((r >> 2) << 3) + ((r & 2) ? e_ot : 0) + ((r & 1) ? e_tr : 0) + t;
                    if (P) {
                        wt[r] = vuz[64 + f * 16 + e4];
                    } else {
                        const double v0 = vuz[(f * 2 + 0) * 16 + e4], v1 = vuz[(f * 2 + 1) * 16 + e4];
                        wt[r] = v0 + v1;
                        // HAPLOS of the root straight from the entries: phase f ^ s0 (cnF2freq.cpp:1227)
                        const double Re = tab[(f << 4) | e4];
                        const double a = Re * v0, b = Re * v1;
                        hroot0 += f ? b : a;
                        hroot1 += f ? a : b;
                    }
                }
                tile_accumulate(T, M, t, wt, do_par, do_tr, do_ot, do_tr, &S);
                S.hz[0] = S.hz[1] = 0.0;
                if (P == 0) {
                    double z[8];
#pragma unroll
                    for (int r = 0; r < 8; r++)
                        z[r] = vuz[96 + (f * 2 + 0) * 16 + ((r >> 2) << 3) + ((r & 2) ? e_ot : 0) + ((r & 1) ? e_tr : 0) + t];
                    S.hz[0] = tile_homozyg(T, M, z, T.w0, hzs0);
#pragma unroll
                    for (int r = 0; r < 8; r++)
                        z[r] = vuz[96 + (f * 2 + 1) * 16 + ((r >> 2) << 3) + ((r & 2) ? e_ot : 0) + ((r & 1) ? e_tr : 0) + t];
                    S.hz[1] = tile_homozyg(T, M, z, T.w1, hzs1);
                }
            }
            wave_lds_fence();                                             // tables and weights are read: stage over the tables
            {
                // sums over the two traced grandparents of a (P, f) first (lane bit 3), then one writer per pair
                double sv[16] = {S.inf_root[0], S.inf_root[1], S.inf_par[0][0], S.inf_par[0][1], S.inf_par[1][0], S.inf_par[1][1],
                                 S.hz[0], S.hz[1], hroot0, hroot1, S.hap_par[0], S.hap_par[1],
                                 t ? S.hap_ot[0] : S.hap_tr[0], t ? S.hap_ot[1] : S.hap_tr[1],        // grandparent 0: traced iff t == 0
                                 t ? S.hap_tr[0] : S.hap_ot[0], t ? S.hap_tr[1] : S.hap_ot[1]};
#pragma unroll
                for (int k = 0; k < 16; k++) sv[k] += lane_xor8(sv[k]);
                double* blk = L + APT_TAB + mi * APT_SSTRIDE;
                if (t == 0) {
                    double* st = blk + (part >> 1) * 17;
#pragma unroll
                    for (int k = 0; k < 16; k++) st[k] = sv[k];
                }
                double* sp = blk + 68 + part * 5;
                sp[0] = S.inf_tr[0][0];
                sp[1] = S.inf_tr[0][1];
                sp[2] = S.inf_tr[1][0];
                sp[3] = S.inf_tr[1][1];
            }
            wave_lds_fence();
            // ---- D1. lane = accumulator x marker
#pragma unroll
            for (int round = 0; round < 6; round++) {
                const int o = (lane >> 3) + 8 * round;
                if (o < 44) {
                    const int*    list = (const int*)(L + APT_LIST) + o * 2;
                    const double* st = L + APT_TAB + mi * APT_SSTRIDE;
                    L[APT_OUT + mi * APT_OSTRIDE + o] = st[list[0]] + st[list[1]];
                }
            }
        }
        wave_lds_fence();
        // ---- D2. per-locus reductions (cnF2freq.cpp:5876-5902, 3577-3616): lane = window member x marker
        double* out = L + APT_OUT + mi * APT_OSTRIDE;
        const bool valid = mi < nvalid;
        if (dk < 7) {
            // doupdatehaplo (cnF2freq.cpp:1224-1239): nothing for a slot that is homozygous with equal sure here
            const bool upd = mypresent && !(mine.a0 == mine.a1 && mine.s0 == mine.s1);
            if (!upd) out[30 + kk * 2] = out[30 + kk * 2 + 1] = 0.0;
        }
        wave_lds_fence();
        double self0 = 0.0;
        for (int mm = mask0; mm; mm &= mm - 1) {
            const int k = __ffs(mm) - 1;
            self0 += out[k * 4 + 0] + out[k * 4 + 1];
        }
        const double sum = 1.0 / self0;                                                            // cnF2freq.cpp:5880-5885
        if (valid && dk < 2) q.acc_hz[((size_t)jb.ind * p.n_markers + m) * 2 + dk] = out[28 + dk] * sum;
        if (valid && myfirst) {
            double inf[4] = {0, 0, 0, 0}, h0 = 0.0, h1 = 0.0;
            for (int mm = mymask; mm; mm &= mm - 1) {
                const int k2 = __ffs(mm) - 1;
#pragma unroll
                for (int i = 0; i < 4; i++) inf[i] += out[k2 * 4 + i];
                h0 += out[30 + k2 * 2];
                h1 += out[30 + k2 * 2 + 1];
            }
            const double norm = sum * mynorm;
            const bool   hap = (h0 != 0.0 || h1 != 0.0) && fabs(mine.hw - 0.5) < 0.5 - 1e-12;     // cnF2freq.cpp:3601-3616
            const double md = (double)0.000005f;
            const double b1 = h0 + exp(-400.0) * md * md * 0.5;
            const double b2 = h1 + exp(-400.0) * md * md * 0.5;
            acc_emit(q, jb.ind, kk, myrec, m, inf, norm, hap, b1 / (b1 + b2) * descf, descf);
        }
    }
}

void launch_acc_rows(const AccParams& q, hipStream_t stream)
{
    dim3 grid(q.n_jobs, (q.max_len + CNF2_WAVES_PER_BLOCK - 1) / CNF2_WAVES_PER_BLOCK);
    const int per_block = CNF2_WAVES_PER_BLOCK * CNF2_APL_TILE;
    dim3 gridp(q.n_jobs, (q.max_len + per_block - 1) / per_block);
    const int per_block_t = APT_WAVES * 8 * CNF2_APT_TILES;
    dim3 gridt(q.n_jobs, (q.max_len + per_block_t - 1) / per_block_t);
    if (!(q.flags & KP_ACC_TABLE)) {
        if (q.flags & KP_ACC_LANES) hipLaunchKernelGGL(acc_paths_kernel, gridp, dim3(CNF2_BLOCK), 0, stream, q);
        else hipLaunchKernelGGL(acc_tile_kernel, gridt, dim3(64 * APT_WAVES), 0, stream, q);
    }
    if (q.flags & (KP_ACC_TABLE | KP_ACC_ATTOP)) hipLaunchKernelGGL(acc_rows_kernel, grid, dim3(CNF2_BLOCK), 0, stream, q);
}

void launch_addvariance(const KernelParams& p, int first, int len, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(addvariance_kernel, dim3(len), dim3(256), 0, stream, p, first, out);
}

void launch_infprobs_rows(const Stage2Params& q, uint32_t flags, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(infprobs_rows_kernel, dim3(q.len), dim3(64), 0, stream, q, flags, out);
}

void launch_infprobs(const Stage2Params& q, int marker, uint32_t flags, double* out, hipStream_t stream)
{
    (void)hipMemsetAsync(out, 0, 30 * sizeof(double), stream);
    hipLaunchKernelGGL(infprobs_kernel, dim3(8 * 64), dim3(128), 0, stream, q, marker, flags, out);
}

void launch_locked_query(const Stage2Params& q, int marker, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(locked_query_kernel, dim3(512), dim3(128), 0, stream, q, marker, out);
}
void launch_turn_scan(const Stage2Params& q, int marker, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(turn_scan_kernel, dim3(1024), dim3(64), 0, stream, q, marker, out);
}
// every marker of the chromosome: out[len][128][8]
__global__ __launch_bounds__(64) void turn_scan_rows_kernel(Stage2Params q, double* out)
{
    const int k    = threadIdx.x;
    const int s    = blockIdx.x & 7;
    const int turn = blockIdx.x >> 3;
    const int ml   = blockIdx.y;
    const int xorturn = turn & 54;                                                      // cnF2freq.cpp:508
    const int shiftx  = (turn >> 6) | ((turn & 1) ? 2 : 0) | ((turn & 8) ? 4 : 0);     // cnF2freq.cpp:509-510
    const int s2      = s ^ shiftx;
    double v = s2_fw(q, s, ml, 2, k ^ xorturn) * s2_fw(q, s2, ml, 1, k);
    v += lane_xor1(v);
    v += lane_xor2(v);
    v += dpp_mov_all<0x141>(v);
    v += lane_xor8(v);
    v += lane_xor16(v);
    v += lane_xor32(v);
    if (k == 0) {
        double r = (v > 0.0) ? s2_ff(q, s, ml, 2) + s2_ff(q, s2, ml, 1) + log(v) : (double)CNF2_MINFACTOR_F;
        out[((size_t)ml * 128 + turn) * 8 + s] = r - q.loglik[0];
    }
}
void launch_turn_scan_rows(const Stage2Params& q, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(turn_scan_rows_kernel, dim3(1024, q.len), dim3(64), 0, stream, q, out);
}
void launch_haplos_rows(const Stage2Params& q, uint32_t flags, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(haplos_rows_kernel, dim3(q.len), dim3(64), 0, stream, q, flags, out);
}
void launch_state_rows(const Stage2Params& q, uint32_t flags, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(state_rows_kernel, dim3(q.len), dim3(64), 0, stream, q, flags, out);
}

static void zero_job_counter(const KernelParams& p, hipStream_t stream)
{
    if (p.job_next) (void)hipMemsetAsync(p.job_next, 0, sizeof(int), stream);
}

void launch_fb_packed(const KernelParams& p, int grid, hipStream_t stream)
{
    zero_job_counter(p, stream);
    hipLaunchKernelGGL(fb_packed_kernel, dim3(grid), dim3(CNF2_BLOCK), 0, stream, p);
}

// dynamic LDS of the tied instantiations: 8 table rows per wave, each with the restricted tables of two tie combinations
#define CNF2_TIED_LDS_BYTES (CNF2_WAVES_PER_BLOCK * 8 * (TAB_STRIDE + 128) * (int)sizeof(double))
template <class K>
static void allow_tied_lds(K kernel)
{
    static bool done = false;        // (per instantiation; more than 64 KB of dynamic LDS has to be asked for once)
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CNF2_TIED_LDS_BYTES);
        done = true;
    }
}
void launch_fb_fast_tied(const KernelParams& p, int grid, hipStream_t stream)
{
    zero_job_counter(p, stream);
    allow_tied_lds(fb_fast_kernel<true, 0, false, true>);
    hipLaunchKernelGGL((fb_fast_kernel<true, 0, false, true>), dim3(grid), dim3(CNF2_BLOCK), CNF2_TIED_LDS_BYTES, stream, p);
    launch_likelihood_logs(p, stream);
}
void launch_fb_fast_tied_w(const KernelParams& p, int grid, hipStream_t stream)
{
    zero_job_counter(p, stream);
    allow_tied_lds(fb_fast_kernel<true, 1, false, true>);
    hipLaunchKernelGGL((fb_fast_kernel<true, 1, false, true>), dim3(grid), dim3(CNF2_BLOCK), CNF2_TIED_LDS_BYTES, stream, p);
    launch_likelihood_logs(p, stream);
}
void launch_fb_fast_xpose(const KernelParams& p, int grid, hipStream_t stream)
{
    zero_job_counter(p, stream);
    hipLaunchKernelGGL((fb_fast_kernel<true, 0, true>), dim3(grid), dim3(CNF2_BLOCK), 0, stream, p);
    launch_likelihood_logs(p, stream);
}
void launch_fb_fast(const KernelParams& p, int grid, bool half_spill, hipStream_t stream)
{
    zero_job_counter(p, stream);
    if (half_spill) hipLaunchKernelGGL(fb_fast_kernel<true>, dim3(grid), dim3(CNF2_BLOCK), 0, stream, p);
    else hipLaunchKernelGGL(fb_fast_kernel<false>, dim3(grid), dim3(CNF2_BLOCK), 0, stream, p);
    launch_likelihood_logs(p, stream);
}

void launch_row_flags(const uint8_t* allele8, const double2* sure, int n_rows, int n_markers, uint8_t* flags,
                      hipStream_t stream)
{
    hipLaunchKernelGGL(row_flags_kernel, dim3(n_rows), dim3(256), 0, stream, allele8, sure, n_markers, flags);
}

void launch_fb_fast_w(const KernelParams& p, int grid, hipStream_t stream, bool rows)
{
    zero_job_counter(p, stream);
    if (rows) hipLaunchKernelGGL((fb_fast_kernel<true, 1>), dim3(grid), dim3(CNF2_BLOCK), 0, stream, p);
    else hipLaunchKernelGGL((fb_fast_kernel<true, 3>), dim3(grid), dim3(CNF2_BLOCK), 0, stream, p);
    launch_likelihood_logs(p, stream);
}
void launch_fb_w(const KernelParams& p, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL((fb_kernel<false, 1>), dim3(grid), dim3(CNF2_BLOCK), 0, stream, p);
}
void launch_fb_fast_ab(const KernelParams& p, int grid, hipStream_t stream)
{
    zero_job_counter(p, stream);
    hipLaunchKernelGGL((fb_fast_kernel<true, 2>), dim3(grid), dim3(CNF2_BLOCK), 0, stream, p);
    launch_likelihood_logs(p, stream);
}
void launch_fb_ab(const KernelParams& p, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL((fb_kernel<false, 2>), dim3(grid), dim3(CNF2_BLOCK), 0, stream, p);
}

// =====================================================================================
// Batched turn scan (HOT LOOP 3, cnF2freq.cpp:5686-5752): rawervals[turn][s] =
// doanalyze<aroundturner>(turn, classicstop(q, -1)) - factor for every marker of every job of a turn-scan sweep.
// aroundturner(turn) XORs the states with turn & 54 and the shift mode with
// (turn >> 6) | (turn & 1 ? 2 : 0) | (turn & 8 ? 4 : 0) (cnF2freq.cpp:506-511), so
//     rawervals[turn][s] = log sum_k A_s(k ^ (turn & 54)) B_{s ^ shiftx}(k) + scales - factor:
// per (job, marker) 64 pairs of shift modes (s, s ^ sx) x 16 flips of state bits 1, 2, 4, 5 = 1 024 dot products of
// 64 terms -- 131 072 flops on 8.4 KB read: the kernel is bound by the f64 FMA rate, not by HBM.
// One wavefront per (job, marker), lane = sx << 3 | s.  The sweep's rows go through LDS once ([chain][k][l][2], rows
// padded to 66 doubles so that the 8 chains a read touches sit in different banks); a lane keeps its A_s whole in
// registers (64 doubles), streams B_{s ^ sx} from LDS (one 16-byte read per 32 FMAs) and runs 16 accumulators, one
// per flip: the flips are register renaming (A[l ^ L(x)][j ^ J(x)] with compile-time indices), there is no
// cross-lane sum and no multiply is done twice.  Scales are mantissa and binary exponent (CNF2_TURN_ROW), so the
// log-sum-exp over the admissible shift modes per turn (what the clause weights of cnF2freq.cpp:5800-5817 are made of)
// is exact scaling (frexp / ldexp), a halving exchange over the 8 lanes of a group that leaves every lane 2 of the
// group's 16 sums, and 2 logarithms per lane.  The full table costs 16 logarithms per lane (parity / debugging).
// Each wave walks CNF2_TURN_SPAN consecutive markers; the second wave of the SIMD computes while this one waits for its rows.
// =====================================================================================
// log of a positive normal double: frexp to [sqrt(1/2), sqrt(2)), log m = 2 atanh((m-1)/(m+1)) by its series to t^21
// (|t| <= 0.1716: remainder 6e-19), reciprocal-based quotient.  About a third of the library routine's instructions;
// absolute error of log m below 2e-16 (the turn scan is compared at 1e-9).
__device__ __forceinline__ double log_pos(double w)
{
    int    e;
    double m = frexp(w, &e);                     // [0.5, 1)
    const bool lowhalf = m < 0.70710678118654752440;
    m = lowhalf ? m + m : m;
    e = lowhalf ? e - 1 : e;
    const double n = m - 1.0, d = m + 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    double t = n * r;
    t = fma(fma(-d, t, n), r, t);
    const double t2 = t * t;
    double p = 1.0 / 21.0;
    p = fma(p, t2, 1.0 / 19.0);
    p = fma(p, t2, 1.0 / 17.0);
    p = fma(p, t2, 1.0 / 15.0);
    p = fma(p, t2, 1.0 / 13.0);
    p = fma(p, t2, 1.0 / 11.0);
    p = fma(p, t2, 1.0 / 9.0);
    p = fma(p, t2, 1.0 / 7.0);
    p = fma(p, t2, 1.0 / 5.0);
    p = fma(p, t2, 1.0 / 3.0);
    p = p * t2;                                  // log m = 2 t (1 + p)
    const double lm = fma(t + t, p, t + t);
    const double ed = (double)e;
    return fma(ed, 0.69314718036912381649, fma(ed, 1.9082149292705877e-10, lm));   // ln 2 split: the high part has 21 trailing zero bits
}

// Sums over the 8 admissible shift modes of a lane group (lanes differing in bits 0-2) of acc[x] K 2^E for the 16 flips
// x, as mantissa sums w2[i] relative to 2^e2[i]: lane s returns the flips x = (s0 s1 s2 i), i = 0, 1 (halving exchange:
// after the step with partner lane ^ 1 a lane keeps the 8 flips whose bit 3 is its s bit 0, then bit 2 <- s bit 1,
// bit 1 <- s bit 2).  PER_FLIP: the reference exponent is the largest E + exponent(acc[x] K) of the flip (exact for
// any spread); otherwise the largest E of the group.
template <bool PER_FLIP>
__device__ __forceinline__ void turn_group_sums(const double (&acc)[16], double K, int E, bool s_ok, int s, double (&w2)[2], int (&e2)[2])
{
    const bool on0 = s_ok && K > 0.0;
    double mt[16];
    int    emax[16];
    if (PER_FLIP) {
#pragma unroll
        for (int x = 0; x < 16; x++) {
            const double kk = acc[x] * K;
            const bool   on = on0 && kk > 0.0;
            int          fe;
            const double fm = frexp(kk, &fe);
            const int    et = on ? E + fe : -(1 << 29);
            int          mx = et;
            mx = max(mx, __builtin_amdgcn_mov_dpp(mx, 0xB1, 0xF, 0xF, true));
            mx = max(mx, __builtin_amdgcn_mov_dpp(mx, 0x4E, 0xF, 0xF, true));
            mx = max(mx, __builtin_amdgcn_ds_swizzle(mx, (4 << 10) | 0x1F));
            emax[x] = mx;
            const int d = et - mx;
            mt[x] = (on && d > -1100) ? ldexp(fm, d) : 0.0;
        }
    } else {
        const int Ee = on0 ? E : -(1 << 29);
        int       mx = Ee;
        mx = max(mx, __builtin_amdgcn_mov_dpp(mx, 0xB1, 0xF, 0xF, true));
        mx = max(mx, __builtin_amdgcn_mov_dpp(mx, 0x4E, 0xF, 0xF, true));
        mx = max(mx, __builtin_amdgcn_ds_swizzle(mx, (4 << 10) | 0x1F));
        const double Ke = on0 ? K : 0.0;
        int          d  = Ee - mx;
        d = d < -2000 ? -2000 : d;
#pragma unroll
        for (int x = 0; x < 16; x++) {
            mt[x]   = ldexp(acc[x] * Ke, d);
            emax[x] = mx;
        }
    }
    const bool b0 = s & 1, b1 = s & 2, b2 = s & 4;
    double u[8], t4[4];
    int    eu[8], e4[4];
    // (both candidates are read before the choice: a conditional read of an array element would put the array in scratch)
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const double lo = mt[i], hi = mt[8 + i];
        const int    el = emax[i], eh = emax[8 + i];
        u[i]  = (b0 ? hi : lo) + lane_xor1(b0 ? lo : hi);
        eu[i] = b0 ? eh : el;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double lo = u[i], hi = u[4 + i];
        const int    el = eu[i], eh = eu[4 + i];
        t4[i] = (b1 ? hi : lo) + lane_xor2(b1 ? lo : hi);
        e4[i] = b1 ? eh : el;
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const double lo = t4[i], hi = t4[2 + i];
        const int    el = e4[i], eh = e4[2 + i];
        w2[i] = (b2 ? hi : lo) + lane_xor4(b2 ? lo : hi);
        e2[i] = b2 ? eh : el;
    }
}

#ifndef CNF2_TURN_SPAN
#define CNF2_TURN_SPAN 8
#endif
constexpr int TURN_RS = 66;
// MFMA form: the 1 024 dot products of a unit are one 32 x 32 x 64 product C[(s, xa)][(s2, xb)] = sum_k A_s(k ^ F(xa))
// B_s2(k ^ F(xb)) -- the flip x = xa | xb << 2 splits into the two that move the lane-held state bits (on A) and the two
// that move the register-held ones (on B) -- run as 2 x 2 tiles x 16 k-steps of v_mfma_f64_16x16x4_f64.  The operands come
// from LDS images in which every state coordinate is an address bit of a chain's 512-byte row (the four a flip can move --
// fp, fq of the lane position, j1, j2 of the register index -- and the two none moves), so a flip is an XOR on the address;
// the coordinate the lane's own flip bit moves sits at 16 bytes, rows are 544 bytes apart: the 32 lanes a ds_read_b64 serves
// together then cover the 64 banks, and every read is the lane's base address (one of two, by that coordinate) plus an
// immediate.  The sums leave through a [s][s2][x] image in
// the same LDS and are read back by the lanes in the layout the rest of the kernel works in.
typedef double d4v __attribute__((ext_vector_type(4)));
constexpr int TURN_MRS = 68;          // doubles per chain row of the MFMA images (544 bytes)
constexpr int TURN_CS  = 18;          // doubles per (s, s2) cell of the image the sums leave through
constexpr int TURN_LDS_VALU = 16 * TURN_RS, TURN_LDS_MFMA = 64 * TURN_CS;
template <bool MFMA>
__global__ __launch_bounds__(CNF2_BLOCK, MFMA ? 4 : 2) void turn_rows_kernel(TurnParams q)
{
    __shared__ __attribute__((aligned(16))) double lds[CNF2_WAVES_PER_BLOCK][MFMA ? TURN_LDS_MFMA : TURN_LDS_VALU];
    static_assert(16 * TURN_MRS <= TURN_LDS_MFMA, "the operand images and the image of the sums share the LDS");
    const int lane = threadIdx.x & 63;
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int job  = blockIdx.x;
    const KernelParams& p = q.kp;
    const Job jb  = p.jobs[job];
    const int len = jb.last - jb.first + 1;
    const int ml0 = (blockIdx.y * CNF2_WAVES_PER_BLOCK + wib) * CNF2_TURN_SPAN;
    if (ml0 >= len) return;
    const int ml1 = (ml0 + CNF2_TURN_SPAN < len) ? ml0 + CNF2_TURN_SPAN : len;
    // the fast kernel's butterflies drop a constant per gap: its reported log-likelihood carries the chromosome's sum of
    // their logarithms, the stored alpha / beta do not
    const double factor = p.loglik[(size_t)jb.ind * p.n_chrom + jb.chrom] - (q.scaled_transitions ? p.chrom_logk[jb.chrom] : 0.0);
    const Window& w = p.windows[jb.ind];
    const int  s = lane & 7, sx = lane >> 3, s2 = s ^ sx;
    const bool s_ok = !(s & w.shiftignore) && s < w.shiftend;
    const int  n_ok = __builtin_popcountll(__ballot(s_ok) & 0xFFull);
    double*    L  = lds[wib];
    // where this lane's loads go: it holds chain lane >> 3, position lane & 7, registers 2k, 2k + 1 of the sweep's layout
    // (a chain's row is [k][l][2]: the 8 lanes a ds_write_b128 serves together cover the 32 banks)
    double*    Lw = L + (lane >> 3) * TURN_RS + (lane & 7) * 2;
    const double* Ap = L + s * TURN_RS;
    const double* Bp = L + (8 + s2) * TURN_RS;
    // turn bits 0, 3, 6 come from sx bits 1, 2, 0
    const int tshift = ((sx >> 1) & 1) | (((sx >> 2) & 1) << 3) | ((sx & 1) << 6);
    // MFMA form, byte offsets in the wave's LDS.  Loading role (chain lane >> 3, position l = lane & 7): the coordinates of
    // l that the position flips move -- fp (l ^ 2 flips it), fq (l ^ 7 flips it) -- and the one they leave alone, fu.
    int wA = 0, wB = 0, rAe = 0, rAo = 0, rBe = 0, rBo = 0, wC = 0, rC = 0;
    if (MFMA) {
        const int l = lane & 7, ch = lane >> 3;
        const int fp = ((l >> 1) ^ l) & 1, fq = l & 1, fu = (l ^ (l >> 2)) & 1;
        wA = ch * TURN_MRS * 8 + fp * 16 + fq * 32 + fu * 64;                      // + j1 * 128 + j2 * 256; j0 inside the 16 bytes
        wB = (8 + ch) * TURN_MRS * 8 + fp * 32 + fu * 64 + fq * 128;               // + j1 * 16 + j2 * 256
        // operand role: row / column lane & 15 = (mode lane & 7, low flip bit lane >> 3 & 1), k within the step = lane >> 4
        const int so = lane & 7, xbit = (lane >> 3) & 1, g = lane >> 4;
        const int baseA = so * TURN_MRS * 8 + (g >> 1) * 64 + (g & 1) * 8, baseB = baseA + 8 * TURN_MRS * 8;
        rAe = baseA + xbit * 16;      // slot c even: (c ^ xbit) * 16 = c * 16 + xbit * 16
        rAo = baseA - xbit * 16;      //        odd:                  = c * 16 - xbit * 16
        rBe = baseB + xbit * 16;
        rBo = baseB - xbit * 16;
        // sums: this lane holds column (s2 = lane & 7, xb bit 0 = lane >> 3 & 1), rows (lane >> 4) + 4 reg
        wC = ((g * 8 + so) * TURN_CS + xbit) * 8;
        rC = (s * 8 + s2) * TURN_CS * 8;
    }

    const double floor_r   = (double)CNF2_MINFACTOR_F - factor;
    const double floor_lse = n_ok > 0 ? floor_r + log((double)n_ok) : -INFINITY;

    d2v va[4], vb[4];
    d2v scA, scB;
    auto request = [&](int ml) {
        const double* wp = p.wbuf + ((size_t)job * p.wstride + ml) * CNF2_TURN_ROW;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            va[k] = __builtin_nontemporal_load((const d2v*)(wp + k * 128 + lane * 2));
            vb[k] = __builtin_nontemporal_load((const d2v*)(wp + 512 + k * 128 + lane * 2));
        }
        scA = *(const d2v*)(wp + 1024 + 4 * s);
        scB = *(const d2v*)(wp + 1024 + 4 * s2 + 2);
    };
#pragma unroll 1
    for (int ml = ml0; ml < ml1; ml++) {
        const int m = jb.first + ml;
        request(ml);
        const double K = scA.x * scB.x;         // scale of this lane's pair of modes: K * 2^E
        const int    E = (int)scA.y + (int)scB.y;
        double acc[16];
        if constexpr (MFMA) {
            wave_lds_fence();                   // the previous marker's read-back is done
#pragma unroll
            for (int k = 0; k < 4; k++) {
                *(d2v*)((char*)L + wA + ((k & 1) << 7) + ((k >> 1) << 8))  = va[k];      // j bits 1, 2 at 128, 256
                *(d2v*)((char*)L + wB + ((k & 1) << 4) + ((k >> 1) << 8))  = vb[k];      // j bits 1, 2 at 16, 256
            }
            wave_lds_fence();
            d4v c4[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) c4[mt][nt] = d4v{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 16; ks++) {
                // k-step = the four movable state coordinates (fp, fq, j1, j2); the B image orders them (j1, j2, fp, fq)
                const int fp = ks & 1, fq = (ks >> 1) & 1, j1 = (ks >> 2) & 1, j2 = ks >> 3;
                double    a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    // tile t of the operand has its second flip on: fq for A, j2 for B; the first flip (fp / j1) is the lane's
                    // bit, folded into the base address
                    a[t] = *(const double*)((const char*)L + (fp ? rAo : rAe) + fp * 16 + (fq ^ t) * 32 + j1 * 128 + j2 * 256);
                    b[t] = *(const double*)((const char*)L + (j1 ? rBo : rBe) + j1 * 16 + fp * 32 + fq * 128 + (j2 ^ t) * 256);
                }
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
                        c4[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b[nt], c4[mt][nt], 0, 0, 0);
            }
            wave_lds_fence();                   // every lane's operand reads are done: the image of the sums takes the LDS
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++)
                        *(double*)((char*)L + wC + ((reg & 1) * 32 * TURN_CS + ((2 * mt + (reg >> 1)) << 1) + 8 * nt) * 8) = c4[mt][nt][reg];
            wave_lds_fence();
#pragma unroll
            for (int x2 = 0; x2 < 8; x2++) {
                // the image keeps a cell's 16 sums in the order xb0 | xa << 1 | xb1 << 3 (the 16 lanes a ds_write_b64 serves
                // together then fall into different banks); x = xa | xb << 2
                const d2v t = *(const d2v*)((const char*)L + rC + x2 * 16);
                const int xa = x2 & 3, xb1 = x2 >> 2;
                acc[xa | (xb1 << 3)]     = t.x;
                acc[xa | 4 | (xb1 << 3)] = t.y;
            }
        } else {
            wave_lds_fence();                       // the previous marker's reads are done
    #pragma unroll
            for (int k = 0; k < 4; k++) {
                *(d2v*)(Lw + 16 * k)                = va[k];
                *(d2v*)(Lw + 8 * TURN_RS + 16 * k) = vb[k];
            }
            wave_lds_fence();
            double A[8][8];
    #pragma unroll
            for (int l = 0; l < 8; l++)
    #pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const d2v t = *(const d2v*)(Ap + jj * 16 + l * 2);
                    A[l][2 * jj]     = t.x;
                    A[l][2 * jj + 1] = t.y;
                }
    #pragma unroll
            for (int x = 0; x < 16; x++) acc[x] = 0.0;
            // B of one position (8 doubles) is requested a position ahead; the scheduling barriers keep the compiler from
            // pulling all 32 reads to the front (128 registers it does not have)
            d2v bn[4];
    #pragma unroll
            for (int jj = 0; jj < 4; jj++) bn[jj] = *(const d2v*)(Bp + jj * 16);
    #pragma unroll
            for (int l = 0; l < 8; l++) {
                d2v b[4];
    #pragma unroll
                for (int jj = 0; jj < 4; jj++) b[jj] = bn[jj];
                if (l < 7) {
    #pragma unroll
                    for (int jj = 0; jj < 4; jj++) bn[jj] = *(const d2v*)(Bp + jj * 16 + (l + 1) * 2);
                }
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int jj = 0; jj < 4; jj++) {
    #pragma unroll
                    for (int x = 0; x < 16; x++) {
                        // x bit 0 -> state bit 1 (position ^ 2), 1 -> bit 2 (position ^ 7), 2 -> bit 4, 3 -> bit 5 (registers)
                        const int Lx = ((x & 1) ? 2 : 0) ^ ((x & 2) ? 7 : 0), Jx = ((x >> 2) & 3) << 1;
                        acc[x] = fma(A[l ^ Lx][(2 * jj) ^ Jx], b[jj].x, acc[x]);
                        acc[x] = fma(A[l ^ Lx][(2 * jj + 1) ^ Jx], b[jj].y, acc[x]);
                    }
                }
                // the sums are only used under `if (full)` / `if (lse)`: without this the compiler sinks all 1 024 FMAs below
                // the reads and holds every B value in registers
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                                  "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15])
                             : : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        double* full = q.rawervals ? q.rawervals + ((size_t)jb.ind * p.n_markers + m) * 1024 : nullptr;
        double* lse  = q.turn_lse ? q.turn_lse + ((size_t)jb.ind * p.n_markers + m) * 128 : nullptr;
        if (full) {
            const double base = (double)E * 0.69314718055994530942 - factor;
#pragma unroll
            for (int x = 0; x < 16; x++) {
                const int    turn = tshift | ((x & 1) << 1) | (((x >> 1) & 1) << 2) | (((x >> 2) & 1) << 4) | (((x >> 3) & 1) << 5);
                const double kk   = acc[x] * K;
                full[turn * 8 + s] = (kk > 0.0) ? base + log_pos(kk) : floor_r;
            }
        }
        if (lse) {
            // per flip x: sum over the admissible modes s of kk_s 2^E_s.  Usual case: the terms are scaled to the largest
            // exponent E of the group's admissible modes (one max per marker).  Where a sum comes out below 2^-700 --
            // every term an exact zero, or the mode with the largest E has no likelihood at this flip and the others sit
            // hundreds of binary orders below it -- the flips are redone with a maximum per flip (turn_group_sums<true>).
            double w2[2];
            int    e2[2];
            turn_group_sums<false>(acc, K, E, s_ok, s, w2, e2);
            const bool thin = w2[0] < 0x1p-700 || w2[1] < 0x1p-700;
            if (__any(thin)) {
                double w2r[2];
                int    e2r[2];
                turn_group_sums<true>(acc, K, E, s_ok, s, w2r, e2r);
#pragma unroll
                for (int i = 0; i < 2; i++)
                    if (w2[i] < 0x1p-700) {
                        w2[i] = w2r[i];
                        e2[i] = e2r[i];
                    }
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int    x    = ((s & 1) << 3) | (((s >> 1) & 1) << 2) | (((s >> 2) & 1) << 1) | i;
                const int    turn = tshift | ((x & 1) << 1) | (((x >> 1) & 1) << 2) | (((x >> 2) & 1) << 4) | (((x >> 3) & 1) << 5);
                // no admissible mode with a positive sum: every one of them sits at the floor (cnF2freq.cpp:5802-5812)
                lse[turn] = (w2[i] > 0.0) ? ((double)e2[i] * 0.69314718055994530942 - factor) + log_pos(w2[i]) : floor_lse;
            }
        }
    }
}
void launch_turn_rows(const TurnParams& q, hipStream_t stream)
{
    const int per_block = CNF2_WAVES_PER_BLOCK * CNF2_TURN_SPAN;
    dim3 grid(q.n_jobs, (q.max_len + per_block - 1) / per_block);
    if (q.valu_form) hipLaunchKernelGGL(turn_rows_kernel<false>, grid, dim3(CNF2_BLOCK), 0, stream, q);
    else hipLaunchKernelGGL(turn_rows_kernel<true>, grid, dim3(CNF2_BLOCK), 0, stream, q);
}

int fb_fast_blocks_per_cu()
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fb_fast_kernel<true>, CNF2_BLOCK, 0) != hipSuccess) n = 2;
    return n < 1 ? 1 : n;
}

void launch_fb(const KernelParams& p, int grid, bool debug_store, hipStream_t stream)
{
    if (debug_store) hipLaunchKernelGGL(fb_kernel<true>, dim3(grid), dim3(CNF2_BLOCK), 0, stream, p);
    else hipLaunchKernelGGL(fb_kernel<false>, dim3(grid), dim3(CNF2_BLOCK), 0, stream, p);
}

void launch_emission(const KernelParams& p, int ind, int marker, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(emission_kernel, dim3(1), dim3(64), 0, stream, p, ind, marker, out);
}

void launch_xor_selftest(double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(xor_selftest_kernel, dim3(1), dim3(64), 0, stream, out);
}

int fb_blocks_per_cu()
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fb_kernel<false>, CNF2_BLOCK, 0) != hipSuccess) n = 2;
    return n < 1 ? 1 : n;
}

} // namespace cnf2
