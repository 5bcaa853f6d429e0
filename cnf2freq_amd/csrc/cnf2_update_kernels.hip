// cnf2_update_kernels.hip -- the per-iteration parameter updates on the device and the clock
// probe: kernels over cnf2_update.h.  Kept apart from cnf2_kernels.hip (the sweep and its consumers) so that the
// identity bench.py takes of the sweep kernels' sources does not change when these do.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cnf2_device.h"
#include "cnf2_update.h"

// Waves per SIMD the scout kernels are compiled for.  They hold one flow per lane and wait for each other's longest lanes and
// for their scattered loads (34 % of the wave cycles in s_waitcnt): at 4 waves per SIMD (<= 128 VGPRs: 2 - 20 of them spilled,
// outside the bisection loops) they are 9 - 19 % faster than at the 3 the compiler's own 140 - 167 registers allow
// (tools/ab_update.sh, profiles/r04_h_ab_update_occupancy.log).  The finish kernels do not gain (certainty_finish fits 128
// registers without a spill and runs the same 4.6 ms; haploweight_finish is at 123 already): left to the compiler.
#ifndef CNF2_SCOUT_WAVES
#define CNF2_SCOUT_WAVES 4
#endif
#define CNF2_SCOUT_OCC __attribute__((amdgpu_waves_per_eu(CNF2_SCOUT_WAVES, CNF2_SCOUT_WAVES)))
#ifdef CNF2_FINISH_WAVES
#define CNF2_FINISH_OCC __attribute__((amdgpu_waves_per_eu(CNF2_FINISH_WAVES, CNF2_FINISH_WAVES)))
#else
#define CNF2_FINISH_OCC
#endif
// The guided kernels hold a flow and what is known about it (~100 registers of state) around an evaluation that needs ~60:
// left alone the compiler takes 200 - 212 registers (2 waves per SIMD); at 3 waves (<= 168, two dozen spilled outside the
// quadrature loop) the lock-step kernel is 18 % faster, at 4 (<= 128, 95 spilled) 9 % (profiles/r05_d_ab_guided_occupancy.log)
#ifndef CNF2_GUIDED_WAVES
#define CNF2_GUIDED_WAVES 3
#endif
#define CNF2_GUIDED_OCC __attribute__((amdgpu_waves_per_eu(CNF2_GUIDED_WAVES, CNF2_GUIDED_WAVES)))

namespace cnf2 {

__device__ __forceinline__ void wave_lds_fence()
{
    // producer lanes -> consumer lanes of the SAME wave: order the LDS write before the reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The records a pass updates: all of them, or the listed ones (a rank of a multi-process run updates the records it owns;
// u.n_rec counts the listed records then).  Everything indexed by record -- accumulators, rows, scratch -- keeps the
// record's own index; only the flows' item numbers (and flow_out, which they index) count listed records.
__device__ __forceinline__ int update_rec(const UpdateParams& u, size_t listed) { return u.rec_list ? u.rec_list[listed] : (int)listed; }

// ---------------------------------------------------------------------------------------------------
// Per-iteration parameter updates on the device (cnf2_update.h; processinfprobs / updatehaploweights,
// cnF2freq.cpp:4179-4323, 4533-4734), after the sweep of chromosome `chrom` has been accounted for.
// ---------------------------------------------------------------------------------------------------
// CNF2_UPDATE_PLAIN -- the literal form: every same-sign step runs its quadrature, as the reference's cappedgd does.
// One thread per (record, marker of the chromosome): both sides in order (side 1 sees side 0's result only through
// the accumulators, which are per side), then the record's evidence at the marker is cleared.
__global__ __launch_bounds__(256) void certainty_update_kernel(UpdateParams u)
{
    const int len = u.last - u.first + 1;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)u.n_rec * len) return;
    const int r = update_rec(u, (size_t)(t / len)), m = u.first + (int)(t % len);
    double*   inf = u.acc_inf + ((size_t)r * u.n_markers + m) * 4;
    const size_t i = (size_t)u.row_of[r] * u.n_markers + m;
    const bool   has_prior = u.has_prior[r] != 0, empty = u.rec_empty[r] != 0;
    uint8_t ap = u.allele8[i];
    double2 su = u.sure[i];
    const uint8_t pap = has_prior ? u.prior_allele8[i] : 0;
    const double2 psu = has_prior ? u.prior_sure[i] : make_double2(0.0, 0.0);
    const StepControl sc = {u.scalefactor, u.entropyfactor};
    int  hits = 0;
    bool changed = false;
    for (int side = 0; side < 2; side++) {
        if (!(inf[side * 2] > 0) && !(inf[side * 2 + 1] > 0)) continue;
        SideState s;
        s.allele = side ? (ap >> 4) : (ap & 15);
        s.sure = side ? su.y : su.x;
        s.prior_allele = side ? (pap >> 4) : (pap & 15);
        s.prior_sure = side ? psu.y : psu.x;
        int    na;
        double ns;
        if (update_certainty(inf + side * 2, s, side, empty, has_prior, u.children[r], sc, &hits, &na, &ns, true)) {
            if (side) {
                ap = (uint8_t)((ap & 15) | (na << 4));
                su.y = ns;
            } else {
                ap = (uint8_t)((ap & 0xF0) | na);
                su.x = ns;
            }
            changed = true;
        }
    }
    inf[0] = inf[1] = inf[2] = inf[3] = 0.0;                  // infprobs[j][side].clear() (cnF2freq.cpp:4315)
    if (changed) {
        u.allele8[i] = ap;
        u.sure[i] = su;
    }
    if (hits) atomicAdd(u.hits, hits);
}

// One thread per (record, chromosome <= chrom): does the chromosome hold any information (haplocount != 0), and if
// so the phase-consistency ratio of every marker (relskewhmm; relhaplo is the constant the PlantImpute path keeps).
__global__ __launch_bounds__(64) void phase_ratio_kernel(UpdateParams u)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= u.n_rec * (u.chrom + 1)) return;
    const int r = update_rec(u, (size_t)(t / (u.chrom + 1))), c = t % (u.chrom + 1);
    const int c0 = u.chromstarts[c], c1 = u.chromstarts[c + 1];
    const double* hc = u.acc_hc + (size_t)r * u.n_markers;
    bool any = false;
    for (int k = c0; k < c1 && !any; k++) any = hc[k] != 0.0;
    u.anyinfo[(size_t)r * u.n_chrom + c] = any ? 1 : 0;
    if (!any) return;
    const double* hw = u.hw + (size_t)u.row_of[r] * u.n_markers;
    double* fw = u.fw + ((size_t)r * u.n_markers + c0) * 2;
    double* ratio = u.ratio + (size_t)r * u.n_markers + c0;
    // phase_ratio() of cnf2_update.h with a constant relhaplo.  The recurrences are serial per (record, chromosome) and a
    // thread's loads are a cache line apart from its neighbours': the weights (and, on the way back, the stored forward
    // pairs) are requested 8 markers at a time so that their latencies overlap; the arithmetic is the serial one.
    double s0 = 0.5, s1 = 0.5;
    const double n = u.relhaplo, nb = 1 - n;
    for (int m0 = c0; m0 < c1; m0 += 8) {
        double w8[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w8[i] = hw[m0 + i < c1 ? m0 + i : c1 - 1];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int m = m0 + i;
            if (m < c1) {
                const double w = w8[i];
                s0 *= fabs(1 - w);
                s1 *= fabs(0 - w);
                fw[(m - c0) * 2] = s0;
                fw[(m - c0) * 2 + 1] = s1;
                if (s0 + s1 < 1e-10) {
                    s0 *= 1e20;
                    s1 *= 1e20;
                }
                const double t0 = s0 * n + s1 * nb, t1 = s1 * n + s0 * nb;
                s0 = t0;
                s1 = t1;
            }
        }
    }
    s0 = s1 = 0.5;
    const int last = c1 - c0 - 1;
    ratio[last] = fw[last * 2 + 1] / (fw[last * 2] + fw[last * 2 + 1]);
    for (int mh = c1 - 2; mh >= c0; mh -= 8) {
        double w8[8], f8[8][2];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int m = mh - i < c0 ? c0 : mh - i;
            w8[i] = hw[m + 1];
            f8[i][0] = fw[(m - c0) * 2];
            f8[i][1] = fw[(m - c0) * 2 + 1];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int m = mh - i;
            if (m >= c0) {
                const double w = w8[i];
                s0 *= fabs(1 - w);
                s1 *= fabs(0 - w);
                const double t0 = s0 * n + s1 * nb, t1 = s1 * n + s0 * nb;
                s0 = t0;
                s1 = t1;
                if (s0 + s1 < 1e-10) {
                    s0 *= 1e20;
                    s1 *= 1e20;
                }
                const double r0 = s0 * f8[i][0], r1 = s1 * f8[i][1];
                ratio[m - c0] = r1 / (r0 + r1);
            }
        }
    }
}

// One thread per (record, marker of the chromosomes <= chrom)
__global__ __launch_bounds__(256) void haploweight_update_kernel(UpdateParams u)
{
    const int    upto = u.chromstarts[u.chrom + 1];
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)u.n_rec * upto) return;
    const int r = update_rec(u, (size_t)(t / upto)), m = (int)(t % upto);
    int c = 0;
    while (m >= u.chromstarts[c + 1]) c++;
    if (!u.anyinfo[(size_t)r * u.n_chrom + c]) return;
    const size_t i = (size_t)u.row_of[r] * u.n_markers + m;
    const double hw = u.hw[i];
    if (!(hw != 0.0 && hw != 1.0) || u.row_of[r] == 0) return;              // cnF2freq.cpp:4591; the shared blank row is never written
    const uint8_t ap = u.allele8[i];
    const double2 su = u.sure[i];
    const size_t  k = (size_t)r * u.n_markers + m;
    double hb = u.acc_hb[k], hcv = u.acc_hc[k];
    const StepControl sc = {u.scalefactor, u.entropyfactor};
    int hits = 0;
    const double nw = update_haploweight(hw, &hb, &hcv, ap & 15, ap >> 4, su.x, su.y, u.ratio[k], u.children[r],
                                         u.descendants[r], sc, false, &hits, true);
    u.acc_hb[k] = hb;
    u.acc_hc[k] = hcv;
    u.hw[i] = nw;
    if (hits) atomicAdd(u.hits, hits);
}

// ---- the same two updates as flow kernels --------------------------------------------------------------------
// A flow (cnf2_update.h) takes between 1 and 51 bisection steps, a step one gradient evaluation or none (the midpoint's
// sign settles it, or a bound does) or sixteen (a 15-point quadrature); with one thread per element nearly every
// wavefront waits for its longest lane, and most of the evaluations locate a root by bisection.  Two passes instead:
//   scout   one thread per flow (flow_scout): set-up, the flows pinned to their clamp, and every step that needs no
//           quadrature -- with the gradient shown monotone the root is found superlinearly and the bisection's own
//           midpoints are answered from what is known about it.  In the steady state of a run three flows in four
//           end here, after ~20 evaluations instead of ~65.  A flow that reaches a quadrature is set aside: its item,
//           the number of steps it has completed and their decisions (16 or 24 bytes).
//           The certainties' scout runs in two passes: the first stops every flow after FLOW_SCOUT_STEPS steps (early in a
//           run 98 % of them are at a quadrature by then), the second takes the few still scouting, 64 per wavefront.
//   finish  the flows set aside, as a persistent kernel: a wavefront runs ONE literal step (flow_advance: midpoint, the
//           bound, the quadrature) of 64 independent flows per round and hands a lane the next flow from a global counter
//           when its own has ended (refills are batched: a lane waits until FLOW_REFILL lanes are free, or nothing else
//           is running, because taking a flow up -- loads, the prior's logarithms, the gradient at the start, the replay
//           of its decisions -- is executed by the whole wave).  These flows spend their steps in quadratures, so the
//           lanes of a wave stay in step.
// Both passes make the decisions of the literal algorithm (CNF2_UPDATE_PLAIN); what differs is how much is computed.
#ifndef FLOW_REFILL
#define FLOW_REFILL 16      /* 4: +2.8 %, 8: +0.8 %, 32: +3.2 % on 40 iterations (tools/ab_scout.py) */
#endif

struct FlowTodo {
    unsigned long long item_steps;   // item << 6 | steps completed
    unsigned long long path;         // their decisions (FlowState::path)
};
struct HaploTodo {
    unsigned long long item_steps, path;
    double             similarity;   // what haplo_rewrite returned in the scout (the rewrite is not repeated)
};

__device__ __forceinline__ double haplo_similarity_of(const FlowTodo&) { return 0.0; }
__device__ __forceinline__ double haplo_similarity_of(const HaploTodo& e) { return e.similarity; }

// diagnostics: stats[0..3] += a, b, c, d summed over the wavefront
__device__ __forceinline__ void flow_stats(unsigned long long* stats, unsigned a_, unsigned b_, unsigned c_, unsigned d_)
{
    if (!stats) return;
    unsigned long long a = a_, b = b_, c = c_, d = d_;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
        c += __shfl_xor(c, o);
        d += __shfl_xor(d, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(stats + 0, a);
        atomicAdd(stats + 1, b);
        atomicAdd(stats + 2, c);
        atomicAdd(stats + 3, d);
    }
}

// the hit counter: one atomic per wavefront that has any
__device__ __forceinline__ void flow_hits(int* counter, int hits)
{
    int h = hits;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
    if ((threadIdx.x & 63) == 0 && h) atomicAdd(counter, h);
}

// The flows the scout set aside reach the lanes of a finish wavefront through a queue of the wave's own in LDS: the wave
// reserves FLOW_CHUNK slots of the scout's list at a time (an atomic on one address costs tens of nanoseconds across the
// 8 XCDs: a counter bumped once per refill was most of the run time of these kernels), reads them 64 at a time -- one
// coalesced load -- and queues the entries that hold a flow.  Free lanes take from the queue.
#define FLOW_CHUNK 4096ull
#define FLOW_QUEUE 128          /* entries: at most 63 left over plus 64 from one read */
template <class Entry>
struct FlowSupply {
    unsigned long long pos, end;      // the wave's reservation in the scout's list
    int                count;         // entries queued
    bool               more;          // the list has slots the wave has not read
};
// tops the queue up to at least `wanted` entries (or until the list is exhausted)
// An entry's item_steps: ~0 = nothing in this slot; bit 63 set = the flow is still scouting (the second scout pass takes it);
// bit 62 = it waits for the step-per-round kernels; neither = for the guided bisection.  want_flags: the kind a kernel takes.
#define FLOW_SCOUTING (1ull << 63)
// bit 62 set = set aside for the step-per-round kernels (the gradient is not known to be monotone, or the flow closes in on a
// root: cheap steps, many of them); neither bit = for the guided bisection
#define FLOW_LITERAL (1ull << 62)
#define FLOW_FLAGS (FLOW_SCOUTING | FLOW_LITERAL)
#define FLOW_ITEM(e) (((e).item_steps & ~FLOW_FLAGS) >> 6)
template <class Entry>
__device__ __forceinline__ void flow_supply(FlowSupply<Entry>* q, Entry* queue, unsigned long long* next, const Entry* todo,
                                            unsigned long long n_items, int wanted, unsigned long long want_flags = FLOW_LITERAL)
{
    const int lane = threadIdx.x & 63;
    while (q->more && q->count < wanted) {
        if (q->pos >= q->end) {                              // reservation used up: a new one
            // slots per reservation: FLOW_CHUNK on the scouts' long, sparse lists; on a short list (the packed lists of the guided
            // kernels: every slot a flow) as many as leave every wavefront of the launch two reservations -- with 4 096 slots
            // apiece a list of 10^6 flows would keep 250 of the 4 096 resident wavefronts busy and the rest idle
            unsigned long long chunk = (n_items / ((unsigned long long)gridDim.x * 2ull) + 63ull) & ~63ull;
            chunk = chunk < 64ull ? 64ull : (chunk > FLOW_CHUNK ? FLOW_CHUNK : chunk);
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(next, chunk);
            base = __shfl(base, 0);
            if (base >= n_items) {
                q->more = false;
                break;
            }
            q->pos = base;
            q->end = base + chunk < n_items ? base + chunk : n_items;
        }
        const unsigned long long slot = q->pos + lane;
        Entry e;
        e.item_steps = ~0ull;
        if (slot < q->end) e = todo[slot];
        const bool               holds = e.item_steps != ~0ull && (e.item_steps & FLOW_FLAGS) == want_flags;
        const unsigned long long mask = __ballot(holds);
        if (holds) queue[q->count + __popcll(mask & ((1ull << lane) - 1ull))] = e;
        q->count += __popcll(mask);
        q->pos = q->pos + 64 < q->end ? q->pos + 64 : q->end;
        wave_lds_fence();
    }
}
// entries for the lanes that want one: the last `n` queued, n = min(lanes that want, queued)
template <class Entry>
__device__ __forceinline__ bool flow_pop(FlowSupply<Entry>* q, const Entry* queue, bool want, Entry* e)
{
    const unsigned long long need = __ballot(want);
    const int rank = __popcll(need & ((1ull << (threadIdx.x & 63)) - 1ull)), n = __popcll(need);
    const int take = n < q->count ? n : q->count;
    const bool got = want && rank < take;
    if (got) *e = queue[q->count - 1 - rank];
    wave_lds_fence();
    q->count -= take;
    return got;
}

// item = ((r * len + mi) * 2 + side) * 2 + v: the flow of value v + 1 on one side of (record, marker);
// flow_out[item] = its new probability (0 where the value has no evidence)
// partner (optional): what the side's other value is to this one -- PARTNER_NONE: it has no evidence; PARTNER_MIRROR: its
// flow is this one's mirror image; PARTNER_APART: it has evidence but an allele value other than 1, 2 (the sex-marker
// sentinel) is called or prior on this side, for which the two starts and priors are not complementary: its own flow
// PARTNER_TIE: both values start at exactly 1/2 (unknown allele, or a certainty of 1/2): the two flows are mirror images here
// too, but where there is no net evidence both stay at 1/2 and the pick between them is made by the last bits of two
// independently rounded results -- such sides keep both flows (a pass of their own, u.mirror == 2) so that the labels of
// uninformative alleles come out as in the form that runs every flow.
enum { PARTNER_NONE = 0, PARTNER_MIRROR = 1, PARTNER_APART = 2, PARTNER_TIE = 3 };
__device__ __forceinline__ bool certainty_item(const UpdateParams& u, unsigned long long item, const StepControl& sc, CertaintyFlow* c,
                                               int* partner = nullptr)
{
    const int          len = u.last - u.first + 1;
    const int          v = (int)(item & 1), side = (int)((item >> 1) & 1);
    const unsigned long long e = item >> 2;
    const int          r = update_rec(u, (size_t)(e / len)), m = u.first + (int)(e % len);
    const double*      inf = u.acc_inf + ((size_t)r * u.n_markers + m) * 4 + side * 2;
    const double       pair[2] = {inf[0], inf[1]};
    const size_t       i = (size_t)u.row_of[r] * u.n_markers + m;
    const bool         has_prior = u.has_prior[r] != 0;
    const uint8_t      ap = u.allele8[i], pap = has_prior ? u.prior_allele8[i] : 0;
    const double2      su = u.sure[i];
    const double2      psu = has_prior ? u.prior_sure[i] : make_double2(0.0, 0.0);
    SideState s;
    s.allele = side ? (ap >> 4) : (ap & 15);
    s.sure = side ? su.y : su.x;
    s.prior_allele = side ? (pap >> 4) : (pap & 15);
    s.prior_sure = side ? psu.y : psu.x;
    if (partner)
        *partner = !(pair[0] > 0 && pair[1] > 0) ? PARTNER_NONE
                 : ((s.allele > 2 || s.prior_allele > 2) ? PARTNER_APART : ((s.allele == 0 || s.sure == 0.5) ? PARTNER_TIE : PARTNER_MIRROR));
    return certainty_flow_setup(pair, v, s, u.children[r], sc, c);
}

// Mirror (u.mirror): the two values of a side start at y and 1 - y, their evidence shares are g and h - g, their priors p and
// 1 - p: the gradient of one is the negative of the other's at the mirrored position, G_2(1 - x) = -G_1(x) (the data term
// swaps a and b, logit changes sign), so the second flow is the first one mirrored and ends at 1 - its end, with the same
// capped moves.  Where both values have evidence (the side's called and prior alleles being 0, 1 or 2 and its start not
// exactly 1/2) only the called value's flow is run; the other gets 1 - result and the hits count twice.  The reference runs both (cnF2freq.cpp:4222-4290) and lands within rounding of this; CNF2_CERTAINTY_BOTH=1 in the
// environment (and CNF2_UPDATE_PLAIN) keep that literal form, which the bit-exactness tests compare.
// which item a position (item >> 1) runs under the mirror: the value the side is called as (the one that starts at or above
// 1/2 and, as a rule, wins the pick of cnF2freq.cpp:4292-4300: its result is then bit for bit the literal one, and only the
// value that loses is a mirror image), value 1 for an unknown allele; a value without evidence yields to the other
__device__ __forceinline__ unsigned long long certainty_mirror_item(const UpdateParams& u, unsigned long long pos)
{
    const int          len = u.last - u.first + 1;
    const int          side = (int)(pos & 1);
    const unsigned long long e = pos >> 1;
    const int          r = update_rec(u, (size_t)(e / len)), m = u.first + (int)(e % len);
    const double*      inf = u.acc_inf + ((size_t)r * u.n_markers + m) * 4 + side * 2;
    const size_t       i = (size_t)u.row_of[r] * u.n_markers + m;
    const uint8_t      ap = u.allele8[i];
    const double2      su = u.sure[i];
    const int          allele = side ? (ap >> 4) : (ap & 15);
    const double       sure = side ? su.y : su.x;
    // the starting probability of value 1: |[allele == 1] - sure|, 1/2 for an unknown allele
    const double       y0 = allele == 0 ? 0.5 : fabs((allele == 1 ? 1.0 : 0.0) - sure);
    int                v = y0 >= 0.5 ? 0 : 1;
    if (!(inf[v] > 0)) v ^= 1;
    return pos * 2 + v;
}
// the pass of the ties looks at every item: most are sorted out from the called allele and its certainty alone
__device__ __forceinline__ bool certainty_may_tie(const UpdateParams& u, unsigned long long item)
{
    const int          len = u.last - u.first + 1;
    const int          side = (int)((item >> 1) & 1);
    const unsigned long long e = item >> 2;
    const int          r = update_rec(u, (size_t)(e / len)), m = u.first + (int)(e % len);
    const size_t       i = (size_t)u.row_of[r] * u.n_markers + m;
    const uint8_t      ap = u.allele8[i];
    const double2      su = u.sure[i];
    return (side ? (ap >> 4) : (ap & 15)) == 0 || (side ? su.y : su.x) == 0.5;
}
// the result of a flow (and of its mirror image)
__device__ __forceinline__ void certainty_store(const UpdateParams& u, double* flow_out, unsigned long long item, int partner, double out,
                                                int* hits, int flow_hits_)
{
    flow_out[item] = out;
    if (u.mirror == 1 && partner == PARTNER_MIRROR) {
        flow_out[item ^ 1] = 1.0 - out;
        *hits += flow_hits_;
    } else if (u.mirror == 1 && partner == PARTNER_NONE) {
        flow_out[item ^ 1] = 0.0;
    }
}

#ifndef HAPLO_SCOUT_STEPS
#define HAPLO_SCOUT_STEPS 8     /* steps of the weights' first scout pass (iteration probe's update passes / late iterations of a 1 500-family run against one pass: 1: +4 % / +4.5 %, 4: -1.5 % / -1.3 %, 8: -6 % / -2.6 %, 12: -5 % / 0, 16: -4.5 % / 0, 24: -4.5 % / +0.5 %; profiles/r05_zz_ab_haploweight_scout_two_passes.log) */
#endif
#ifndef FLOW_SCOUT_STEPS
#define FLOW_SCOUT_STEPS 12     /* steps of the first scout pass (round 4, against 8: 2: +5 %, 3: +2.5 %, 5: +2 %, 12: -0.3 %, 16: +0.6 % on 40 iterations, tools/ab_scout.py; round 5, with the hand-over: 6: +1 % / +0.3 %, 12: -1.9 % / -0.4 % on the probe's update passes / late iterations, profiles/r05_zz_ab_haploweight_scout_two_passes.log) */
#endif
template <bool TWO_PASSES>
__global__ CNF2_SCOUT_OCC __launch_bounds__(256) void certainty_scout_kernel(UpdateParams u, unsigned long long item0, unsigned long long n_items,
                                                              double* flow_out, FlowTodo* todo)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const StepControl        sc = {u.scalefactor, u.entropyfactor};
    int      hits = 0, evals = 0;
    unsigned n_flows = 0, n_pinned = 0, n_done = 0;
    bool     aside = false;
    FlowTodo e;
    if (t < n_items) {
        // item0, n_items count items, or positions (items >> 1) in the mirror pass (u.mirror == 1); the pass of the ties
        // (u.mirror == 2) goes over the items again and takes the sides the mirror pass left out, each value on its own
        const unsigned long long item = u.mirror == 1 ? certainty_mirror_item(u, item0 + t) : item0 + t;
        CertaintyFlow c;
        int           both = PARTNER_NONE;
        const bool    look = u.mirror != 2 || certainty_may_tie(u, item);
        const bool    present = look && certainty_item(u, item, sc, &c, &both);
        if (!look || (u.mirror == 1 && both == PARTNER_TIE) || (u.mirror == 2 && both != PARTNER_TIE)) {
            // not this pass's
        } else if (!present) {
            flow_out[item] = 0.0;
            if (u.mirror == 1) flow_out[item ^ 1] = 0.0;
        } else {
            if (u.mirror == 1 && both == PARTNER_APART) {            // rare: the other value's own flow, literally and to its end
                CertaintyFlow c2;
                certainty_item(u, item ^ 1, sc, &c2);
                FlowState f2;
                auto grad2 = [&](double x) CNF2_LI { return certainty_rgradient(c2, x); };
                flow_begin(&f2, grad2, c2.curprob, c2.epsilon, sc.scalefactor, false);
                while (flow_advance(&f2, grad2, sc.scalefactor)) {}
                flow_out[item ^ 1] = flow_end(f2, sc.scalefactor, &hits, false);
            }
            FlowState f;
            auto grad = [&](double x) CNF2_LI { return certainty_rgradient(c, x); };
            flow_begin(&f, grad, c.curprob, c.epsilon, sc.scalefactor, false);
            n_flows = 1;
            if (f.pinned) {                 // no gradient evaluations left
                while (flow_advance(&f, grad, sc.scalefactor)) {}
                int          h = 0;
                const double out = flow_end(f, sc.scalefactor, &h, false);
                hits += h;
                certainty_store(u, flow_out, item, both, out, &hits, h);
                n_pinned = 1;
            } else {
                const SlopeTerms st = certainty_slope(c);
                const int        r = flow_scout(&f, grad, st, sc.scalefactor, &evals, TWO_PASSES ? FLOW_SCOUT_STEPS : 1 << 30, !u.literal_finish, !TWO_PASSES);
#ifdef CNF2_X_RC_HIST
                if (u.stats && (r == 2 || r >= 4)) atomicAdd(u.stats + 20 + (r == 2 ? 0 : r - 3), 1ull);
#endif
                if (r == 0) {
                    int          h = 0;
                    const double out = flow_end(f, sc.scalefactor, &h, false);
                    hits += h;
                    certainty_store(u, flow_out, item, both, out, &hits, h);
                    n_done = 1;
                } else {
                    aside = true;
                    e.item_steps = (item << 6) | (unsigned long long)f.it | (r == 3 ? FLOW_SCOUTING : (r == 6 || u.literal_finish ? FLOW_LITERAL : 0ull));
                    e.path = f.path;
                }
            }
        }
    }
    if (t < n_items) {
        if (!aside) e.item_steps = ~0ull;                     // nothing set aside in this slot
        todo[t] = e;
    }
    // the pass of the ties mostly sets nothing aside: its second scout pass and its finish pass return at once unless told
    if (u.mirror == 2 && aside) u.flow_next[(e.item_steps & FLOW_SCOUTING) ? 27 : 26] = 1ull;
    flow_hits(u.hits, hits);
    flow_stats(u.stats, n_flows, (unsigned)evals, n_done, n_pinned);
}

// Second scout pass: the flows the first left scouting, 64 of them per wavefront and round (persistent, the finish pass's
// supply); a flow ends here or its slot is rewritten for the finish pass.
__global__ CNF2_SCOUT_OCC __launch_bounds__(64) void certainty_scout2_kernel(UpdateParams u, unsigned long long* next, FlowTodo* todo,
                                                              unsigned long long item0, unsigned long long n_items, double* flow_out)
{
    const StepControl        sc = {u.scalefactor, u.entropyfactor};
    __shared__ FlowTodo  queue[FLOW_QUEUE];
    FlowSupply<FlowTodo> q = {0ull, 0ull, 0, true};
    int      hits = 0, evals_all = 0;
    unsigned n_done = 0;
    if (u.mirror == 2 && u.flow_next[27] == 0ull) return;
    for (;;) {
        flow_supply(&q, queue, next, (const FlowTodo*)todo, n_items, 64, FLOW_SCOUTING);
        if (q.count == 0) break;
        FlowTodo e;
        const bool got = flow_pop(&q, queue, true, &e);
        if (got) {
            const unsigned long long item = FLOW_ITEM(e);
            CertaintyFlow c;
            int both = PARTNER_NONE;
            certainty_item(u, item, sc, &c, &both);
            FlowState f;
            auto grad = [&](double x) CNF2_LI { return certainty_rgradient(c, x); };
            flow_begin(&f, grad, c.curprob, c.epsilon, sc.scalefactor, false);
            flow_replay(&f, e.path, (int)(e.item_steps & 63));
            const SlopeTerms st = certainty_slope(c);
            int evals = 0;
            FlowTodo out;
            const int rs = flow_scout(&f, grad, st, sc.scalefactor, &evals, 1 << 30, !u.literal_finish);
#ifdef CNF2_X_RC_HIST
            if (u.stats && (rs == 2 || rs >= 4)) atomicAdd(u.stats + 20 + (rs == 2 ? 0 : rs - 3), 1ull);
#endif
            if (rs == 0) {
                int          h = 0;
                const double res = flow_end(f, sc.scalefactor, &h, false);
                hits += h;
                certainty_store(u, flow_out, item, both, res, &hits, h);
                out.item_steps = ~0ull;
                out.path = 0;
                n_done++;
            } else {
                out.item_steps = (item << 6) | (unsigned long long)f.it | (rs == 6 || u.literal_finish ? FLOW_LITERAL : 0ull);
                out.path = f.path;
                if (u.mirror == 2) u.flow_next[26] = 1ull;
            }
            todo[(u.mirror == 1 ? item >> 1 : item) - item0] = out;
            evals_all += evals;
        }
    }
    if (hits) atomicAdd(u.hits, hits);
    flow_stats(u.stats, 0u, (unsigned)evals_all, n_done, 0u);
}

__global__ CNF2_FINISH_OCC __launch_bounds__(64) void certainty_finish_kernel(UpdateParams u, unsigned long long* next, const FlowTodo* todo,
                                                              unsigned long long n_max, double* flow_out, const unsigned long long* n_in,
                                                              unsigned long long want_flags)
{
    const StepControl        sc = {u.scalefactor, u.entropyfactor};
    const unsigned long long n_items = n_in ? (*n_in < n_max ? *n_in : n_max) : n_max;      // n_in: the list's length where only the device knows it
    __shared__ FlowTodo  queue[FLOW_QUEUE];
    FlowSupply<FlowTodo> q = {0ull, 0ull, 0, true};
    if (u.mirror == 2 && u.flow_next[26] == 0ull) return;
    bool               have = false;
    int                both = PARTNER_NONE;
    unsigned long long item = 0;
    CertaintyFlow      c;
    FlowState          f;
    SlopeTerms         st;
    int                hits = 0;
    unsigned           n_steps = 0, n_rounds = 0, n_quads = 0, n_why1 = 0;
    auto grad = [&](double x) CNF2_LI { return certainty_rgradient(c, x); };
    auto bound = [&](double xa, double xb, double pc, double lim) CNF2_LI { return flow_time_under(st, xa, xb, pc, lim); };
    for (;;) {
        const int busy = __popcll(__ballot(have));
        if ((q.more || q.count > 0) && busy <= 64 - FLOW_REFILL) {
            flow_supply(&q, queue, next, todo, n_items, 64 - busy, want_flags);
            FlowTodo e;
            if (flow_pop(&q, queue, !have, &e)) {
                item = FLOW_ITEM(e);
                certainty_item(u, item, sc, &c, &both);
                flow_begin(&f, grad, c.curprob, c.epsilon, sc.scalefactor, false);
                flow_replay(&f, e.path, (int)(e.item_steps & 63));
                st = certainty_slope(c);
                have = true;
            }
            continue;
        }
        if (busy == 0) break;               // nothing running and nothing left
        n_rounds++;
        if (have) {
            n_steps++;
            if (!flow_advance(&f, grad, sc.scalefactor, bound)) {
                int          h = 0;
                const double res = flow_end(f, sc.scalefactor, &h, false);
                hits += h;
                certainty_store(u, flow_out, item, both, res, &hits, h);
                have = false;
                n_quads += f.quads;
                n_why1 += f.why == 1;
            }
        }
    }
    if (hits) atomicAdd(u.hits, hits);
    flow_stats(u.stats ? u.stats + 8 : nullptr, n_steps, n_rounds, n_quads, n_why1);
}

// one thread per (record, marker of the chromosome): cnF2freq.cpp:4292-4322 from the flows' results
__global__ __launch_bounds__(256) void certainty_pick_kernel(UpdateParams u, const double* flow_out)
{
    const int    len = u.last - u.first + 1;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)u.n_rec * len) return;
    const int r = update_rec(u, (size_t)(t / len)), m = u.first + (int)(t % len);
    double*   inf = u.acc_inf + ((size_t)r * u.n_markers + m) * 4;
    const double4 a = *(const double4*)inf, o = *(const double4*)(flow_out + t * 4);
    if (a.x == 0.0 && a.y == 0.0 && a.z == 0.0 && a.w == 0.0) return;          // nothing was added, nothing to clear
    const size_t i = (size_t)u.row_of[r] * u.n_markers + m;
    const bool   has_prior = u.has_prior[r] != 0, empty = u.rec_empty[r] != 0;
    uint8_t ap = u.allele8[i];
    double2 su = u.sure[i];
    bool    changed = false;
    const double in2[2][2] = {{a.x, a.y}, {a.z, a.w}}, out2[2][2] = {{o.x, o.y}, {o.z, o.w}};
#pragma unroll
    for (int side = 0; side < 2; side++) {
        int    na;
        double ns;
        if (!(in2[side][0] > 0) && !(in2[side][1] > 0)) continue;
        if (certainty_pick(in2[side], out2[side], side, empty, has_prior, &na, &ns)) {
            if (side) {
                ap = (uint8_t)((ap & 15) | (na << 4));
                su.y = ns;
            } else {
                ap = (uint8_t)((ap & 0xF0) | na);
                su.x = ns;
            }
            changed = true;
        }
    }
    *(double4*)inf = make_double4(0.0, 0.0, 0.0, 0.0);          // infprobs[j][side].clear() (cnF2freq.cpp:4315)
    if (changed) {
        u.allele8[i] = ap;
        u.sure[i] = su;
    }
}

// item = r * upto + m over the markers of the chromosomes <= chrom.  false: nothing to update there
__device__ __forceinline__ bool haplo_item(const UpdateParams& u, unsigned long long item, size_t* row_i, size_t* k, int* r_out)
{
    const int upto = u.chromstarts_host_upto;
    const int r = update_rec(u, (size_t)(item / upto)), m = (int)(item % upto);
    int       c = 0;
    while (m >= u.chromstarts[c + 1]) c++;
    *row_i = (size_t)u.row_of[r] * u.n_markers + m;
    *k = (size_t)r * u.n_markers + m;
    *r_out = r;
    const double hw = u.hw[*row_i];
    return u.anyinfo[(size_t)r * u.n_chrom + c] && hw != 0.0 && hw != 1.0 && u.row_of[r] != 0;          // cnF2freq.cpp:4591
}

__global__ CNF2_SCOUT_OCC __launch_bounds__(256) void haploweight_scout_kernel(UpdateParams u, unsigned long long item0, unsigned long long n_items,
                                                                HaploTodo* todo)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const StepControl        sc = {u.scalefactor, u.entropyfactor};
    int      hits = 0, evals = 0;
    unsigned n_flows = 0, n_pinned = 0, n_done = 0;
    size_t   row_i, k;
    int      r;
    bool      aside = false;
    HaploTodo e;
    if (t < n_items && haplo_item(u, item0 + t, &row_i, &k, &r)) {
        const double  hw = u.hw[row_i];
        const uint8_t ap = u.allele8[row_i];
        const double2 su = u.sure[row_i];
        double hb = u.acc_hb[k], hcv = u.acc_hc[k];
        const double similarity = haplo_rewrite(hw, &hb, &hcv, haplo_similarity(ap & 15, ap >> 4, su.x, su.y));
        u.acc_hb[k] = hb;
        u.acc_hc[k] = hcv;
        HaploFlow h;
        haplo_flow_terms(hw, hb, hcv, similarity, u.ratio[k], u.children[r], u.descendants[r], sc, &h);
        FlowState f;
        auto grad = [&](double x) CNF2_LI { return haplo_rgradient(h, x); };
        flow_begin(&f, grad, hw, h.epsilon, sc.scalefactor, false);
        n_flows = 1;
        if (f.pinned) {
            while (flow_advance(&f, grad, sc.scalefactor)) {}
            u.hw[row_i] = flow_end(f, sc.scalefactor, &hits, false);
            n_pinned = 1;
        } else {
            const SlopeTerms st = haplo_slope(h);
            // Two passes like the certainties' since the flows that reach their step size leave at step 0 (hand_over): what stays
            // scouts for dozens of steps, a few lanes of every wavefront, and the second pass packs those 64 to a wavefront.
            // (Round 4, before the hand-over, nearly every flow was still going after the first pass: one pass was 1.2 % faster.)
            const int rs = flow_scout(&f, grad, st, sc.scalefactor, &evals, u.scout_passes == 2 ? HAPLO_SCOUT_STEPS : 1 << 30, !u.literal_finish,
                                      u.scout_passes != 2);
            if (rs == 0) {
                u.hw[row_i] = flow_end(f, sc.scalefactor, &hits, false);
                n_done = 1;
            } else {
                aside = true;
                e.item_steps = ((item0 + t) << 6) | (unsigned long long)f.it |
                               (rs == 3 ? FLOW_SCOUTING : (rs == 6 || u.literal_finish ? FLOW_LITERAL : 0ull));
                e.path = f.path;
                e.similarity = similarity;
            }
        }
    }
    if (t < n_items) {
        if (!aside) e.item_steps = ~0ull;                     // nothing set aside in this slot
        todo[t] = e;
    }
    flow_hits(u.hits, hits);
    flow_stats(u.stats ? u.stats + 4 : nullptr, n_flows, (unsigned)evals, n_done, n_pinned);
}

// Second scout pass of the weights: the flows the first pass left scouting after HAPLO_SCOUT_STEPS steps, 64 of them per wavefront
// (persistent, the finish pass's supply); a flow ends here or its slot is rewritten for the finish pass.
__global__ CNF2_SCOUT_OCC __launch_bounds__(64) void haploweight_scout2_kernel(UpdateParams u, unsigned long long* next, HaploTodo* todo,
                                                                unsigned long long item0, unsigned long long n_items)
{
    const StepControl         sc = {u.scalefactor, u.entropyfactor};
    __shared__ HaploTodo  queue[FLOW_QUEUE];
    FlowSupply<HaploTodo> q = {0ull, 0ull, 0, true};
    int      hits = 0, evals_all = 0;
    unsigned n_done = 0;
    for (;;) {
        flow_supply(&q, queue, next, (const HaploTodo*)todo, n_items, 64, FLOW_SCOUTING);
        if (q.count == 0) break;
        HaploTodo e;
        if (flow_pop(&q, queue, true, &e)) {
            const unsigned long long item = FLOW_ITEM(e);
            size_t row_i, k;
            int    r;
            haplo_item(u, item, &row_i, &k, &r);
            const double hw = u.hw[row_i];
            HaploFlow    h;
            haplo_flow_terms(hw, u.acc_hb[k], u.acc_hc[k], e.similarity, u.ratio[k], u.children[r], u.descendants[r], sc, &h);
            FlowState f;
            auto grad = [&](double x) CNF2_LI { return haplo_rgradient(h, x); };
            flow_begin(&f, grad, hw, h.epsilon, sc.scalefactor, false);
            flow_replay(&f, e.path, (int)(e.item_steps & 63));
            const SlopeTerms st = haplo_slope(h);
            int       evals = 0;
            HaploTodo out;
            const int rs = flow_scout(&f, grad, st, sc.scalefactor, &evals, 1 << 30, !u.literal_finish);
            if (rs == 0) {
                u.hw[row_i] = flow_end(f, sc.scalefactor, &hits, false);
                out.item_steps = ~0ull;
                out.path = 0;
                n_done++;
            } else {
                out.item_steps = (item << 6) | (unsigned long long)f.it | (rs == 6 || u.literal_finish ? FLOW_LITERAL : 0ull);
                out.path = f.path;
            }
            out.similarity = e.similarity;
            todo[item - item0] = out;
            evals_all += evals;
        }
    }
    if (hits) atomicAdd(u.hits, hits);
    flow_stats(u.stats ? u.stats + 4 : nullptr, 0u, (unsigned)evals_all, n_done, 0u);
}

__global__ CNF2_FINISH_OCC __launch_bounds__(64) void haploweight_finish_kernel(UpdateParams u, unsigned long long* next, const HaploTodo* todo,
                                                                unsigned long long n_max, const unsigned long long* n_in, unsigned long long want_flags)
{
    const StepControl        sc = {u.scalefactor, u.entropyfactor};
    const unsigned long long n_items = n_in ? (*n_in < n_max ? *n_in : n_max) : n_max;
    __shared__ HaploTodo  queue[FLOW_QUEUE];
    FlowSupply<HaploTodo> q = {0ull, 0ull, 0, true};
    bool               have = false;
    size_t             row_i = 0;
    HaploFlow          h;
    FlowState          f;
    SlopeTerms         st;
    int                hits = 0;
    unsigned           n_steps = 0, n_rounds = 0, n_quads = 0, n_why1 = 0;
    auto grad = [&](double x) CNF2_LI { return haplo_rgradient(h, x); };
    auto bound = [&](double xa, double xb, double pc, double lim) CNF2_LI { return flow_time_under(st, xa, xb, pc, lim); };
    for (;;) {
        const int busy = __popcll(__ballot(have));
        if ((q.more || q.count > 0) && busy <= 64 - FLOW_REFILL) {
            flow_supply(&q, queue, next, todo, n_items, 64 - busy, want_flags);
            HaploTodo e;
            if (flow_pop(&q, queue, !have, &e)) {
                size_t k;
                int    r;
                haplo_item(u, FLOW_ITEM(e), &row_i, &k, &r);
                const double hw = u.hw[row_i];
                haplo_flow_terms(hw, u.acc_hb[k], u.acc_hc[k], e.similarity, u.ratio[k], u.children[r], u.descendants[r], sc, &h);
                flow_begin(&f, grad, hw, h.epsilon, sc.scalefactor, false);
                flow_replay(&f, e.path, (int)(e.item_steps & 63));
                st = haplo_slope(h);
                have = true;
            }
            continue;
        }
        if (busy == 0) break;
        n_rounds++;
        if (have) {
            n_steps++;
            if (!flow_advance(&f, grad, sc.scalefactor, bound)) {
                u.hw[row_i] = flow_end(f, sc.scalefactor, &hits, false);
                have = false;
                n_quads += f.quads;
                n_why1 += f.why == 1;
            }
        }
    }
    if (hits) atomicAdd(u.hits, hits);
    flow_stats(u.stats ? u.stats + 12 : nullptr, n_steps, n_rounds, n_quads, n_why1);
}

#ifdef CNF2_X_GUIDED_PERSISTENT
// ---- the guided finish: the flows the scouts set aside, with the literal decisions from three or four quadratures each ----
// (cnf2_update.h, "the guided bisection").  Persistent like the finish kernels above -- a wavefront is 64 independent workers
// over a queue of its own -- but a round is not "the next bisection step": a lane asks for the literal evaluation of ONE
// point (flow_guide_next: the bisection's midpoints that facts cover are decided on the way, the point asked for is the one
// that settles most), all lanes evaluate theirs together (flow_point: the midpoint's gradient and the 15 nodes, the same
// arithmetic as a step of cappedgd), and the result goes back as a fact (flow_guide_feed).  The decisions, and with them the
// results, are those of the literal kernels to the bit (tests/test_host_update.py on the host; the GPU suite compares the
// passes); what changes is that a flow which reaches its step size takes 3 - 4 rounds instead of ~10.
// KIND 0: genotype certainties (FlowTodo), 1: haplotype weights (HaploTodo).
// A flow lives 3 - 5 rounds here and taking one up (loads, the prior's logarithms, the gradient at the start, the replay, the
// slope bound, the seed) is executed by the whole wavefront for the lanes that are free: flows are taken up in batches --
// when at most 64 - GUIDED_REFILL lanes are still busy -- so that the set-up runs nearly full and a batch's lanes go
// through their rounds together.
#ifndef GUIDED_REFILL
#define GUIDED_REFILL 48
#endif
template <int KIND, class Entry>
__global__ CNF2_GUIDED_OCC __launch_bounds__(64) void guided_finish_kernel(UpdateParams u, unsigned long long* next, const Entry* todo,
                                                                           const unsigned long long* n_in, unsigned long long n_max, double* flow_out)
{
    const StepControl  sc = {u.scalefactor, u.entropyfactor};
    const unsigned long long n_items = n_in ? (*n_in < n_max ? *n_in : n_max) : n_max;
    __shared__ Entry   queue[FLOW_QUEUE];
    FlowSupply<Entry>  q = {0ull, 0ull, 0, true};
    if (KIND == 0 && u.mirror == 2 && u.flow_next[26] == 0ull) return;
    bool               have = false;
    int                both = PARTNER_NONE;
    unsigned long long item = 0;
    size_t             row_i = 0;
    CertaintyFlow      c;
    HaploFlow          h;
    FlowState          f;
    FlowGuide          g;
    SlopeTerms         st;
    int                hits = 0;
    unsigned           n_points = 0, n_rounds = 0, n_evals = 0, n_why1 = 0;
    auto grad = [&](double x) CNF2_LI { return KIND == 0 ? certainty_rgradient(c, x) : haplo_rgradient(h, x); };
    for (;;) {
        const int busy = __popcll(__ballot(have));
        if ((q.more || q.count > 0) && busy <= 64 - GUIDED_REFILL) {
            flow_supply(&q, queue, next, todo, n_items, 64 - busy, 0ull);
            Entry e;
            if (flow_pop(&q, queue, !have, &e)) {
                double start;
                if (KIND == 0) {
                    item = FLOW_ITEM(e);
                    certainty_item(u, item, sc, &c, &both);
                    st = certainty_slope(c);
                    start = c.curprob;
                    f.epsilon = c.epsilon;
                } else {
                    size_t k;
                    int    r;
                    haplo_item(u, FLOW_ITEM(e), &row_i, &k, &r);
                    start = u.hw[row_i];
                    haplo_flow_terms(start, u.acc_hb[k], u.acc_hc[k], haplo_similarity_of(e), u.ratio[k], u.children[r], u.descendants[r], sc, &h);
                    st = haplo_slope(h);
                    f.epsilon = h.epsilon;
                }
                flow_begin(&f, grad, start, f.epsilon, sc.scalefactor, false);
                flow_replay(&f, e.path, (int)(e.item_steps & 63));
                flow_guide_begin(&g);
                flow_guide_try_mono(f, &g, st);
                g.mono_tried = true;
                flow_guide_seed(f, &g, grad, st, sc.scalefactor);
                have = true;
            }
            continue;
        }
        if (busy == 0) break;               // nothing running and nothing left
        n_rounds++;
        int    rc = 0;
        double p = 0.0;
        if (have) {
            rc = flow_guide_next(&f, &g, st, sc.scalefactor, &p);
            if (rc == 0) {
                int          hh = 0;
                const double res = flow_end(f, sc.scalefactor, &hh, false);
                hits += hh;
                if (KIND == 0) certainty_store(u, flow_out, item, both, res, &hits, hh);
                else u.hw[row_i] = res;
                have = false;
                n_points += g.points;
                n_evals += g.evals;
                n_why1 += f.why == 1;
            }
        }
        if (rc == 2) flow_guide_feed(f, &g, p, flow_point(f, grad, p, sc.scalefactor), sc.scalefactor);
        else if (rc == 3) flow_guide_feed_clear(f, &g, p, flow_pace(grad, p, f.epsilon));
    }
    if (hits) atomicAdd(u.hits, hits);
    flow_stats(u.stats ? u.stats + (KIND == 0 ? 8 : 12) : nullptr, n_points, n_rounds, n_evals, n_why1);
}

#endif  // CNF2_X_GUIDED_PERSISTENT

// The scouts' list has a slot per flow they looked at, and most slots are empty (a certainty flow exists only where a
// record has evidence at the marker: a quarter of the slots of a pass; the flows that ended in the scout are gone too).
// Three small kernels pack the entries that hold a flow into a second list, in no particular order (a flow's result does
// not depend on who runs it or when): entries per wavefront-sized stretch of 256 slots, an exclusive scan of those counts
// by one block, the scatter.  What they move is 16 - 24 bytes per slot: a fraction of a millisecond per pass.
template <class Entry>
__device__ __forceinline__ bool todo_holds(const Entry& e) { return e.item_steps != ~0ull && !(e.item_steps & FLOW_FLAGS); }
#define TODO_STRETCH 256ull          /* slots a wavefront looks at (4 reads of 64) */
#define TODO_BLOCK 4096ull           /* slots of a block of 16 wavefronts: the unit the counts are kept for (a list of 2^26 slots: 16 384 counts) */
template <class Entry>
__device__ __forceinline__ unsigned todo_wave_count(const Entry* todo, unsigned long long n, unsigned long long wave)
{
    const int lane = threadIdx.x & 63;
    unsigned  c = 0;
    for (int k = 0; k < 4; k++) {
        const unsigned long long slot = wave * TODO_STRETCH + (unsigned long long)k * 64 + lane;
        bool holds = false;
        if (slot < n) holds = todo_holds(todo[slot]);
        c += __popcll(__ballot(holds));
    }
    return c;
}
template <class Entry>
__global__ __launch_bounds__(1024) void todo_count_kernel(const Entry* todo, const unsigned long long* n_in, unsigned long long n_max,
                                                          unsigned long long* counts)
{
    __shared__ unsigned wc[16];
    const unsigned long long n = n_in ? (*n_in < n_max ? *n_in : n_max) : n_max;
    const unsigned           w = threadIdx.x >> 6;
    const unsigned           c = todo_wave_count(todo, n, (unsigned long long)blockIdx.x * 16ull + w);
    if ((threadIdx.x & 63) == 0) wc[w] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int i = 0; i < 16; i++) t += wc[i];
        counts[blockIdx.x] = t;
    }
}
// exclusive scan of counts[0 .. n_blocks) in place by ONE block; total[0] = their sum.  A thread sums a stretch of its own
// (a handful of values: one count per 4 096 slots), the 1 024 sums are scanned in LDS, the thread writes its stretch's
// running sums.
__global__ __launch_bounds__(1024) void todo_scan_kernel(unsigned long long* counts, unsigned long long n_blocks, unsigned long long* total)
{
    __shared__ unsigned long long part[1024];
    const unsigned long long per = (n_blocks + 1023ull) / 1024ull;
    const unsigned long long lo = threadIdx.x * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    unsigned long long       sum = 0ull;
    for (unsigned long long i = lo; i < hi; i++) sum += counts[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                           // Hillis-Steele inclusive scan of the 1 024 sums
        const unsigned long long a = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0ull;
        __syncthreads();
        part[threadIdx.x] += a;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - sum;
    for (unsigned long long i = lo; i < hi; i++) {
        const unsigned long long v = counts[i];
        counts[i] = run;
        run += v;
    }
    if (threadIdx.x == 1023) total[0] = part[1023];
}
template <class Entry>
__global__ __launch_bounds__(1024) void todo_scatter_kernel(const Entry* todo, const unsigned long long* n_in, unsigned long long n_max,
                                                            const unsigned long long* offsets, Entry* dense)
{
    __shared__ unsigned wc[16];
    const unsigned long long n = n_in ? (*n_in < n_max ? *n_in : n_max) : n_max;
    const unsigned           w = threadIdx.x >> 6;
    const int                lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * 16ull + w;
    if ((unsigned long long)blockIdx.x * TODO_BLOCK >= n) return;
    const unsigned c = todo_wave_count(todo, n, wave);
    if (lane == 0) wc[w] = c;
    __syncthreads();
    unsigned long long at = offsets[blockIdx.x];
    for (unsigned i = 0; i < w; i++) at += wc[i];
    for (int k = 0; k < 4; k++) {
        const unsigned long long slot = wave * TODO_STRETCH + (unsigned long long)k * 64 + lane;
        Entry e;
        e.item_steps = ~0ull;
        if (slot < n) e = todo[slot];
        const bool               holds = todo_holds(e);
        const unsigned long long mask = __ballot(holds);
        if (holds) dense[at + __popcll(mask & ((1ull << lane) - 1ull))] = e;
        at += __popcll(mask);
    }
}
// src[0 .. *n_in or n_max) -> dst packed, *total = entries.  counts: scratch of ceil(n_max / 256) + 1 values.
template <class Entry>
static void launch_todo_pack(const Entry* src, const unsigned long long* n_in, size_t n_max, Entry* dst, unsigned long long* counts,
                             unsigned long long* total, hipStream_t stream)
{
    const size_t blocks = (n_max + TODO_BLOCK - 1) / TODO_BLOCK;
    hipLaunchKernelGGL(todo_count_kernel<Entry>, dim3((unsigned)blocks), dim3(1024), 0, stream, src, n_in, (unsigned long long)n_max, counts);
    hipLaunchKernelGGL(todo_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, (unsigned long long)blocks, total);
    hipLaunchKernelGGL(todo_scatter_kernel<Entry>, dim3((unsigned)blocks), dim3(1024), 0, stream, src, n_in, (unsigned long long)n_max,
                       (const unsigned long long*)counts, dst);
}

#ifdef CNF2_X_GUIDED_ROUNDS
// The first rounds of the guided bisection in lock step: one thread per slot of the scout's list.  A flow that was set aside
// lives 3 - 5 rounds in the guided form, so in the persistent kernel above a wavefront takes flows up nearly every round --
// set-up code (loads, the prior's logarithms, the gradient at the start, the replay, the slope bound) that runs for the 16
// or so lanes that are free while the others wait.  Here every lane sets its flow up at once, all lanes ask for their
// first, second, ... point together, and a flow that has not ended after GUIDED_ROUNDS points goes back on the list (its
// steps and decisions so far) for the persistent kernel, which then has little left to do.
#ifndef GUIDED_ROUNDS
#define GUIDED_ROUNDS 4
#endif
template <int KIND, class Entry>
__global__ CNF2_GUIDED_OCC __launch_bounds__(64) void guided_rounds_kernel(UpdateParams u, Entry* todo, const unsigned long long* n_in,
                                                                           unsigned long long n_max, double* flow_out)
{
    const StepControl        sc = {u.scalefactor, u.entropyfactor};
    const unsigned long long n_items = *n_in < n_max ? *n_in : n_max;
    // a wavefront takes 64 entries at a time (the launch has no more blocks than the chip holds several times over: the list's
    // length is known on the device only, and a million blocks that find nothing to do cost a millisecond all the same)
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; (t & ~63ull) < n_items;
         t += (unsigned long long)gridDim.x * blockDim.x) {
    Entry e;
    e.item_steps = ~0ull;
    if (t < n_items) e = todo[t];
    bool               have = todo_holds(e);
    const bool         had = have;
    int                both = PARTNER_NONE;
    unsigned long long item = 0;
    size_t             row_i = 0;
    CertaintyFlow      c;
    HaploFlow          h;
    FlowState          f;
    FlowGuide          g;
    SlopeTerms         st;
    int                hits = 0;
    unsigned           n_points = 0, n_rounds = 0, n_evals = 0, n_why1 = 0;
    auto grad = [&](double x) CNF2_LI { return KIND == 0 ? certainty_rgradient(c, x) : haplo_rgradient(h, x); };
    if (have) {
        double start;
        item = FLOW_ITEM(e);
        if (KIND == 0) {
            certainty_item(u, item, sc, &c, &both);
            st = certainty_slope(c);
            start = c.curprob;
            f.epsilon = c.epsilon;
        } else {
            size_t k;
            int    r;
            haplo_item(u, item, &row_i, &k, &r);
            start = u.hw[row_i];
            haplo_flow_terms(start, u.acc_hb[k], u.acc_hc[k], haplo_similarity_of(e), u.ratio[k], u.children[r], u.descendants[r], sc, &h);
            st = haplo_slope(h);
            f.epsilon = h.epsilon;
        }
        flow_begin(&f, grad, start, f.epsilon, sc.scalefactor, false);
        flow_replay(&f, e.path, (int)(e.item_steps & 63));
        flow_guide_begin(&g);
        flow_guide_try_mono(f, &g, st);
        g.mono_tried = true;
        flow_guide_seed(f, &g, grad, st, sc.scalefactor);
    }
    for (int round = 0; round <= GUIDED_ROUNDS; round++) {
        if (!__ballot(have)) break;
        int    rc = 0;
        double p = 0.0;
        if (have) {
            n_rounds++;
            rc = flow_guide_next(&f, &g, st, sc.scalefactor, &p);
            if (rc == 0) {
                int          hh = 0;
                const double res = flow_end(f, sc.scalefactor, &hh, false);
                hits += hh;
                if (KIND == 0) certainty_store(u, flow_out, item, both, res, &hits, hh);
                else u.hw[row_i] = res;
                have = false;
                n_why1 += f.why == 1;
                e.item_steps = ~0ull;
            }
        }
        if (round == GUIDED_ROUNDS) break;
        if (rc == 2) flow_guide_feed(f, &g, p, flow_point(f, grad, p, sc.scalefactor), sc.scalefactor);
        else if (rc == 3) flow_guide_feed_clear(f, &g, p, flow_pace(grad, p, f.epsilon));
    }
    if (t < n_items && had) {
        if (have) {                                  // not ended: back on the list as it stands
            e.item_steps = (item << 6) | (unsigned long long)f.it;
            e.path = f.path;
        }
        todo[t] = e;
    }
    if (had) {
        n_points = g.points;
        n_evals = g.evals;
    }
    flow_hits(u.hits, hits);
    flow_stats(u.stats ? u.stats + (KIND == 0 ? 16 : 20) : nullptr, n_points, n_rounds, n_evals, n_why1);
    }
}

#endif  // CNF2_X_GUIDED_ROUNDS

// The first of the lock-step kernels: no decisions between the points.  Nine flows in ten take exactly the same course --
// estimate, plan, the plan's three points, the bisection's steps from the facts they leave -- so the course is laid out
// as straight-line code: the set-up, the estimate (the certainties': one literal evaluation where an Euler step lands and
// Newton on its value; the weights': a 5-point rule, their integrands being the smoother ones), ONE plan, its points one
// after the other, the steps.  What an adaptive form (guided_rounds_kernel, an experiment now) spends between its points -- the steps' loop and the plan's,
// each as long as the wavefront's longest lane, every round -- is spent once.  A flow whose gradient is not known to be
// monotone, or that is still going after its steps, stays on the list as it stands for the step-per-round kernels.
template <int KIND, class Entry>
__global__ CNF2_GUIDED_OCC __launch_bounds__(64) void guided_first_kernel(UpdateParams u, Entry* todo, const unsigned long long* n_in,
                                                                          unsigned long long n_max, double* flow_out)
{
    const StepControl        sc = {u.scalefactor, u.entropyfactor};
    const unsigned long long n_items = *n_in < n_max ? *n_in : n_max;
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; (t & ~63ull) < n_items;
         t += (unsigned long long)gridDim.x * blockDim.x) {          // 64 entries at a time (see guided_rounds_kernel)
    Entry e;
    e.item_steps = ~0ull;
    if (t < n_items) e = todo[t];
    const bool         had = todo_holds(e);
    int                both = PARTNER_NONE;
    unsigned long long item = 0;
    size_t             row_i = 0;
    CertaintyFlow      c;
    HaploFlow          h;
    FlowState          f;
    FlowGuide          g;
    SlopeTerms         st;
    int                hits = 0;
    unsigned           n_why1 = 0;
    auto grad = [&](double x) CNF2_LI { return KIND == 0 ? certainty_rgradient(c, x) : haplo_rgradient(h, x); };
    bool go = false;
    if (had) {
        double start;
        item = FLOW_ITEM(e);
        if (KIND == 0) {
            certainty_item(u, item, sc, &c, &both);
            st = certainty_slope(c);
            start = c.curprob;
            f.epsilon = c.epsilon;
        } else {
            size_t k;
            int    r;
            haplo_item(u, item, &row_i, &k, &r);
            start = u.hw[row_i];
            haplo_flow_terms(start, u.acc_hb[k], u.acc_hc[k], haplo_similarity_of(e), u.ratio[k], u.children[r], u.descendants[r], sc, &h);
            st = haplo_slope(h);
            f.epsilon = h.epsilon;
        }
        flow_begin(&f, grad, start, f.epsilon, sc.scalefactor, false);
        flow_replay(&f, e.path, (int)(e.item_steps & 63));
        flow_guide_begin(&g);
        flow_guide_try_mono(f, &g, st);
        g.mono_tried = true;
        flow_guide_seed(f, &g, grad, st, sc.scalefactor, KIND == 1);
        go = g.mono;
    }
    double p = 0.0;
    int    rc = 0;
    // the estimate to plan with: a rule's value (literal: a fact as well) where none is there yet
    if (go && !(g.best_dt < HUGE_VAL)) {
        rc = flow_guide_next(&f, &g, st, sc.scalefactor, &p);
        if (rc == 2) flow_guide_feed(f, &g, p, flow_point(f, grad, p, sc.scalefactor), sc.scalefactor);
        else if (rc == 3) flow_guide_feed_clear(f, &g, p, flow_pace(grad, p, f.epsilon));
        go = rc != 0;
    }
    if (go && f.live && !g.capped) {
        double x[3];
        bool   use[3];
        flow_guide_points(f, &g, sc.scalefactor, x, use);
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (use[k] && flow_guide_known(f, g, x[k]) == PT_NONE) flow_guide_feed(f, &g, x[k], flow_point(f, grad, x[k], sc.scalefactor), sc.scalefactor);
    }
    if (had) {
        rc = flow_guide_next(&f, &g, st, sc.scalefactor, &p, true);
        if (rc == 0) {
            int          hh = 0;
            const double res = flow_end(f, sc.scalefactor, &hh, false);
            hits += hh;
            if (KIND == 0) certainty_store(u, flow_out, item, both, res, &hits, hh);
            else u.hw[row_i] = res;
            n_why1 += f.why == 1;
            e.item_steps = ~0ull;
        } else {                                     // not ended: back on the list as it stands
            e.item_steps = (item << 6) | (unsigned long long)f.it;
            e.path = f.path;
        }
        todo[t] = e;
    }
    flow_hits(u.hits, hits);
#ifdef CNF2_X_RC_HIST
    if (KIND == 0)
#endif
    flow_stats(u.stats ? u.stats + (KIND == 0 ? 16 : 20) : nullptr, had ? g.points : 0u, had ? 5u : 0u, had ? g.evals : 0u, n_why1);
    }
}

static void launch_literal_tail(const UpdateParams& u, unsigned long long* next, const FlowTodo* list, const unsigned long long* n_in, size_t n,
                                unsigned long long want_flags, unsigned grid, hipStream_t stream)
{
    hipLaunchKernelGGL(certainty_finish_kernel, dim3(grid), dim3(64), 0, stream, u, next, list, (unsigned long long)n, u.flow_out, n_in, want_flags);
}
static void launch_literal_tail(const UpdateParams& u, unsigned long long* next, const HaploTodo* list, const unsigned long long* n_in, size_t n,
                                unsigned long long want_flags, unsigned grid, hipStream_t stream)
{
    hipLaunchKernelGGL(haploweight_finish_kernel, dim3(grid), dim3(64), 0, stream, u, next, list, (unsigned long long)n, n_in, want_flags);
}
// The flows the scouts set aside among the n slots of u.todo, the guided way: pack them, the straight-line kernel, and the
// step-per-round kernels for what it leaves.  u.flow_next[28]: the packed list's length (device side: nothing here waits for
// the host).
template <int KIND, class Entry>
static void launch_guided(const UpdateParams& u, size_t n, hipStream_t stream)
{
    const size_t        resident = (size_t)256 * 16;
    const size_t        w = (n + 63) / 64;
    const unsigned      grid = (unsigned)(w < resident ? w : resident);
    const unsigned      batch_grid = (unsigned)(w < resident * 8 ? w : resident * 8);      // the lock-step kernel: 64 entries per wavefront and pass
    Entry*              l0 = (Entry*)u.todo;        // the scouts' list: a slot per flow they looked at
    Entry*              l1 = (Entry*)u.todo2;       // the flows set aside for the guided bisection, packed
    unsigned long long* n1 = u.flow_next + 28;
    double*             out = KIND == 0 ? u.flow_out : nullptr;
    launch_todo_pack<Entry>(l0, nullptr, n, l1, u.todo_counts, n1, stream);
    hipLaunchKernelGGL((guided_first_kernel<KIND, Entry>), dim3(batch_grid), dim3(64), 0, stream, u, l1, (const unsigned long long*)n1,
                       (unsigned long long)n, out);
    const Entry*              left = l1;            // what the first kernel leaves: its list, holes and all
    const unsigned long long* n_left = n1;
#ifdef CNF2_X_GUIDED_ROUNDS
    // experiment (profiles/r05_t_*): pack what is left and give it four adaptive rounds in lock step before the tail
    Entry*              l2 = (Entry*)u.todo3;
    unsigned long long* n2 = u.flow_next + 29;
    launch_todo_pack<Entry>(l1, n1, n, l2, u.todo_counts, n2, stream);
    hipLaunchKernelGGL((guided_rounds_kernel<KIND, Entry>), dim3(batch_grid), dim3(64), 0, stream, u, l2, (const unsigned long long*)n2,
                       (unsigned long long)n, out);
    left = l2;
    n_left = n2;
#endif
    // What is left -- one flow in ten early in a run: a plan that did not come true, a gradient not known to be monotone -- and
    // the flows the scouts set aside for them in the first place (flow_scout: next to their root, where the rule's value
    // saturates) go to the step-per-round kernels with their time bound, from where they stand.
    (void)hipMemsetAsync(u.flow_next, 0, sizeof(unsigned long long), stream);
    launch_literal_tail(u, u.flow_next, left, n_left, n, 0ull, grid, stream);
    (void)hipMemsetAsync(u.flow_next, 0, sizeof(unsigned long long), stream);
    launch_literal_tail(u, u.flow_next, l0, nullptr, n, FLOW_LITERAL, grid, stream);
}

void launch_update_pass(const UpdateParams& u, hipStream_t stream)
{
    const int    len = u.last - u.first + 1;
    const size_t n1 = (size_t)u.n_rec * len;
    const int    n2 = u.n_rec * (u.chrom + 1);
    const size_t n3 = (size_t)u.n_rec * u.chromstarts_host_upto;
    if (!u.flow_next) {                                       // CNF2_UPDATE_PLAIN: one thread per element, the literal steps
        hipLaunchKernelGGL(certainty_update_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, stream, u);
        hipLaunchKernelGGL(phase_ratio_kernel, dim3((n2 + 63) / 64), dim3(64), 0, stream, u);
        hipLaunchKernelGGL(haploweight_update_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, stream, u);
        return;
    }
    const size_t resident = (size_t)256 * 16;                 // wavefronts the chip holds at 4 per SIMD
    // counters: [0] finish kernel's next slot, [1] flows set aside; then 24 statistics.  The scouts run in chunks of
    // todo_cap flows so that the list of flows set aside stays bounded.
    // the statistics add up over the passes of an iteration (chromosome 0's pass starts them afresh)
    if (u.chrom == 0) (void)hipMemsetAsync(u.flow_next + 2, 0, 24 * sizeof(unsigned long long), stream);
    const size_t cap = u.todo_cap;
    // the certainties: every flow (u.mirror == 0), or the mirror pass over the sides (1) and then the pass of the ties (2)
    for (int pass = u.mirror ? 1 : 0; pass <= (u.mirror ? 2 : 0); pass++) {
    UpdateParams up = u;
    up.mirror = pass;
    const UpdateParams& u = up;
    const size_t nc = pass == 1 ? n1 * 2 : n1 * 4;           // positions in the mirror pass, else items
    for (size_t i0 = 0; i0 < nc; i0 += cap) {
        const size_t n = nc - i0 < cap ? nc - i0 : cap;
        (void)hipMemsetAsync(u.flow_next, 0, 2 * sizeof(unsigned long long), stream);
        (void)hipMemsetAsync(u.flow_next + 26, 0, 2 * sizeof(unsigned long long), stream);     // the pass of the ties: anything set aside?
        const size_t w = (n + 63) / 64;
        if (u.scout_passes == 1) {
            hipLaunchKernelGGL(certainty_scout_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, u, (unsigned long long)i0,
                               (unsigned long long)n, u.flow_out, (FlowTodo*)u.todo);
        } else {
            hipLaunchKernelGGL(certainty_scout_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, u, (unsigned long long)i0,
                               (unsigned long long)n, u.flow_out, (FlowTodo*)u.todo);
            hipLaunchKernelGGL(certainty_scout2_kernel, dim3((unsigned)(w < resident ? w : resident)), dim3(64), 0, stream, u, u.flow_next + 1,
                               (FlowTodo*)u.todo, (unsigned long long)i0, (unsigned long long)n, u.flow_out);
        }
        if (u.literal_finish)
            hipLaunchKernelGGL(certainty_finish_kernel, dim3((unsigned)(w < resident ? w : resident)), dim3(64), 0, stream, u, u.flow_next,
                               (const FlowTodo*)u.todo, (unsigned long long)n, u.flow_out, (const unsigned long long*)nullptr, FLOW_LITERAL);
        else
            launch_guided<0, FlowTodo>(u, n, stream);
    }
    }
    hipLaunchKernelGGL(certainty_pick_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, stream, u, u.flow_out);
    hipLaunchKernelGGL(phase_ratio_kernel, dim3((n2 + 63) / 64), dim3(64), 0, stream, u);
    for (size_t i0 = 0; i0 < n3; i0 += cap) {
        const size_t n = n3 - i0 < cap ? n3 - i0 : cap;
        (void)hipMemsetAsync(u.flow_next, 0, 2 * sizeof(unsigned long long), stream);
        // one wavefront per block: a block's slot is free as soon as its own 64 flows are done (-1 % against blocks of 256)
        hipLaunchKernelGGL(haploweight_scout_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, u, (unsigned long long)i0,
                           (unsigned long long)n, (HaploTodo*)u.todo);
        const size_t w = (n + 63) / 64;
        if (u.scout_passes == 2)
            hipLaunchKernelGGL(haploweight_scout2_kernel, dim3((unsigned)(w < resident ? w : resident)), dim3(64), 0, stream, u, u.flow_next + 1,
                               (HaploTodo*)u.todo, (unsigned long long)i0, (unsigned long long)n);
        if (u.literal_finish)
            hipLaunchKernelGGL(haploweight_finish_kernel, dim3((unsigned)(w < resident ? w : resident)), dim3(64), 0, stream, u, u.flow_next,
                               (const HaploTodo*)u.todo, (unsigned long long)n, (const unsigned long long*)nullptr, FLOW_LITERAL);
        else
            launch_guided<1, HaploTodo>(u, n, stream);
    }
}


// Listed rows of a table to or from a packed buffer (the exchanges of a multi-process run: accumulators of the records ranks
// share, rows of the records a rank has updated).  Row i of the copy is src[(src_idx ? src_idx[i] : i) * src_stride ..] ->
// dst[(dst_idx ? dst_idx[i] : i) * dst_stride ..], `elems` elements; blockIdx.x walks a row, blockIdx.y the rows.
template <class T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const T* __restrict__ src, size_t src_stride, const int32_t* __restrict__ src_idx,
                                                        T* __restrict__ dst, size_t dst_stride, const int32_t* __restrict__ dst_idx, int n,
                                                        size_t elems)
{
    for (int i = blockIdx.y; i < n; i += gridDim.y) {
        const T* s = src + (size_t)(src_idx ? src_idx[i] : i) * src_stride;
        T*       d = dst + (size_t)(dst_idx ? dst_idx[i] : i) * dst_stride;
        for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < elems; e += (size_t)gridDim.x * blockDim.x) d[e] = s[e];
    }
}
template <class T>
static void copy_rows_t(const void* src, size_t src_stride, const int32_t* src_idx, void* dst, size_t dst_stride, const int32_t* dst_idx,
                        int n, size_t elems, hipStream_t stream)
{
    if (n <= 0 || elems == 0) return;
    size_t gx = (elems + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(copy_rows_kernel<T>, dim3((unsigned)gx, (unsigned)(n < 65535 ? n : 65535)), dim3(256), 0, stream, (const T*)src,
                       src_stride, src_idx, (T*)dst, dst_stride, dst_idx, n, elems);
}
void launch_copy_rows_f64(const double* src, size_t src_stride, const int32_t* src_idx, double* dst, size_t dst_stride,
                          const int32_t* dst_idx, int n, size_t elems, hipStream_t stream)
{
    copy_rows_t<double>(src, src_stride, src_idx, dst, dst_stride, dst_idx, n, elems, stream);
}
void launch_copy_rows_u8(const uint8_t* src, size_t src_stride, const int32_t* src_idx, uint8_t* dst, size_t dst_stride,
                         const int32_t* dst_idx, int n, size_t elems, hipStream_t stream)
{
    copy_rows_t<uint8_t>(src, src_stride, src_idx, dst, dst_stride, dst_idx, n, elems, stream);
}

// Clock probe: every SIMD of the chip gets 4 wavefronts that each issue `iters` x 8 independent double-precision FMAs and
// nothing else: a SIMD issues one wave-wide f64 FMA per 4 cycles, so the kernel lasts 4 x 8 x iters x 4 cycles and its
// duration gives the shader clock the device actually runs at under a vector-ALU load (boxes differ by several per cent,
// and an issue-bound kernel like the sweep tracks it).
__global__ __launch_bounds__(256) void clock_probe_kernel(int iters, double* sink)
{
    double a0 = threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; i++) {
        a0 = fma(a0, m, c);
        a1 = fma(a1, m, c);
        a2 = fma(a2, m, c);
        a3 = fma(a3, m, c);
        a4 = fma(a4, m, c);
        a5 = fma(a5, m, c);
        a6 = fma(a6, m, c);
        a7 = fma(a7, m, c);
    }
    const double v = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (v == 12345.678) sink[0] = v;         // never true: keeps the chain alive
}
void launch_clock_probe(int n_cu, int iters, double* sink, hipStream_t stream)
{
    // 4 SIMDs x 4 waves = 16 waves = 4 blocks of 256 threads per CU
    hipLaunchKernelGGL(clock_probe_kernel, dim3(n_cu * 4), dim3(256), 0, stream, iters, sink);
}

} // namespace cnf2
