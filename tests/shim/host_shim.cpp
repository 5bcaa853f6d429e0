// Host build of the product's emission/window code for CPU unit tests
// (tests/test_host_emission.py).  Compiled with g++ from the same headers the HIP
// kernels include; contains no algorithmic code of its own beyond the lane loop.
#include <string.h>
#include <algorithm>
#include <vector>

#include "cnf2_lane.h"
#include "cnf2_emtab.h"
#include "cnf2_accum.h"
#include "cnf2_acctab.h"
#include "cnf2_accpath.h"

using namespace cnf2;

static HostPedigree make_ped(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
                             const int32_t* row_of)
{
    HostPedigree P;
    P.n_rec = n_rec;
    P.par.assign(par, par + 2 * n_rec);
    P.empty.assign(empty, empty + n_rec);
    P.gen.assign(gen, gen + n_rec);
    P.row_of.assign(row_of, row_of + n_rec);
    derive_founders(P);
    return P;
}

#include "cnf2_update.h"
#include "cnf2_variance.h"
#include "host/cnf2_partition.h"
#include "host/cnf2_shm_transport.h"
#include "host/cnf2_format.h"
#include "host/cnf2_text.h"
#include <sys/wait.h>
#include <unistd.h>

extern "C" {

// The plan of a multi-process run (csrc/host/cnf2_partition.h) for `world` ranks from the window tables the product derives:
// bounds[world + 1]; owner[n_rec] = rank that updates the record (-1: no window touches it); shared[n_rec] = 1 where several
// ranks' windows touch the record; returns the records per segment of the exchange buffer.
int shim_partition(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen, const int32_t* row_of,
                   const int32_t* dous, int n_dous, int n_markers, int world, int32_t* bounds, int32_t* owner, uint8_t* shared)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    std::vector<int32_t> table((size_t)n_dous * 17);
    for (int j = 0; j < n_dous; j++) {
        Window  w;
        int32_t slot_rec[7];
        derive_window(P, dous[j], &w, slot_rec);
        int32_t* o = &table[(size_t)j * 17];
        o[0] = w.shiftignore;
        o[1] = w.flag2ignore;
        o[2] = P.founder[dous[j]];
        for (int i = 0; i < 7; i++) {
            o[3 + i]  = slot_rec[i];
            o[10 + i] = w.tie[i];
        }
    }
    for (int r = 0; r < n_rec; r++) {
        owner[r] = -1;
        shared[r] = 0;
    }
    int seg = 0;
    for (int q = 0; q < world; q++) {
        const cnf2host::Partition Q = cnf2host::plan_partition(n_rec, n_dous, n_markers, table.data(), q, world);
        if (q == 0) {
            for (int k = 0; k <= world; k++) bounds[k] = Q.bounds[k];
            for (int k = 0; k < world; k++)
                for (int r : Q.shared_of[k]) shared[r] = 1;
            seg = (int)Q.seg_shared;
        }
        for (int r : Q.owned) {
            if (owner[r] >= 0) return -1;          // two ranks claim a record
            owner[r] = q;
        }
    }
    return seg;
}

// The shared-memory transport of `cnF2freq --gpus N` (csrc/host/cnf2_shm_transport.h) on HOST buffers: `world` forked ranks run
// the reduce-scatter, the all-gather, the hit-counter sum and a barrier on seeded data with slots of `slot_bytes` (small slots
// force the chunked path) and check what they receive; returns the number of ranks that failed.
// fmt_fixed (csrc/host/cnf2_format.h) against snprintf("%.Nf") on n values: the number of values whose characters differ
// (first_bad: index of the first one, -1 if none) and, in *fallbacks, how many took the snprintf route inside fmt_fixed.
int shim_format_check(const double* v, int n, int decimals, int* first_bad, int* fallbacks)
{
    int bad = 0, fb = 0;
    *first_bad = -1;
    for (int i = 0; i < n; i++) {
        char a[512], b[512];
        char* e = cnf2host::fmt_fixed(a, v[i], decimals);
        *e = 0;
        snprintf(b, sizeof(b), "%.*f", decimals, v[i]);
        if (strcmp(a, b) != 0) {
            if (!bad) *first_bad = i;
            bad++;
        }
        const double x = fabs(v[i]);
        if (!(x < 1e9)) fb++;
        else {
            static const double P10[10] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9};
            const double s = x * P10[decimals], off = fabs(s - floor(s) - 0.5);
            if (!(s < 4.0e15 && off > s * 2.3e-16 + 1e-300)) fb++;
        }
    }
    *fallbacks = fb;
    return bad;
}
// rows_text / dump_text (csrc/host/cnf2_text.h) against the reference's fprintf calls (cnF2freq.cpp:6183-6188, 8157-8192) on
// one made-up individual of M markers: 0 = both texts identical, 1 = rows differ, 2 = dump differs, 3 = both
int shim_text_check(int M, const double* dosage, const double* hw, const unsigned char* allele, const double* sure,
                    const unsigned char* prior_allele, const double* prior_sure, int has_prior, int n, const char* name)
{
    using namespace cnf2host;
    Individual I;
    I.n = n;
    I.name = name;
    I.hw.assign(hw, hw + M);
    I.allele.assign(allele, allele + 2 * M);
    I.sure.assign(sure, sure + 2 * M);
    I.has_prior = has_prior != 0;
    if (has_prior) {
        I.prior_allele.assign(prior_allele, prior_allele + 2 * M);
        I.prior_sure.assign(prior_sure, prior_sure + 2 * M);
    }
    int     bad = 0;
    TextBuf tb;
    char*   mem = nullptr;
    size_t  len = 0;
    for (int skipped = 0; skipped < 2; skipped++) {
        FILE* f = open_memstream(&mem, &len);
        fprintf(f, "%s:%d\n", name, 7);
        if (!skipped)
            for (int m = 1; m < M; m++) fprintf(f, "%.5lf\t%.5lf\t%.5lf\n", dosage[m * 3], dosage[m * 3 + 1], dosage[m * 3 + 2]);
        fprintf(f, "\n");
        fclose(f);
        tb.clear();
        rows_text(I.name, 7, dosage, 1, M, skipped != 0, tb);
        if (tb.n != len || memcmp(tb.s.data(), mem, len) != 0) bad |= 1;
        free(mem);
    }
    FILE* f = open_memstream(&mem, &len);
    fprintf(f, "%d %s\n", I.n, I.name.c_str());
    for (int m = 0; m < M; m++) {
        if (I.has_prior)
            fprintf(f, "%f\t%d\t%d\t\t%f\t%lf %lf %lf\t%d\t%d\t%lf\t%lf\n", I.hw[m], I.allele[m * 2], I.allele[m * 2 + 1], 0.0,
                    I.sure[m * 2], I.sure[m * 2 + 1], 0.5, I.prior_allele[m * 2], I.prior_allele[m * 2 + 1], I.prior_sure[m * 2],
                    I.prior_sure[m * 2 + 1]);
        else
            fprintf(f, "%f\t%d\t%d\t\t%f\t%lf %lf %lf\n", I.hw[m], I.allele[m * 2], I.allele[m * 2 + 1], 0.0, I.sure[m * 2],
                    I.sure[m * 2 + 1], 0.5);
    }
    fclose(f);
    tb.clear();
    dump_text(I, M, tb);
    if (tb.n != len || memcmp(tb.s.data(), mem, len) != 0) bad |= 2;
    free(mem);
    return bad;
}
// parse_dump_line (csrc/host/cnf2_format.h) against sscanf(line, "%lf %d %d %lf %lf %lf") on `n` NUL-separated lines: the
// number of lines on which the return value or any converted field differs (bitwise for the doubles)
int shim_parse_check(const char* lines, int n, int* first_bad, int* fast)
{
    int bad = 0;
    *first_bad = -1;
    *fast = 0;
    const char* p = lines;
    for (int k = 0; k < n; k++) {
        double fa[4] = {-7, -7, -7, -7}, fb[4] = {-7, -7, -7, -7};
        int    ia[2] = {-7, -7}, ib[2] = {-7, -7};
        const int ra = cnf2host::parse_dump_line(p, &fa[0], &ia[0], &ia[1], &fa[1], &fa[2], &fa[3]);
        const int rb = sscanf(p, "%lf %d %d %lf %lf %lf", &fb[0], &ib[0], &ib[1], &fb[1], &fb[2], &fb[3]);
        bool same = ra == rb && memcmp(fa, fb, sizeof(fa)) == 0 && memcmp(ia, ib, sizeof(ia)) == 0;
        if (!same) {
            if (!bad) *first_bad = k;
            bad++;
        }
        const char* q = p;
        double      d;
        if (cnf2host::fast_decimal(q, &d)) (*fast)++;
        p += strlen(p) + 1;
    }
    return bad;
}
int shim_format_int(int v, char* out)
{
    char* e = cnf2host::fmt_int(out, v);
    *e = 0;
    return (int)(e - out);
}

int shim_shm_transport_selftest(int world, int seg_doubles, int seg_bytes, int slot_bytes)
{
    using namespace cnf2host;
    ShmRegion* R = shm_region_create(world, (size_t)slot_bytes);
    if (!R) return -1;
    auto value = [](int rank, size_t i) { return (double)((rank + 1) * 1000003 % 9973) + 0.25 * (double)(i % 1000) + (double)i * 1e-3; };
    std::vector<pid_t> kids;
    for (int rank = 0; rank < world; rank++) {
        const pid_t pid = fork();
        if (pid == 0) {
            ShmTransport T;
            T.R = R;
            T.rank = rank;
            int bad = 0;
            // reduce-scatter of world x seg_doubles doubles
            std::vector<double> d((size_t)world * seg_doubles);
            for (size_t i = 0; i < d.size(); i++) d[i] = value(rank, i);
            T.host_buf = (unsigned char*)d.data();
            bad |= ShmTransport::call(&T, X_SUM_SEGMENTS, d.data(), d.size(), (size_t)seg_doubles) != 0;
            for (int i = 0; i < seg_doubles; i++) {
                double want = 0;
                for (int r = 0; r < world; r++) want += value(r, (size_t)rank * seg_doubles + i);
                bad |= d[(size_t)rank * seg_doubles + i] != want;
            }
            // all-gather of world x seg_bytes bytes
            std::vector<unsigned char> b((size_t)world * seg_bytes, 0);
            for (int i = 0; i < seg_bytes; i++) b[(size_t)rank * seg_bytes + i] = (unsigned char)(rank * 37 + i * 7);
            T.host_buf = b.data();
            bad |= ShmTransport::call(&T, X_GATHER_SEGMENTS, b.data(), b.size(), (size_t)seg_bytes) != 0;
            for (int r = 0; r < world; r++)
                for (int i = 0; i < seg_bytes; i++) bad |= b[(size_t)r * seg_bytes + i] != (unsigned char)(r * 37 + i * 7);
            // hit counters, barrier
            int32_t h[2] = {rank + 1, 5};
            bad |= ShmTransport::call(&T, X_SUM_HITS, h, 2, 2) != 0;
            bad |= h[0] != world * (world + 1) / 2 || h[1] != 5 * world;
            bad |= ShmTransport::call(&T, X_BARRIER, nullptr, 0, 0) != 0;
            // broadcast of rank 0's host bytes
            std::vector<unsigned char> hb((size_t)seg_bytes * 3 + 5);
            for (size_t i = 0; i < hb.size(); i++) hb[i] = (unsigned char)(rank == 0 ? i * 13 + 1 : 0xEE);
            T.host_buf = nullptr;
            bad |= ShmTransport::call(&T, X_BCAST_HOST, hb.data(), hb.size(), 0) != 0;
            for (size_t i = 0; i < hb.size(); i++) bad |= hb[i] != (unsigned char)(i * 13 + 1);
            _exit(bad ? 1 : 0);
        }
        kids.push_back(pid);
    }
    int failed = 0;
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        failed += !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
    }
    return failed;
}

int shim_founders(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
                  const int32_t* row_of, uint8_t* out)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    memcpy(out, P.founder.data(), n_rec);
    return 0;
}

// out17: shiftignore, flag2ignore, founder, 7 slot records, 7 tie groups
int shim_window(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
                const int32_t* row_of, int rec, int32_t* out17)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    Window w;
    int32_t slot_rec[7];
    derive_window(P, rec, &w, slot_rec);
    out17[0] = w.shiftignore;
    out17[1] = w.flag2ignore;
    out17[2] = P.founder[rec];
    for (int i = 0; i < 7; i++) {
        out17[3 + i]  = slot_rec[i];
        out17[10 + i] = w.tie[i];
    }
    return w.n_groups;
}

// All 64 table entries of (rec, marker) for tie combination `combo`:
// tot[64], rtot[64], two[64] in lane order and c[f][s0] -> c4[f*2+s0].
int shim_emtab(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
               const int32_t* row_of, const uint8_t* allele, const double* sure, const double* hw,
               int n_markers, int rec, int marker, int combo, double* tot, double* rtot, double* two,
               double* c4)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    Window w;
    derive_window(P, rec, &w, nullptr);
    auto slot_at = [&](int row) {
        size_t i = (size_t)row * n_markers + marker;
        return unpack_slot((uint8_t)(allele[i * 2] | (allele[i * 2 + 1] << 4)), sure[i * 2], sure[i * 2 + 1], hw[i]);
    };
    Slot root = slot_at(w.row[0]);
    bool root_attop = w.flags[0] & SLOT_FOUNDER;
    for (int lane = 0; lane < 64; lane++) {
        LaneJob L;
        make_lane(w, lane, &L);
        RootTerms R;
        root_terms(root, root_attop, L.f, &R);
        if (root_attop) {
            // root is the top of its single line (cnF2freq.cpp:1260-1271)
            tot[lane] = rtot[lane] = 1.0;
            two[lane] = (L.P == 0 && R.inmv0 == 2) ? 1.0 : 0.0;
            continue;
        }
        LineTerms T;
        line_terms(L.cfg, slot_at(L.row_par), slot_at(L.row_tr), slot_at(L.row_ot),
                   L.P ? R.inmv1 : R.inmv0, L.P ? R.sv1 : R.sv0, L.P == 0 && R.inmv0 == 2, &T);
        tot[lane] = line_total(T);
        line_restricted(L.cfg, T, tie_force(L.tie_par, combo), tie_force(L.tie_tr, combo),
                        tie_force(L.tie_ot, combo), &rtot[lane], &two[lane]);
    }
    for (int f = 0; f < 2; f++) {
        RootTerms R;
        root_terms(root, root_attop, f, &R);
        for (int s0 = 0; s0 < 2; s0++) c4[f * 2 + s0] = R.cbase * phase_weight(root, f ^ s0);
    }
    return w.n_groups;
}

// Same tables through the tile producer of the fast kernel (cnf2_emtab.h), no ties:
// tot/rtot/two[64] in table-index order, c4[f*2+s0] = root weight.
int shim_emtab_fast(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
                    const int32_t* row_of, const uint8_t* allele, const double* sure, const double* hw,
                    int n_markers, int rec, int marker, double* tot, double* rtot, double* two, double* c4)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    Window w;
    derive_window(P, rec, &w, nullptr);
    auto slot_at = [&](int row) {
        size_t i = (size_t)row * n_markers + marker;
        return unpack_slot((uint8_t)(allele[i * 2] | (allele[i * 2 + 1] << 4)), sure[i * 2], sure[i * 2 + 1], hw[i]);
    };
    Slot root = slot_at(w.row[0]);
    for (int part = 0; part < 8; part++) {
        PartCfg c;
        int32_t rp, rt, ro;
        make_part(w, part, &c, &rp, &rt, &ro);
        double t[8], r[8], t2[8], cw[2];
        emtab_part<true>(c, root, slot_at(rp), slot_at(rt), slot_at(ro), t, r, t2, cw);
        for (int e = 0; e < 8; e++) {
            int idx = part_entry_index(part, e);
            tot[idx] = t[e];
            rtot[idx] = r[e];
            two[idx] = t2[e];
        }
        c4[c.f * 2 + 0] = cw[0];
        c4[c.f * 2 + 1] = cw[1];
    }
    return w.n_groups;
}

// Same tables with the single-parent-allele shortcut (HOMPAR) wherever the parent is homozygous with equal
// sure at this marker
// ... with the tie rule for combination `combo` (the tile producer of the tied windows' sweep)
int shim_emtab_fast_ties(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
                         const int32_t* row_of, const uint8_t* allele, const double* sure, const double* hw,
                         int n_markers, int rec, int marker, int combo, double* tot, double* rtot, double* two, double* c4)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    Window w;
    derive_window(P, rec, &w, nullptr);
    auto slot_at = [&](int row) {
        size_t i = (size_t)row * n_markers + marker;
        return unpack_slot((uint8_t)(allele[i * 2] | (allele[i * 2 + 1] << 4)), sure[i * 2], sure[i * 2 + 1], hw[i]);
    };
    Slot root = slot_at(w.row[0]);
    for (int part = 0; part < 8; part++) {
        PartCfg c;
        int32_t rp, rt, ro;
        make_part(w, part, &c, &rp, &rt, &ro);
        part_forces(w, part, combo, &c);
        double t[8], r[8], t2[8], cw[2];
        emtab_part_to<true, false, false, false, true>(c, root, slot_at(rp), slot_at(rt), slot_at(ro),
                      [&](int kind, int e, double v) { (kind == 0 ? t : (kind == 1 ? r : t2))[e] = v; }, cw);
        for (int e = 0; e < 8; e++) {
            int idx = part_entry_index(part, e);
            tot[idx] = t[e];
            rtot[idx] = r[e];
            two[idx] = t2[e];
        }
        c4[c.f * 2 + 0] = cw[0];
        c4[c.f * 2 + 1] = cw[1];
    }
    return w.n_groups;
}

int shim_emtab_hompar(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
                      const int32_t* row_of, const uint8_t* allele, const double* sure, const double* hw,
                      int n_markers, int rec, int marker, double* tot, double* rtot, double* two, double* c4)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    Window w;
    derive_window(P, rec, &w, nullptr);
    auto slot_at = [&](int row) {
        size_t i = (size_t)row * n_markers + marker;
        return unpack_slot((uint8_t)(allele[i * 2] | (allele[i * 2 + 1] << 4)), sure[i * 2], sure[i * 2 + 1], hw[i]);
    };
    Slot root = slot_at(w.row[0]);
    int used = 0;
    for (int part = 0; part < 8; part++) {
        PartCfg c;
        int32_t rp, rt, ro;
        make_part(w, part, &c, &rp, &rt, &ro);
        double t[8], r[8], t2[8], cw[2];
        auto sink = [&](int kind, int e, double v) { (kind == 0 ? t : (kind == 1 ? r : t2))[e] = v; };
        const Slot ps = slot_at(rp), ts = slot_at(rt), os = slot_at(ro);
        auto hom = [](const Slot& d) { return d.a0 == d.a1 && d.s0 == d.s1; };
        if (hom(ps) && (c.tr & SLOT_PRESENT) && (c.ot & SLOT_PRESENT) && hom(ts) && hom(os)) {
            emtab_part_to<true, true, true>(c, root, ps, ts, os, sink, cw);
            used += 1000;
        } else if (hom(ps)) {
            emtab_part_to<true, true, false>(c, root, ps, ts, os, sink, cw);
            used++;
        } else if (w.flag2ignore == 0) {
            emtab_part_to<true, false, false, true>(c, root, ps, ts, os, sink, cw);
            used += 1000000;
        } else emtab_part_to<true, false, false>(c, root, ps, ts, os, sink, cw);
        for (int e = 0; e < 8; e++) {
            int idx = part_entry_index(part, e);
            tot[idx] = t[e];
            rtot[idx] = r[e];
            two[idx] = t2[e];
        }
        c4[c.f * 2 + 0] = cw[0];
        c4[c.f * 2 + 1] = cw[1];
    }
    return used;
}

// Same tables from the 7 per-slot records (slot_table / SlotTable, the tile producer's two-phase form)
int shim_emtab_tables(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
                      const int32_t* row_of, const uint8_t* allele, const double* sure, const double* hw,
                      int n_markers, int rec, int marker, double* tot, double* rtot, double* two, double* c4)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    Window w;
    derive_window(P, rec, &w, nullptr);
    double recs[7 * SLOTTAB_DOUBLES];
    for (int k = 0; k < 7; k++) {
        const int row = w.row[k] < 0 ? 0 : w.row[k];
        size_t i = (size_t)row * n_markers + marker;
        slot_table(unpack_slot((uint8_t)(allele[i * 2] | (allele[i * 2 + 1] << 4)), sure[i * 2], sure[i * 2 + 1], hw[i]),
                   recs + k * SLOTTAB_DOUBLES);
    }
    for (int part = 0; part < 8; part++) {
        PartCfg c;
        int32_t rp, rt, ro;
        make_part(w, part, &c, &rp, &rt, &ro);
        double t[8], r[8], t2[8], cw[2];
        emtab_part_tables<true>(c, recs, [&](int kind, int e, double v) { (kind == 0 ? t : (kind == 1 ? r : t2))[e] = v; }, cw);
        for (int e = 0; e < 8; e++) {
            int idx = part_entry_index(part, e);
            tot[idx] = t[e];
            rtot[idx] = r[e];
            two[idx] = t2[e];
        }
        c4[c.f * 2 + 0] = cw[0];
        c4[c.f * 2 + 1] = cw[1];
    }
    return w.n_groups;
}

// Closed form of the infprobs / homozyg accumulators (cnf2_accum.h) at one marker: wg[8][64] = weight of
// (shift mode, state) or 0 where the reference skips the mode; inf_out[7][2][2], hz_out[2].
int shim_accum_infprobs(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
                        const int32_t* row_of, const uint8_t* allele, const double* sure, const double* hw,
                        int n_markers, int rec, int marker, const double* wg, int no_ties, double* inf_out,
                        double* hz_out)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    Window w;
    derive_window(P, rec, &w, nullptr);
    Slot slot[7];
    for (int k = 0; k < 7; k++) {
        const int row = w.row[k] < 0 ? 0 : w.row[k];
        size_t i = (size_t)row * n_markers + marker;
        slot[k] = unpack_slot((uint8_t)(allele[i * 2] | (allele[i * 2 + 1] << 4)), sure[i * 2], sure[i * 2 + 1], hw[i]);
    }
    for (int k = 0; k < 28; k++) inf_out[k] = 0;
    hz_out[0] = hz_out[1] = 0;
    for (int s = 0; s < 8; s++)
        for (int g = 0; g < 64; g++)
            if (wg[s * 64 + g] != 0.0) accum_infprobs(w, slot, g, s, wg[s * 64 + g], no_ties != 0, inf_out, hz_out);
    return w.n_groups;
}

// Table form of all HOT LOOP 2 accumulators (cnf2_acctab.h) at one marker: inf_out[7][2][2], hz_out[2], hap_out[7][2].
int shim_acc_contract(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
                      const int32_t* row_of, const uint8_t* allele, const double* sure, const double* hw,
                      int n_markers, int rec, int marker, const double* wg, int no_ties, double* inf_out,
                      double* hz_out, double* hap_out)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    Window w;
    derive_window(P, rec, &w, nullptr);
    Slot slot[7];
    for (int k = 0; k < 7; k++) {
        const int row = w.row[k] < 0 ? 0 : w.row[k];
        size_t i = (size_t)row * n_markers + marker;
        slot[k] = unpack_slot((uint8_t)(allele[i * 2] | (allele[i * 2 + 1] << 4)), sure[i * 2], sure[i * 2 + 1], hw[i]);
    }
    for (int k = 0; k < 28; k++) inf_out[k] = 0;
    for (int k = 0; k < 14; k++) hap_out[k] = 0;
    hz_out[0] = hz_out[1] = 0;
    acc_contract_scalar(w, slot, wg, no_ties != 0, inf_out, hz_out, hap_out);
    // doupdatehaplo (cnF2freq.cpp:1224-1239): a slot that is homozygous with equal sure at the marker, or absent, adds nothing
    for (int k = 0; k < 7; k++) {
        const bool upd = (w.flags[k] & SLOT_PRESENT) && !(slot[k].a0 == slot[k].a1 && slot[k].s0 == slot[k].s1);
        if (!upd) hap_out[k * 2] = hap_out[k * 2 + 1] = 0;
    }
    return w.n_groups;
}

// Path form of the same accumulators (cnf2_accpath.h: what the fast accumulate kernel evaluates).  Returns -1 for an
// individual that is the top of its own lines (the kernel leaves those to the table form).
int shim_acc_contract_paths(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
                            const int32_t* row_of, const uint8_t* allele, const double* sure, const double* hw,
                            int n_markers, int rec, int marker, const double* wg, int no_ties, double* inf_out,
                            double* hz_out, double* hap_out)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    Window w;
    derive_window(P, rec, &w, nullptr);
    if (w.flags[0] & SLOT_FOUNDER) return -1;
    Slot slot[7];
    for (int k = 0; k < 7; k++) {
        const int row = w.row[k] < 0 ? 0 : w.row[k];
        size_t i = (size_t)row * n_markers + marker;
        slot[k] = unpack_slot((uint8_t)(allele[i * 2] | (allele[i * 2 + 1] << 4)), sure[i * 2], sure[i * 2 + 1], hw[i]);
    }
    for (int k = 0; k < 28; k++) inf_out[k] = 0;
    for (int k = 0; k < 14; k++) hap_out[k] = 0;
    hz_out[0] = hz_out[1] = 0;
    if (no_ties & 2) acc_contract_tile(w, slot, wg, (no_ties & 1) != 0, inf_out, hz_out, hap_out);
    else acc_contract_paths(w, slot, wg, (no_ties & 1) != 0, inf_out, hz_out, hap_out);
    for (int k = 0; k < 7; k++) {
        const bool upd = (w.flags[k] & SLOT_PRESENT) && !(slot[k].a0 == slot[k].a1 && slot[k].s0 == slot[k].s1);
        if (!upd) hap_out[k * 2] = hap_out[k * 2 + 1] = 0;
    }
    return w.n_groups;
}

// every table entry of one (individual, marker): the factored form against the path-walking form.  out_*[64][AK_COUNT]
int shim_acc_entries(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen, const int32_t* row_of,
                     const uint8_t* allele, const double* sure, const double* hw, int n_markers, int rec, int marker,
                     int combo, double* out_fast, double* out_paths)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    Window w;
    derive_window(P, rec, &w, nullptr);
    Slot slot[7];
    for (int k = 0; k < 7; k++) {
        const int row = w.row[k] < 0 ? 0 : w.row[k];
        size_t i = (size_t)row * n_markers + marker;
        slot[k] = unpack_slot((uint8_t)(allele[i * 2] | (allele[i * 2 + 1] << 4)), sure[i * 2], sure[i * 2 + 1], hw[i]);
    }
    const bool attop = (w.flags[0] & SLOT_FOUNDER) != 0;
    for (int e = 0; e < 64; e++) {
        AccRoot ar;
        acc_root(slot[0], attop, (e >> 4) & 1, &ar);
        acc_entry(w, slot, e, combo, false, ar, out_fast + e * AK_COUNT);
        acc_entry_paths(w, slot, e, combo, false, ar, out_paths + e * AK_COUNT);
    }
    return w.n_groups;
}

// closed form of addvariance (cnf2_variance.h) for record `rec` at `marker`; returns 0 when the entry is left alone
int shim_variance(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen, const int32_t* row_of,
                  const uint8_t* allele, const double* sure, const double* hw, int n_markers, int rec, int marker,
                  double* out)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    Window w;
    derive_window(P, rec, &w, nullptr);
    Slot slot[7];
    for (int k = 0; k < 7; k++) {
        const int row = w.row[k] < 0 ? 0 : w.row[k];
        size_t i = (size_t)row * n_markers + marker;
        slot[k] = unpack_slot((uint8_t)(allele[i * 2] | (allele[i * 2 + 1] << 4)), sure[i * 2], sure[i * 2 + 1], hw[i]);
    }
    bool valid;
    *out = variance_closed(w, slot, &valid);
    return valid ? 1 : 0;
}

// the same entry summed in the reference's order (variance_exact): the reference's bits
int shim_variance_exact(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen, const int32_t* row_of,
                        const uint8_t* allele, const double* sure, const double* hw, int n_markers, int rec, int marker,
                        double* out)
{
    HostPedigree P = make_ped(n_rec, par, empty, gen, row_of);
    Window w;
    derive_window(P, rec, &w, nullptr);
    Slot slot[7];
    for (int k = 0; k < 7; k++) {
        const int row = w.row[k] < 0 ? 0 : w.row[k];
        size_t i = (size_t)row * n_markers + marker;
        slot[k] = unpack_slot((uint8_t)(allele[i * 2] | (allele[i * 2 + 1] << 4)), sure[i * 2], sure[i * 2 + 1], hw[i]);
    }
    bool valid;
    *out = variance_exact(w, slot, &valid);
    return valid ? 1 : 0;
}

// ---- per-iteration updates (cnf2_update.h) ----
double shim_cap_step(double intended, double orig, double epsilon, int* hits, int breakathalf)
{
    return cnf2::cap_step(intended, orig, epsilon, hits, breakathalf != 0);
}
double shim_gauss15_reciprocal_linear(double slope, double icpt, double a, double b)
{
    return cnf2::gauss15([&](double x) { return 1.0 / (slope * x + icpt); }, a, b);
}
double shim_evidence_slope(double y, double g, double h, double x) { return cnf2::evidence_slope(y, g, h, x); }
int shim_update_certainty(const double* inf, int side, int allele, double sure, int has_prior, int prior_allele,
                          double prior_sure, int empty, int children, double scalefactor, double entropyfactor,
                          int* hits, int* new_allele, double* new_sure)
{
    cnf2::SideState s = {allele, sure, has_prior ? prior_allele : 0, prior_sure};
    cnf2::StepControl sc = {scalefactor, entropyfactor};
    return cnf2::update_certainty(inf, s, side, empty != 0, has_prior != 0, children, sc, hits, new_allele, new_sure) ? 1 : 0;
}
void shim_phase_ratio(const double* hw, const double* relhaplo, int first, int end, double* ratio)
{
    std::vector<double> fw((size_t)(end - first) * 2);
    cnf2::phase_ratio(hw, relhaplo, first, end, fw.data(), ratio);
}
// updatehaploweights for one individual: every marker of the chromosomes that have any haplocount
void shim_update_haploweights(int n_chrom, const int* chromstarts, double* hw, double* haplobase, double* haplocount,
                              const int32_t* allele, const double* sure, const double* relhaplo, int children,
                              int descendants, double scalefactor, double entropyfactor, int* hits)
{
    cnf2::StepControl sc = {scalefactor, entropyfactor};
    for (int c = 0; c < n_chrom; c++) {
        const int c0 = chromstarts[c], c1 = chromstarts[c + 1];
        bool any = false;
        for (int k = c0; k < c1; k++) any |= (haplocount[k] != 0);
        if (!any) continue;
        std::vector<double> fw((size_t)(c1 - c0) * 2), ratio(c1 - c0);
        cnf2::phase_ratio(hw, relhaplo, c0, c1, fw.data(), ratio.data());
        for (int j = c0; j < c1; j++) {
            if (!(hw[j] != 0 && hw[j] != 1)) continue;
            hw[j] = cnf2::update_haploweight(hw[j], &haplobase[j], &haplocount[j], allele[j * 2], allele[j * 2 + 1],
                                             sure[j * 2], sure[j * 2 + 1], ratio[j - c0], children, descendants, sc,
                                             false, hits);
        }
    }
}
// Step log of one certainty flow (tuning aid, tools/analyse_flows.py): log[step][4] = mid, kind (0 sign-only, 1 quadrature),
// t / scalefactor (quadrature steps), 1 / gradient at mid; returns the number of steps, *why as FlowState::why, *result the
// new probability.  Absent value: returns -1.
int shim_certainty_flow_trace(const double* inf, int v, int allele, double sure, int prior_allele, double prior_sure,
                              int children, double scalefactor, double* log, int* why, double* result, double* g_orig)
{
    cnf2::SideState s = {allele, sure, prior_allele, prior_sure};
    cnf2::StepControl sc = {scalefactor, 1.0};
    cnf2::CertaintyFlow c;
    if (!cnf2::certainty_flow_setup(inf, v, s, children, sc, &c)) return -1;
    auto grad = [&](double x) { return cnf2::certainty_rgradient(c, x); };
    cnf2::FlowState f;
    cnf2::flow_begin(&f, grad, c.curprob, c.epsilon, scalefactor, false);
    *g_orig = f.g0;
    int n = 0;
    for (;;) {
        const double lo = f.lo, hi = f.hi;
        const int    q0 = f.quads;
        if (!f.live || f.it >= 51 || lo > f.hilim || hi < f.lolim) {
            cnf2::flow_advance(&f, grad, scalefactor);
            break;
        }
        const double mid = (lo + hi) / 2;
        const double gm = cnf2::flow_pace(grad, mid, f.epsilon);
        double       t = 0;
        double a = f.orig, b = mid;
        if (a > b) std::swap(a, b);
        if (isfinite(gm) && ((gm < 0) == f.falling) && b - a >= 1e-10) {
            const double eps = f.epsilon;
            t = cnf2::gauss15([&](double x) { return cnf2::flow_pace(grad, x, eps); }, a, b);
            if (b != mid) t = -t;
        }
        const bool more = cnf2::flow_advance(&f, grad, scalefactor);
        log[n * 4 + 0] = mid;
        log[n * 4 + 1] = f.quads - q0;
        log[n * 4 + 2] = t / scalefactor;
        log[n * 4 + 3] = gm;
        n++;
        if (!more) break;
    }
    int hits = 0;
    *why = f.why;
    *result = cnf2::flow_end(f, scalefactor, &hits, false);
    return n;
}
// One flow with or without the time bound (flow_time_bound): kind 0 = certainty (a, b from the evidence g of h at belief y;
// e = entropy factor, c0 = prior term), kind 1 = haplotype weight (e = entropy coefficient, d = descendants, pr = phase
// ratio).  out[4] = result, quadrature steps, spared steps, hits.
void shim_flow(int kind, double y, double g, double h, double e, double c0, double d, double pr, double epsilon,
               double scalefactor, int screened, double* out)
{
    cnf2::SlopeTerms st;
    st.ev = cnf2::evidence_terms(y, g, h);
    st.e = e;
    st.d = d;
    st.pr = pr;
    st.c0 = kind == 0 ? e * c0 : 0.0;
    cnf2::CertaintyFlow c;
    c.ev = st.ev;
    c.ef = e;
    c.priord = c0;
    cnf2::HaploFlow hf;
    hf.ev = st.ev;
    hf.ent = e;
    hf.phaseratio = pr;
    hf.descendants = d;
    auto grad = [&](double x) { return kind == 0 ? cnf2::certainty_rgradient(c, x) : cnf2::haplo_rgradient(hf, x); };
    auto bound = [&](double xa, double xb, double pc, double lim) { return screened ? cnf2::flow_time_under(st, xa, xb, pc, lim) : false; };
    cnf2::FlowState f;
    cnf2::flow_begin(&f, grad, y, epsilon, scalefactor, false);
    if (screened == 3 && !f.pinned) {
        // what the device does: the scout, and where it stops at a quadrature a second pass that begins the flow anew,
        // replays the scout's decisions and goes on literally (with the bound)
        int evals = 0;
        const int rc = cnf2::flow_scout(&f, grad, st, scalefactor, &evals);
        out[5] = evals;
        out[6] = rc;
        if (rc == 2 || rc == 5 || rc == 6) {
            const unsigned long long path = f.path;
            const int                steps = f.it, spared = f.spared;
            cnf2::FlowState g2;
            cnf2::flow_begin(&g2, grad, y, epsilon, scalefactor, false);
            cnf2::flow_replay(&g2, path, steps);
            auto b1 = [&](double xa, double xb, double pc, double lim) { return cnf2::flow_time_under(st, xa, xb, pc, lim); };
            while (cnf2::flow_advance(&g2, grad, scalefactor, b1)) {}
            g2.spared += spared;
            f = g2;
        }
    } else if ((screened == 4 || screened == 5) && !f.pinned) {
        // the guided bisection (4: from the start; 5: after the scout, as the device runs it): out[5] = gradient evaluations, out[6] = literal points
        int evals = 0, rc = 2;
        if (screened == 5) rc = cnf2::flow_scout(&f, grad, st, scalefactor, &evals);
        cnf2::FlowGuide g;
        cnf2::flow_guide_begin(&g);
        if (rc != 0) {
            const unsigned long long path = f.path;
            const int                steps = f.it, spared = f.spared;
            cnf2::flow_begin(&f, grad, y, epsilon, scalefactor, false);
            cnf2::flow_replay(&f, path, steps);
            f.spared = spared;
            cnf2::flow_guide_try_mono(f, &g, st);
            g.mono_tried = true;
            cnf2::flow_guide_seed(f, &g, grad, st, scalefactor);
            double p;
            for (int rcn; (rcn = cnf2::flow_guide_next(&f, &g, st, scalefactor, &p)) != 0;) {
                if (rcn == 3) cnf2::flow_guide_feed_clear(f, &g, p, cnf2::flow_pace(grad, p, f.epsilon));
                else cnf2::flow_guide_feed(f, &g, p, cnf2::flow_point(f, grad, p, scalefactor), scalefactor);
            }
        }
        out[5] = evals + g.evals;
        out[6] = g.points;
    } else if (screened == 2 && !f.pinned) {
        // the one-evaluation-at-a-time machine the kernels run
        cnf2::FlowRun r;
        r.f = f;
        r.phase = 0;
        double v;
        auto   b1 = [&](double xa, double xb, double pc, double lim) { return cnf2::flow_time_under(st, xa, xb, pc, lim); };
        while (cnf2::flow_want(&r, scalefactor, &v)) cnf2::flow_feed(&r, grad(v), scalefactor, b1);
        f = r.f;
    } else {
        while (cnf2::flow_advance(&f, grad, scalefactor, bound)) {}
    }
    int hits = 0;
    out[0] = cnf2::flow_end(f, scalefactor, &hits, false);
    out[1] = f.quads;
    out[2] = f.spared;
    out[3] = hits;
    out[4] = f.it;
}
// flow_time_bound against what the rule really reports over [xa, xb] (xa or xb = the flow's start, sign of G constant):
// out[0] = bound, out[1] = |rule|, out[2] = min over 200 sample points of -G' by central differences, out[3] = the
// slope bound s1 the time bound used (recovered from the bound: 0 when it is infinite)
void shim_time_bound(int kind, double y, double g, double h, double e, double c0, double d, double pr, double xa, double xb,
                     int mid_is_b, double* out)
{
    cnf2::SlopeTerms st;
    st.ev = cnf2::evidence_terms(y, g, h);
    st.e = e;
    st.d = d;
    st.pr = pr;
    st.c0 = kind == 0 ? e * c0 : 0.0;
    cnf2::CertaintyFlow c;
    c.ev = st.ev;
    c.ef = e;
    c.priord = c0;
    cnf2::HaploFlow hf;
    hf.ev = st.ev;
    hf.ent = e;
    hf.phaseratio = pr;
    hf.descendants = d;
    auto rg = [&](double x) { return kind == 0 ? cnf2::certainty_rgradient(c, x) : cnf2::haplo_rgradient(hf, x); };
    const double xm = mid_is_b ? xb : xa;
    out[0] = cnf2::flow_time_bound(st, xa, xb, fabs(rg(xm)));
    out[1] = fabs(cnf2::gauss15(rg, xa, xb));
    double smin = HUGE_VAL;
    bool   sign_const = true;
    const double g_first = 1.0 / rg(xa);
    for (int i = 0; i <= 200; i++) {
        const double x = xa + (xb - xa) * i / 200.0, hstep = 1e-6 * (x < 1 - x ? x : 1 - x);
        const double gp = 1.0 / rg(x + hstep), gm = 1.0 / rg(x - hstep);
        smin = std::min(smin, -(gp - gm) / (2 * hstep));
        sign_const = sign_const && ((1.0 / rg(x) < 0) == (g_first < 0));
    }
    out[2] = smin;
    out[3] = sign_const ? 1.0 : 0.0;
    out[4] = ((1.0 / rg(xa) < 0) == (1.0 / rg(xb) < 0)) ? 1.0 : 0.0;     // the caller's precondition: same sign at both ends
    out[5] = 1.0 / rg(mid_is_b ? xa : xb);                                 // gradient at the flow's start (its sign is the direction)
}
// Many certainty flows three ways (tuning aid and the CPU parity test of the guided bisection): per flow inputs as in
// shim_certainty_flow_trace; out[n][8] = literal result, literal hits, literal gradient evaluations (steps + 15 per quadrature),
// guided result, guided hits, gradient evaluations of the scout, of the guided part, literal points of the guided part;
// scout_first: what the device does (the scout, then the guided bisection from where the scout stopped); else guided from the start.
// Absent values (no evidence) get NaN results.
void shim_certainty_flows_guided(int n, const double* inf, const int32_t* v, const int32_t* allele, const double* sure,
                                 const int32_t* prior_allele, const double* prior_sure, const int32_t* children, double scalefactor,
                                 int scout_first, double* out)
{
    cnf2::StepControl sc = {scalefactor, 1.0};
    for (int i = 0; i < n; i++) {
        double* o = out + (size_t)i * 8;
        for (int k = 0; k < 8; k++) o[k] = 0;
        cnf2::SideState s = {allele[i], sure[i], prior_allele[i], prior_sure[i]};
        cnf2::CertaintyFlow c;
        if (!cnf2::certainty_flow_setup(inf + (size_t)i * 2, v[i], s, children[i], sc, &c)) {
            o[0] = o[3] = NAN;
            continue;
        }
        auto grad = [&](double x) { return cnf2::certainty_rgradient(c, x); };
        const cnf2::SlopeTerms st = cnf2::certainty_slope(c);
        {
            cnf2::FlowState f;
            cnf2::flow_begin(&f, grad, c.curprob, c.epsilon, scalefactor, false);
            while (cnf2::flow_advance(&f, grad, scalefactor)) {}
            int hits = 0;
            o[0] = cnf2::flow_end(f, scalefactor, &hits, false);
            o[1] = hits;
            o[2] = f.pinned ? 1 : f.it + 15 * f.quads;
        }
        cnf2::FlowState f;
        cnf2::FlowGuide g;
        cnf2::flow_begin(&f, grad, c.curprob, c.epsilon, scalefactor, false);
        cnf2::flow_guide_begin(&g);
        int evals = 0;
        if (f.pinned) {
            while (cnf2::flow_advance(&f, grad, scalefactor)) {}
        } else {
            int rc = 2;
            if (scout_first) rc = cnf2::flow_scout(&f, grad, st, scalefactor, &evals);
            if (rc != 0) {
                double p;
                cnf2::flow_guide_try_mono(f, &g, st);
                g.mono_tried = true;
                cnf2::flow_guide_seed(f, &g, grad, st, scalefactor);
                const bool trace = (getenv("SHIM_TRACE") && i < atoi(getenv("SHIM_TRACE"))) || (getenv("SHIM_TRACE_ONE") && i == atoi(getenv("SHIM_TRACE_ONE")));
                if (trace) printf("flow %d orig %.9g eps %.3g lo %.9g hi %.9g falling %d G0 %.4g\n", i, f.orig, f.epsilon, f.lo, f.hi, (int)f.falling, 1 / f.g0);
                for (int rcn; g.points < 320 && (rcn = cnf2::flow_guide_next(&f, &g, st, scalefactor, &p)) != 0;) {
                    if (rcn == 3) {
                        cnf2::flow_guide_feed_clear(f, &g, p, cnf2::flow_pace(grad, p, f.epsilon));
                        continue;
                    }
                    const cnf2::FlowPoint r = cnf2::flow_point(f, grad, p, scalefactor);
                    cnf2::flow_guide_feed(f, &g, p, r, scalefactor);
                    if (g.points == 300) printf("RUNAWAY flow %d\n", i);
                    if (trace || (g.points > 300 && g.points < 310)) printf("   it %2d mono %d point d %.9g (mid d %.9g) kind %d t/sf %.6f G %.4g  near %.9g far %.9g dstar %.9g\n", f.it, (int)g.mono,
                                      cnf2::flow_distance(f, p), cnf2::flow_distance(f, (f.lo + f.hi) / 2), r.kind, r.t / scalefactor, 1 / r.pace, g.near_d, g.far_d, g.anchor_d);
                }
                if (trace) printf("   ended it %d why %d\n", f.it, f.why);
            }
        }
        int hits = 0;
        o[3] = cnf2::flow_end(f, scalefactor, &hits, false);
        o[4] = hits;
        o[5] = evals;
        o[6] = g.evals;
        o[7] = g.points;
    }
}
// The same for haplotype-weight flows (tuning aid): per flow hw, haplobase, haplocount (as the accumulators hold them), the two
// alleles and their certainties, phase ratio, children, descendants.  out[n][10] = literal result, hits, gradient evaluations;
// guided result, hits, scout evaluations, guided evaluations, guided points; what the scout returned; how the bisection ended (why).
void shim_haplo_flows_guided(int n, const double* hw, const double* hb, const double* hc, const int32_t* a0, const int32_t* a1,
                             const double* s0, const double* s1, const double* ratio, const int32_t* children, const int32_t* desc,
                             double scalefactor, int scout_first, int hand_over, double* out)
{
    cnf2::StepControl sc = {scalefactor, 1.0};
    for (int i = 0; i < n; i++) {
        double* o = out + (size_t)i * 10;
        for (int k = 0; k < 10; k++) o[k] = 0;
        double b = hb[i], c = hc[i];
        cnf2::HaploFlow h;
        cnf2::haplo_flow_setup(hw[i], &b, &c, a0[i], a1[i], s0[i], s1[i], ratio[i], children[i], desc[i], sc, &h);
        auto grad = [&](double x) { return cnf2::haplo_rgradient(h, x); };
        const cnf2::SlopeTerms st = cnf2::haplo_slope(h);
        {
            cnf2::FlowState f;
            cnf2::flow_begin(&f, grad, hw[i], h.epsilon, scalefactor, false);
            while (cnf2::flow_advance(&f, grad, scalefactor)) {}
            int hits = 0;
            o[0] = cnf2::flow_end(f, scalefactor, &hits, false);
            o[1] = hits;
            o[2] = f.pinned ? 1 : f.it + 15 * f.quads;
            o[9] = f.why + 10 * f.it;
        }
        cnf2::FlowState f;
        cnf2::FlowGuide g;
        cnf2::flow_begin(&f, grad, hw[i], h.epsilon, scalefactor, false);
        cnf2::flow_guide_begin(&g);
        int evals = 0, rc = 2;
        if (f.pinned) {
            while (cnf2::flow_advance(&f, grad, scalefactor)) {}
            rc = -1;
        } else {
            if (scout_first) rc = cnf2::flow_scout(&f, grad, st, scalefactor, &evals, 1 << 30, hand_over != 0);
            if (rc != 0) {
                double p;
                cnf2::flow_guide_try_mono(f, &g, st);
                g.mono_tried = true;
                cnf2::flow_guide_seed(f, &g, grad, st, scalefactor);
                const bool trace = getenv("SHIM_TRACE_ONE") && i == atoi(getenv("SHIM_TRACE_ONE"));
                if (trace) printf("flow %d orig %.9g eps %.3g lo %.9g hi %.9g falling %d G0 %.4g mono %d capped %d it %d\n", i, f.orig, f.epsilon, f.lo, f.hi, (int)f.falling, 1 / f.g0, (int)g.mono, (int)g.capped, f.it);
                for (int rcn; g.points < 320 && (rcn = cnf2::flow_guide_next(&f, &g, st, scalefactor, &p)) != 0;) {
                    if (rcn == 3) {
                        cnf2::flow_guide_feed_clear(f, &g, p, cnf2::flow_pace(grad, p, f.epsilon));
                        continue;
                    }
                    const cnf2::FlowPoint r = cnf2::flow_point(f, grad, p, scalefactor);
                    cnf2::flow_guide_feed(f, &g, p, r, scalefactor);
                    if (trace) printf("   it %2d mono %d point d %.9g (mid d %.9g) kind %d t/sf %.6f G %.4g  near %.9g far %.9g\n", f.it, (int)g.mono,
                                      cnf2::flow_distance(f, p), cnf2::flow_distance(f, (f.lo + f.hi) / 2), r.kind, r.t / scalefactor, 1 / r.pace, g.near_d, g.far_d);
                }
                if (trace) printf("   ended it %d why %d\n", f.it, f.why);
            }
        }
        int hits = 0;
        o[3] = cnf2::flow_end(f, scalefactor, &hits, false);
        o[4] = hits;
        o[5] = evals;
        o[6] = g.evals;
        o[7] = g.points;
        o[8] = rc;
    }
}
double shim_adapt_scalefactor(double scalefactor, int hits, int* old, int n_analysed)
{
    cnf2::StepControl sc = {scalefactor, 1.0};
    cnf2::StepHistory h;
    h.oldhits = old[0];
    h.oldhits2 = old[1];
    cnf2::adapt_scalefactor(&sc, &h, hits, n_analysed);
    old[0] = h.oldhits;
    old[1] = h.oldhits2;
    return sc.scalefactor;
}
}
