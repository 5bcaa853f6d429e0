// tests/sanitize/sanitize_host.cpp -- the host C++ of the drop-in under sanitizers, on the CPU box (no GPU, no Python):
// the PlantImpute readers with their 4 MB token buffer, the digit former and the text of rows and dumps, the partition
// plan, the update math (literal and guided bisection), and the shared-memory transport of `cnF2freq --gpus N` with its
// ranks run as THREADS of one process (the barrier protocol and the slot traffic are the forked form's; ThreadSanitizer
// sees threads, not processes).  Built three ways by tools/sanitize_host.sh: -fsanitize=address,undefined; -fsanitize=thread;
// plain (the yardstick of the output).  Prints one line per part and "sanitize_host: ok"; any sanitizer report fails the run.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <random>
#include <string>
#include <thread>
#include <vector>

#include "cnf2_update.h"
#include "cnf2_variance.h"
#include "host/cnf2_format.h"
#include "host/cnf2_partition.h"
#include "host/cnf2_readers.h"
#include "host/cnf2_shm_transport.h"
#include "host/cnf2_text.h"

// the two C-ABI calls the transport's device path names; never made here (host buffers)
extern "C" int cnf2_exchange_read(cnf2_ctx*, size_t, void*, size_t) { abort(); }
extern "C" int cnf2_exchange_write(cnf2_ctx*, size_t, const void*, size_t) { abort(); }

using namespace cnf2host;

static int fail(const char* what)
{
    fprintf(stderr, "sanitize_host: %s FAILED\n", what);
    return 1;
}

// ---- readers: files whose tokens straddle the reader's buffer end, CRLF, read counts, a file that ends inside a line
static int part_readers(const std::string& dir)
{
    const int M = 3000, R = 1500;          // ~9 MB of genotype tokens: more than two buffers
    std::mt19937 rng(5);
    {
        FILE* f = fopen((dir + "/s.map").c_str(), "w");
        for (int m = 0; m < M; m++) fprintf(f, "%.6f\r\n", (m % 1000) * 0.1);
        fclose(f);
        f = fopen((dir + "/s.ped").c_str(), "w");
        for (int r = 0; r < R; r++) {
            if (r < 500) fprintf(f, "f%d 0 0 0\n", r);
            else fprintf(f, "k%d f%d f%d %d\r\n", r, (int)(rng() % 500), (int)(rng() % 500), 2 + (int)(rng() % 3));
        }
        fclose(f);
        f = fopen((dir + "/s.gen").c_str(), "w");
        for (int r = 0; r < R; r++) {
            fprintf(f, "%s%d", r < 500 ? "f" : "k", r);
            for (int m = 0; m < M; m++) {
                const unsigned k = rng() % 100;
                if (k < 90) fprintf(f, " %u", k % 3u);
                else if (k < 95) fprintf(f, " 9");
                else fprintf(f, " %u/%u", (unsigned)(rng() % 40), (unsigned)(rng() % 40));
            }
            fprintf(f, r + 1 < R ? "\r\n" : "");      // the last line ends with the file
        }
        fclose(f);
    }
    Pedigree P;
    FILE* f = fopen((dir + "/s.map").c_str(), "r");
    const bool a = read_alpha_map(f, P);
    fclose(f);
    f = fopen((dir + "/s.ped").c_str(), "r");
    const bool b = read_alpha_ped(f, P);
    fclose(f);
    f = fopen((dir + "/s.gen").c_str(), "r");
    const bool c = read_alpha_gen(f, P);
    fclose(f);
    if (!a || !b || !c) return fail("readers");
    if (P.n_markers() != M) return fail("readers: marker count");
    size_t known = 0;
    for (const Individual& I : P.inds)
        for (size_t k = 0; k < I.allele.size(); k++) known += I.allele[k] != 0;
    cap_markers(P, 2500);
    Tables T;
    build_tables(P, T, false);
    printf("readers: %zu individuals, %d markers kept, %zu known alleles, %d rows\n", P.inds.size(), P.n_markers(), known, T.n_rows);
    // the text of rows and dumps of a few individuals, formed side by side by threads (cnf2_text.h)
    std::vector<std::thread> th;
    std::vector<size_t>      bytes(4, 0);
    for (int t = 0; t < 4; t++)
        th.emplace_back([&, t] {
            std::vector<double> dos((size_t)P.n_markers() * 3);
            for (size_t i = 0; i < dos.size(); i++) dos[i] = (double)((i * 2654435761u + t) % 100000) / 99999.0;
            for (int r = t; r < 200; r += 4) {
                TextBuf tb;
                rows_text(P.inds[r].name, 1, dos.data(), 0, P.n_markers(), (r % 17) == 0, tb);
                dump_text(P.inds[r], P.n_markers(), tb);
                bytes[t] += tb.n;
            }
        });
    for (auto& x : th) x.join();
    printf("text: %zu bytes formed by 4 threads\n", bytes[0] + bytes[1] + bytes[2] + bytes[3]);
    return 0;
}

// ---- the digit former against snprintf, the decimal ties and their neighbours first of all
static int part_format()
{
    std::mt19937_64 rng(7);
    size_t bad = 0, n = 0;
    for (int dec : {5, 6}) {
        for (int i = 0; i < 400000; i++) {
            double v;
            if (i < 100000) {
                const double tie = ((double)(rng() % 2000000) + 0.5) / pow(10.0, dec);
                v = nextafter(tie, (i & 1) ? 2.0 : -1.0);
                for (int k = 0; k < (i >> 1) % 4; k++) v = nextafter(v, (i & 1) ? 2.0 : -1.0);
            } else if (i < 200000) v = (double)(rng() >> 11) * (1.0 / 9007199254740992.0);
            else if (i < 300000) v = ldexp((double)(rng() >> 11) * (1.0 / 9007199254740992.0), -(int)(rng() % 60));
            else v = (double)(int64_t)(rng() % 2000001 - 1000000) * 1e-3 * ((i & 1) ? 1.0 : -1.0);
            char a[512], b[512];
            *fmt_fixed(a, v, dec) = 0;
            snprintf(b, sizeof(b), "%.*f", dec, v);
            bad += strcmp(a, b) != 0;
            n++;
        }
    }
    char line[128];
    double hw, ns, s1, s2;
    int    a, b, fields = 0;
    for (int i = 0; i < 100000; i++) {
        snprintf(line, sizeof(line), "%f %d %d %f %f %f", (double)(rng() % 1000) / 999.0, (int)(rng() % 3), (int)(rng() % 3), 0.0,
                 (double)(rng() % 1000) / 1e4, (double)(rng() % 1000) / 1e4);
        fields += parse_dump_line(line, &hw, &a, &b, &ns, &s1, &s2);
    }
    printf("format: %zu values, %zu differ from snprintf; %d dump fields parsed\n", n, bad, fields);
    return bad ? fail("format") : 0;
}

// ---- the partition plan on random window tables
static int part_partition()
{
    std::mt19937 rng(11);
    long shared = 0;
    for (int rep = 0; rep < 200; rep++) {
        const int R = 30 + (int)(rng() % 300), N = 1 + (int)(rng() % 200), M = 7;
        std::vector<int32_t> w((size_t)N * 17, -1);
        for (int j = 0; j < N; j++) {
            w[(size_t)j * 17 + 0] = (int)(rng() % 128);
            w[(size_t)j * 17 + 1] = (int)(rng() % 256);
            for (int k = 0; k < 7; k++) w[(size_t)j * 17 + 2 + k] = (rng() % 5) ? (int)(rng() % R) : -1;
            for (int k = 9; k < 17; k++) w[(size_t)j * 17 + k] = (int)(rng() % 4) - 1;
        }
        for (int world : {1, 2, 3, 8}) {
            std::vector<int> owner(R, -1);
            for (int rank = 0; rank < world; rank++) {
                const Partition Q = plan_partition(R, N, M, w.data(), rank, world);
                for (int r : Q.owned) {
                    if (r < 0 || r >= R || owner[r] != -1) return fail("partition: a record owned twice or out of range");
                    owner[r] = rank;
                }
                shared += (long)Q.n_shared;
            }
        }
    }
    printf("partition: 800 plans, %ld shared records in all\n", shared);
    return 0;
}

// ---- the update math: the guided bisection against the literal one (the decisions must be the same to the bit)
static int part_update()
{
    std::mt19937_64 rng(13);
    auto   uni = [&] { return (double)(rng() >> 11) * (1.0 / 9007199254740992.0); };
    size_t diff = 0, diff2 = 0, passes2 = 0, n = 0;
    for (int i = 0; i < 4000; i++) {
        const int    kind = (int)(rng() & 1);
        const double y = (i % 3 == 0) ? 0.02 + 0.96 * uni() : ((i % 3 == 1) ? pow(10.0, -5.0 + 3.5 * uni()) : 1.0 - pow(10.0, -5.0 + 3.5 * uni()));
        const double h = pow(10.0, -2.0 + 4.0 * uni()), share = fmin(fmax(y * pow(10.0, uni() - 0.5), 1e-9), 1.0 - 1e-9);
        const double eps = 5e-6 / (1 + (int)(rng() % 3)), sf = (i & 4) ? 0.05 : 0.4;
        const double yy = fmin(fmax(y, eps), 1.0 - eps);
        cnf2::SlopeTerms st;
        st.ev = cnf2::evidence_terms(yy, h * share, h);
        st.e = kind ? 0.04 : 1.0;
        st.d = kind ? (double)(1 + rng() % 8) : 0.0;
        st.pr = kind ? uni() : 0.0;
        st.c0 = kind ? 0.0 : 3.0 * (uni() - 0.5);
        cnf2::CertaintyFlow c;
        c.ev = st.ev;
        c.ef = 1.0;
        c.priord = st.c0;
        cnf2::HaploFlow hf;
        hf.ev = st.ev;
        hf.ent = st.e;
        hf.phaseratio = st.pr;
        hf.descendants = st.d;
        auto grad = [&](double x) { return kind == 0 ? cnf2::certainty_rgradient(c, x) : cnf2::haplo_rgradient(hf, x); };
        int  h1 = 0, h2 = 0;
        const double a = cnf2::flow_step(grad, yy, eps, sf, &h1, false);
        const double b = cnf2::flow_step_guided(grad, st, yy, eps, sf, &h2, false);
        diff += (a != b) || (h1 != h2);
        // the device kernels' two scout passes: 8 steps without closing in on the root, the flows still going begun again and
        // scouted to their end (or to the step that needs a quadrature: literal steps from there)
        cnf2::FlowState f;
        cnf2::flow_begin(&f, grad, yy, eps, sf, false);
        int h3 = 0;
        if (!f.pinned) {
            int ev = 0;
            int rs = cnf2::flow_scout(&f, grad, st, sf, &ev, 8, true, false);
            if (rs == 3) {
                const unsigned long long path = f.path;
                const int                it = f.it;
                cnf2::flow_begin(&f, grad, yy, eps, sf, false);
                cnf2::flow_replay(&f, path, it);
                rs = cnf2::flow_scout(&f, grad, st, sf, &ev, 1 << 30, true, true);
                passes2++;
            }
        }
        while (cnf2::flow_advance(&f, grad, sf)) {}
        const double c2 = cnf2::flow_end(f, sf, &h3, false);
        diff2 += (a != c2) || (h1 != h3);
        n++;
    }
    printf("update: %zu flows, %zu differ between the literal and the guided bisection, %zu between the literal and the two scout passes (%zu went into the second)\n",
           n, diff, diff2, passes2);
    return (diff || diff2) ? fail("update") : 0;
}

// ---- addvariance: the closed form against the sums in the reference's order (variance_exact) on random windows
static int part_variance()
{
    std::mt19937_64 rng(29);
    auto   uni = [&] { return (double)(rng() >> 11) * (1.0 / 9007199254740992.0); };
    size_t n = 0, off = 0;
    for (int i = 0; i < 3000; i++) {
        cnf2::Window w;
        memset(&w, 0, sizeof(w));
        cnf2::Slot slot[7];
        for (int k = 0; k < 7; k++) {
            static const int vals[4] = {0, 1, 2, 9};
            slot[k].a0 = vals[rng() % 4];
            slot[k].a1 = vals[rng() % 4];
            slot[k].s0 = (rng() % 4 == 0) ? 0.0 : 0.2 * uni();
            slot[k].s1 = (rng() % 4 == 0) ? 0.0 : 0.2 * uni();
            slot[k].hw = uni();
            w.row[k] = k;
            w.flags[k] = (uint8_t)((rng() % 8 ? cnf2::SLOT_PRESENT : 0) | (rng() % 5 == 0 ? cnf2::SLOT_FOUNDER : 0));
            w.tie[k] = -1;
        }
        w.flags[0] |= cnf2::SLOT_PRESENT;
        w.flag2ignore = (uint8_t)(rng() % 3 == 0 ? rng() & 127 : 0);
        bool         va, vb;
        const double a = cnf2::variance_closed(w, slot, &va), b = cnf2::variance_exact(w, slot, &vb);
        if (va != vb) return fail("variance: validity");
        if (va && fabs(a - b) > 1e-8 * fabs(b) + 1e-18) off++;
        n++;
    }
    printf("variance: %zu windows, %zu beyond 1e-8 between the closed form and the reference's order\n", n, off);
    return off ? fail("variance") : 0;
}

// ---- the transport: one rank's part of the self-test of tests/shim, here with the ranks as threads
static int transport_rank(ShmRegion* R, int rank, int world, int seg_doubles, int seg_bytes)
{
    auto value = [](int r, size_t i) { return (double)((r + 1) * 1000003 % 9973) + 0.25 * (double)(i % 1000) + (double)i * 1e-3; };
    ShmTransport T;
    T.R = R;
    T.rank = rank;
    int bad = 0;
    std::vector<double> d((size_t)world * seg_doubles);
    for (size_t i = 0; i < d.size(); i++) d[i] = value(rank, i);
    T.host_buf = (unsigned char*)d.data();
    bad |= ShmTransport::call(&T, X_SUM_SEGMENTS, d.data(), d.size(), (size_t)seg_doubles) != 0;
    for (int i = 0; i < seg_doubles; i++) {
        double want = 0;
        for (int r = 0; r < world; r++) want += value(r, (size_t)rank * seg_doubles + i);
        bad |= d[(size_t)rank * seg_doubles + i] != want;
    }
    std::vector<unsigned char> b((size_t)world * seg_bytes, 0);
    for (int i = 0; i < seg_bytes; i++) b[(size_t)rank * seg_bytes + i] = (unsigned char)(rank * 37 + i * 7);
    T.host_buf = b.data();
    bad |= ShmTransport::call(&T, X_GATHER_SEGMENTS, b.data(), b.size(), (size_t)seg_bytes) != 0;
    for (int r = 0; r < world; r++)
        for (int i = 0; i < seg_bytes; i++) bad |= b[(size_t)r * seg_bytes + i] != (unsigned char)(r * 37 + i * 7);
    int32_t h[2] = {rank + 1, 5};
    bad |= ShmTransport::call(&T, X_SUM_HITS, h, 2, 2) != 0;
    bad |= h[0] != world * (world + 1) / 2 || h[1] != 5 * world;
    bad |= ShmTransport::call(&T, X_BARRIER, nullptr, 0, 0) != 0;
    std::vector<unsigned char> hb((size_t)seg_bytes * 3 + 5);
    for (size_t i = 0; i < hb.size(); i++) hb[i] = (unsigned char)(rank == 0 ? i * 13 + 1 : 0xEE);
    T.host_buf = nullptr;
    bad |= ShmTransport::call(&T, X_BCAST_HOST, hb.data(), hb.size(), 0) != 0;
    for (size_t i = 0; i < hb.size(); i++) bad |= hb[i] != (unsigned char)(i * 13 + 1);
    return bad;
}
static int part_transport()
{
    int runs = 0;
    for (int world : {2, 3, 4})
        for (int slot : {64, 4096}) {
            ShmRegion* R = shm_region_create(world, (size_t)slot);
            if (!R) return fail("transport: region");
            std::vector<int>         bad(world, 0);
            std::vector<std::thread> th;
            for (int r = 0; r < world; r++) th.emplace_back([&, r] { bad[r] = transport_rank(R, r, world, 1000, 777); });
            for (auto& x : th) x.join();
            for (int r = 0; r < world; r++)
                if (bad[r]) return fail("transport: a rank received wrong data");
            runs++;
        }
    printf("transport: %d runs of 2 - 4 ranks as threads (reduce-scatter, all-gather, hit sums, barrier, broadcast)\n", runs);
    return 0;
}

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    int rc = 0;
    rc |= part_format();
    rc |= part_partition();
    rc |= part_update();
    rc |= part_variance();
    rc |= part_transport();
    rc |= part_readers(dir);
    if (!rc) printf("sanitize_host: ok\n");
    return rc;
}
