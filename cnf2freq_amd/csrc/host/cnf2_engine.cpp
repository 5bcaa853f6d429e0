// cnf2_engine.cpp -- see cnf2_engine.h.
#include "cnf2_engine.h"

#include <memory>
#include "cnf2_text.h"
#include <chrono>

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>

#include <sched.h>

namespace cnf2host {

// Threads the host loops may use: the CPUs this process may really run on -- the affinity mask AND the cgroup's CPU quota (a
// container with 16 CPUs' worth of quota on a 200-thread host must not start 200 threads: they would be throttled to a
// crawl, and so would everything after them).  OMP_NUM_THREADS, when set, is respected as an upper bound.
static int g_host_threads = 0;
void set_host_threads(int n) { g_host_threads = n > 0 ? n : 0; }
int host_threads()
{
    int& n = g_host_threads;
    if (n > 0) return n;
    int k = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) k = CPU_COUNT(&set);
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                   // cgroup v2: "<quota> <period>" or "max <period>"
        char   q[64];
        double period = 0;
        if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
            k = std::min(k, std::max(1, (int)(atof(q) / period + 0.5)));
        fclose(f);
    } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {      // cgroup v1
        double quota = -1, period = 0;
        if (fscanf(g, "%lf", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(h, "%lf", &period) != 1) period = 0;
            fclose(h);
        }
        if (quota > 0 && period > 0) k = std::min(k, std::max(1, (int)(quota / period + 0.5)));
    }
    if (const char* e = getenv("OMP_NUM_THREADS")) {
        const int v = atoi(e);
        if (v > 0) k = std::min(k, v);
    }
    n = std::max(1, std::min(k, 64));
    return n;
}

Engine::Engine(Pedigree& ped, cnf2_ctx* c, const EngineOptions& o) : P(ped), ctx(c), opt(o) {}

void Engine::check(int rc, const char* what)
{
    if (rc != CNF2_OK) throw EngineError(rc, std::string(what) + " failed: " + cnf2_last_error(ctx));
}

void Engine::upload()
{
    M = P.n_markers();
    C = (int)P.chromstarts.size() - 1;
    // one row per individual (row 0 stays the blank row): the updates write rows in place, and postmarkerdata locks a
    // haplotype weight even in individuals without data, so rows cannot be shared the way a single sweep allows
    // (the pedigree tables only; the rows go up in slabs straight from the individuals -- no second copy of every row on the
    // host, which at config 4 would be half a terabyte)
    const int R0 = (int)P.inds.size();
    T = Tables();
    T.par.assign((size_t)R0 * 2, -1);
    T.gen.assign(R0, 0);
    T.empty.assign(R0, 0);
    T.row_of.assign(R0, 0);
    T.dous.assign(P.dous.begin(), P.dous.end());
    for (int r = 0; r < R0; r++) {
        T.par[(size_t)r * 2]     = P.inds[r].pars[0];
        T.par[(size_t)r * 2 + 1] = P.inds[r].pars[1];
        T.gen[r]                 = P.inds[r].gen;
        T.empty[r]               = P.inds[r].empty ? 1 : 0;
        T.row_of[r]              = r + 1;
    }
    T.n_rows = R0 + 1;
    N = (int)T.dous.size();
    check(cnf2_upload_map(ctx, P.pos.data(), M, P.chromstarts.data(), C, nullptr), "cnf2_upload_map");
    check(cnf2_upload_rows(ctx, T.n_rows, nullptr, nullptr, nullptr), "cnf2_upload_rows");       // blank rows (row 0 stays blank)
    push_rows();
    check(cnf2_upload_pedigree(ctx, (int)P.inds.size(), T.par.data(), T.empty.data(), T.gen.data(), T.row_of.data(),
                               T.dous.data(), N),
          "cnf2_upload_pedigree");
    std::vector<uint8_t> has_prior(P.inds.size());
    for (size_t r = 0; r < P.inds.size(); r++) has_prior[r] = P.inds[r].has_prior ? 1 : 0;
    check(cnf2_snapshot_priors(ctx, has_prior.data()), "cnf2_snapshot_priors");
    descendants_.assign(P.inds.size(), 0);
    children_.assign(P.inds.size(), 0);
    variances_.assign(P.inds.size() * (size_t)M, 0.0);
    lockstart_.assign(P.inds.size() * (size_t)(C > 0 ? C : 1), 0);
}

// the device buffers of the iterations now, not inside the first one (the batch buffer of the accumulate sweep is up to half of
// the free memory; its first hipMalloc takes seconds).  Optional: an iteration allocates what it finds missing.
void Engine::reserve()
{
    const int b0 = block_end_ < 0 ? 0 : block_begin_, b1 = block_end_ < 0 ? N : block_end_;
    if (opt.update && b1 > b0) check(cnf2_reserve_accumulate(ctx, b0, b1, deterministic_ ? CNF2_DETERMINISTIC : 0), "cnf2_reserve_accumulate");
}

// rows travel in slabs of records: the staging copy stays bounded (config 4: 300 000 records x 200 080 markers would be
// half a terabyte in one piece)
static const size_t ROW_SLAB_BYTES = (size_t)256 << 20;

void Engine::push_rows()
{
    const int R = (int)P.inds.size();
    if (R == 0) return;
    const int slab = (int)std::max<size_t>(1, ROW_SLAB_BYTES / ((size_t)M * 26));
    std::vector<uint8_t> allele((size_t)std::min(R, slab) * M * 2);
    std::vector<double>  sure((size_t)std::min(R, slab) * M * 2), hw((size_t)std::min(R, slab) * M);
    for (int r0 = 0; r0 < R; r0 += slab) {
        const int k = std::min(slab, R - r0);
#pragma omp parallel for schedule(static) num_threads(host_threads())
        for (int r = 0; r < k; r++) {
            const Individual& I = P.inds[r0 + r];
            std::copy(I.allele.begin(), I.allele.end(), allele.begin() + (size_t)r * M * 2);
            std::copy(I.sure.begin(), I.sure.end(), sure.begin() + (size_t)r * M * 2);
            std::copy(I.hw.begin(), I.hw.end(), hw.begin() + (size_t)r * M);
        }
        check(cnf2_update_rows(ctx, 1 + r0, k, allele.data(), sure.data(), hw.data()), "cnf2_update_rows");
    }
    rows_stale_ = false;
}

void Engine::pull_rows()
{
    const int R = (int)P.inds.size();
    if (R == 0) return;
    const int slab = (int)std::max<size_t>(1, ROW_SLAB_BYTES / ((size_t)M * 26));
    std::vector<uint8_t> allele((size_t)std::min(R, slab) * M * 2);
    std::vector<double>  sure((size_t)std::min(R, slab) * M * 2), hw((size_t)std::min(R, slab) * M);
    for (int r0 = 0; r0 < R; r0 += slab) {
        const int k = std::min(slab, R - r0);
        check(cnf2_download_rows(ctx, 1 + r0, k, allele.data(), sure.data(), hw.data()), "cnf2_download_rows");
#pragma omp parallel for schedule(static) num_threads(host_threads())
        for (int r = 0; r < k; r++) {
            Individual& I = P.inds[r0 + r];
            std::copy(allele.begin() + (size_t)r * M * 2, allele.begin() + (size_t)(r + 1) * M * 2, I.allele.begin());
            std::copy(sure.begin() + (size_t)r * M * 2, sure.begin() + (size_t)(r + 1) * M * 2, I.sure.begin());
            std::copy(hw.begin() + (size_t)r * M, hw.begin() + (size_t)(r + 1) * M, I.hw.begin());
        }
    }
    rows_stale_ = false;
}

void Engine::set_block(int begin, int end)
{
    if (begin < 0 || end > N || begin > end) throw EngineError(CNF2_ERR_ARG, "block of analysed individuals out of range");
    block_begin_ = begin;
    block_end_ = end;
}

std::vector<double> Engine::work_costs()
{
    std::vector<double>  cost(N, 0.0);
    std::vector<int32_t> w((size_t)N * 17);
    if (N > 0) check(cnf2_window_table(ctx, w.data()), "cnf2_window_table");
    for (int j = 0; j < N; j++) {
        int groups = 0;
        for (int k = 0; k < 7; k++) groups = std::max(groups, w[(size_t)j * 17 + 10 + k] + 1);
        cost[j] = (double)M * (1.0 + (double)(1 << groups));
    }
    return cost;
}

Partition Engine::plan(int rank, int world)
{
    if (world < 1 || rank < 0 || rank >= world) throw EngineError(CNF2_ERR_ARG, "rank out of range");
    std::vector<int32_t> w((size_t)N * 17);
    if (N > 0) check(cnf2_window_table(ctx, w.data()), "cnf2_window_table");
    return plan_partition((int)P.inds.size(), N, M, w.data(), rank, world);
}

void Engine::balanced_block(int rank, int world, int* begin, int* end)
{
    const Partition Q = plan(rank, world);
    *begin = Q.bounds[rank];
    *end = Q.bounds[rank + 1];
}

void Engine::set_partition(int rank, int world, ExchangeFn fn, void* user)
{
    if (world > 1 && !fn) throw EngineError(CNF2_ERR_ARG, "a partition over several ranks needs a transport");
    part_ = plan(rank, world);
    exchange_ = fn;
    exchange_user_ = user;
    set_block(part_.bounds[rank], part_.bounds[rank + 1]);
}

void Engine::exchange_bytes(size_t out[4]) const
{
    out[0] = out[1] = out[2] = out[3] = 0;
    if (part_.world <= 1) return;
    const size_t S = cnf2_packed_accumulator_doubles(ctx) * sizeof(double), B = cnf2_packed_row_bytes(ctx);
    out[0] = (size_t)part_.world * part_.seg_shared * S;
    out[1] = (size_t)part_.world * part_.seg_shared * B;
    out[2] = (size_t)C * sizeof(int32_t);
    out[3] = part_.n_shared * (S + B) + out[2];
}

void Engine::exchange(int op, void* buf, size_t count, size_t seg, const char* what)
{
    const int rc = exchange_(exchange_user_, op, buf, count, seg);
    if (rc != 0) throw EngineError(CNF2_ERR_STATE, std::string("the transport failed in ") + what + " (" + std::to_string(rc) + ")");
}

// every rank's newer rows of its private records to every rank: one all-gather of packed rows, in slices so that the
// staging buffer stays bounded (a full config-5 state is 6 GB)
void Engine::gather_private_rows()
{
    if (part_.world <= 1 || !rows_partial_) return;
    const size_t B = cnf2_packed_row_bytes(ctx);
    const size_t slice = std::max<size_t>(1, ((size_t)1 << 30) / ((size_t)part_.world * B));
    for (size_t i0 = 0; i0 < part_.seg_private; i0 += slice) {
        const size_t k = std::min(slice, part_.seg_private - i0);
        void* buf = nullptr;
        check(cnf2_exchange_buffer(ctx, (size_t)part_.world * k * B, &buf), "cnf2_exchange_buffer");
        auto part_of = [&](int q, const int32_t** recs) {
            const std::vector<int32_t>& v = part_.private_of[q];
            if (i0 >= v.size()) return 0;
            *recs = v.data() + i0;
            return (int)std::min(k, v.size() - i0);
        };
        const int32_t* mine = nullptr;
        const int      nm = part_of(part_.rank, &mine);
        check(cnf2_pack_rows(ctx, mine, nm, (uint8_t*)buf + (size_t)part_.rank * k * B), "cnf2_pack_rows");
        exchange(X_GATHER_SEGMENTS, buf, (size_t)part_.world * k * B, k * B, "the gather of the rows");
        for (int q = 0; q < part_.world; q++) {
            if (q == part_.rank) continue;
            const int32_t* theirs = nullptr;
            const int      nt = part_of(q, &theirs);
            check(cnf2_unpack_rows(ctx, theirs, nt, (const uint8_t*)buf + (size_t)q * k * B), "cnf2_unpack_rows");
        }
    }
    rows_partial_ = false;
}

void Engine::sync_rows()
{
    gather_private_rows();
    if (rows_stale_) pull_rows();
}

void Engine::accumulators(double* haplobase, double* haplocount)
{
    check(cnf2_download_accumulators(ctx, nullptr, haplobase, haplocount), "cnf2_download_accumulators");
}

// dosureval (cnF2freq.cpp:3084-3097): certainty from the number of supporting relatives and the product of their odds
static double sureval_from(int what, int /*count*/, double oddsproduct)
{
    if (oddsproduct == 0) return 0;
    double t = exp(log(oddsproduct) / what * 4);
    return t / (1 + t);
}

void Engine::postmarkerdata(int indcount)
{
    const bool multi = part_.world > 1 && exchange_ != nullptr;
    if (!multi) {
        postmarkerdata_local(indcount);
        return;
    }
    sync_rows();
    if (part_.rank == 0) {
        const int saved = host_threads();
        if (root_threads_ > 0) set_host_threads(root_threads_);
        postmarkerdata_local(indcount);
        set_host_threads(saved);
    }
    broadcast_state();
}

// rank 0's rows, descendant counts and lock positions to every rank (host memory, in slabs), then up to the devices
void Engine::broadcast_state()
{
    const int    R = (int)P.inds.size();
    const size_t per = (size_t)M * 26;
    const int    slab = (int)std::max<size_t>(1, ROW_SLAB_BYTES / per);
    std::vector<unsigned char> buf((size_t)std::min(R, slab) * per);
    for (int r0 = 0; r0 < R; r0 += slab) {
        const int k = std::min(slab, R - r0);
        if (part_.rank == 0)
#pragma omp parallel for schedule(static) num_threads(host_threads())
            for (int r = 0; r < k; r++) {
                const Individual& I = P.inds[r0 + r];
                unsigned char*    q = buf.data() + (size_t)r * per;
                memcpy(q, I.sure.data(), (size_t)M * 16);
                memcpy(q + (size_t)M * 16, I.hw.data(), (size_t)M * 8);
                memcpy(q + (size_t)M * 24, I.allele.data(), (size_t)M * 2);
            }
        exchange(X_BCAST_HOST, buf.data(), (size_t)k * per, 0, "the broadcast of the rows after postmarkerdata");
        if (part_.rank != 0)
#pragma omp parallel for schedule(static) num_threads(host_threads())
            for (int r = 0; r < k; r++) {
                Individual&          I = P.inds[r0 + r];
                const unsigned char* q = buf.data() + (size_t)r * per;
                memcpy(I.sure.data(), q, (size_t)M * 16);
                memcpy(I.hw.data(), q + (size_t)M * 16, (size_t)M * 8);
                memcpy(I.allele.data(), q + (size_t)M * 24, (size_t)M * 2);
            }
    }
    std::vector<int32_t> meta((size_t)R + lockstart_.size());
    if (part_.rank == 0) {
        std::copy(descendants_.begin(), descendants_.end(), meta.begin());
        std::copy(lockstart_.begin(), lockstart_.end(), meta.begin() + R);
    }
    exchange(X_BCAST_HOST, meta.data(), meta.size() * sizeof(int32_t), 0, "the broadcast of the descendant counts");
    if (part_.rank != 0) {
        std::copy(meta.begin(), meta.begin() + R, descendants_.begin());
        std::copy(meta.begin() + R, meta.end(), lockstart_.begin());
        T = Tables();
        T.dous.assign(P.dous.begin(), P.dous.end());
        push_rows();
    }
}

void Engine::postmarkerdata_local(int indcount)
{
    sync_rows();                // the genotype inference below works on the host copies of the rows
    const int R = (int)P.inds.size();
    // individuals the reference's loops reach: numbers 1 .. indcount - 1
    auto in_scope = [&](int r) { return P.inds[r].n < indcount; };
    std::vector<int32_t> recs;
    for (int r = 0; r < R; r++)
        if (in_scope(r)) recs.push_back(r);
    T = Tables();     // the upload tables are not needed any more (large runs)
    T.dous.assign(P.dous.begin(), P.dous.end());
    const unsigned char UNKNOWN = 0, SEXMARKER = 9;
    // allele value -> (supporting count, product of odds), ascending by value like the reference's std::map; a marker sees
    // a handful of values at most, and the map is formed for every (individual, marker) with an unknown allele
    struct ValMap {
        int    n = 0;
        int    key[8];
        std::pair<int, double> val[8];
        int find(int k) const
        {
            for (int i = 0; i < n; i++)
                if (key[i] == k) return i;
            return -1;
        }
        std::pair<int, double>& at_or_insert(int k, bool* fresh)
        {
            int i = find(k);
            *fresh = i < 0;
            if (i >= 0) return val[i];
            if (n == 8) {                                    // cannot happen with allele values 1, 2, 9; keep the last slot
                *fresh = false;
                return val[7];
            }
            i = n++;
            while (i > 0 && key[i - 1] > k) {
                key[i] = key[i - 1];
                val[i] = val[i - 1];
                i--;
            }
            key[i] = k;
            val[i] = std::make_pair(0, 1.0);
            return val[i];
        }
        void set(int k, std::pair<int, double> v)             // operator[] = v
        {
            bool fresh;
            at_or_insert(k, &fresh) = v;
        }
        void insert(int k, std::pair<int, double> v)          // std::map::insert: keeps an existing entry
        {
            bool fresh;
            std::pair<int, double>& slot = at_or_insert(k, &fresh);
            if (fresh) slot = v;
        }
        void erase(int k)
        {
            const int i = find(k);
            if (i < 0) return;
            for (int j = i; j + 1 < n; j++) {
                key[j] = key[j + 1];
                val[j] = val[j + 1];
            }
            n--;
        }
        size_t size() const { return (size_t)n; }
    };
    int any, anyrem;
    // CNF2_TIMING=1: wall-clock of the steps on stderr (tuning aid)
    const bool timing = getenv("CNF2_TIMING") != nullptr;
    auto       t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  [postmarkerdata] %-32s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
        t_prev = now;
    };
    do {
        for (int r : recs) children_[r] = 0;
        // fixkid (cnF2freq.cpp:1470-1487): a child without a genotype takes the allele of a homozygous parent
        // (markers do not see each other: blocks of markers in parallel, the records in ascending order inside a block --
        // the order in which a parent filled in by its own fixkid is seen by its children stays the sequential one)
        {
            const int blk = 512, nblk = (M + blk - 1) / blk;
#pragma omp parallel for schedule(dynamic, 1) num_threads(host_threads())
            for (int b = 0; b < nblk; b++) {
                const int g0 = b * blk, g1 = std::min(M, g0 + blk);
                for (int r : recs) {
                    Individual& I = P.inds[r];
                    for (int g = g0; g < g1; g++) {
                        if (I.allele[g * 2] != UNKNOWN || I.allele[g * 2 + 1] != UNKNOWN) continue;
                        for (int p = 0; p < 2; p++) {
                            if (I.pars[p] < 0) continue;
                            const Individual& Q = P.inds[I.pars[p]];
                            if (Q.allele[g * 2] == UNKNOWN || Q.allele[g * 2] != Q.allele[g * 2 + 1]) continue;
                            I.allele[g * 2 + p] = Q.allele[g * 2];
                            I.sure[g * 2 + p]   = 0.5;
                        }
                    }
                }
            }
        }
        // descendant counts (cnF2freq.cpp:3224-3255): every individual sends what it has not yet sent to both parents;
        // the counts are never reset, so a further round of the outer loop sends everything again (reference behaviour)
        {
            std::vector<int> upsent(R, 0);
            bool changed;
            do {
                changed = false;
                for (int r : recs) {
                    int now = descendants_[r] ? descendants_[r] : 1;
                    now -= upsent[r];
                    if (now > 0) {
                        for (int k = 0; k < 2; k++)
                            if (P.inds[r].pars[k] >= 0) descendants_[P.inds[r].pars[k]] += now;
                        upsent[r] += now;
                        changed = true;
                    }
                }
            } while (changed);
            for (int r : recs)
                if (descendants_[r] == 0) descendants_[r] = 1;
        }
        for (int r : recs)
            for (int k = 0; k < 2; k++)
                if (P.inds[r].pars[k] >= 0) children_[P.inds[r].pars[k]]++;
        lap("fixkid, descendants");
        // fixparents (cnF2freq.cpp:1392-1468): the admissibility test runs on the GPU for every individual and marker
        push_rows();
        lap("rows to the device");
        std::vector<uint8_t> ok(recs.size() * (size_t)M * 2);
        if (!recs.empty()) check(cnf2_fixparents_scan(ctx, recs.data(), (int)recs.size(), ok.data()), "cnf2_fixparents_scan");
        lap("fixparents scan (GPU)");
        // what fixparents does to the individual itself: no admissible interpretation at all clears the genotype
        std::vector<size_t> qof(R, (size_t)-1);               // record -> position in recs
        for (size_t q = 0; q < recs.size(); q++) qof[recs[q]] = q;
        // (rare: the records that have such a marker are found by all threads, then handled in order, messages and all)
        std::vector<uint8_t> clears(recs.size(), 0);
#pragma omp parallel for schedule(static) num_threads(host_threads())
        for (size_t q = 0; q < recs.size(); q++) {
            const uint8_t* o = &ok[q * M * 2];
            uint8_t        hit = 0;
            for (int g = 0; g < M && !hit; g++) hit = !o[g * 2] && !o[g * 2 + 1];
            clears[q] = hit;
        }
        for (size_t q = 0; q < recs.size(); q++) {
            if (!clears[q]) continue;
            Individual& I = P.inds[recs[q]];
            for (int g = 0; g < M; g++)
                if (!ok[(q * M + g) * 2] && !ok[(q * M + g) * 2 + 1]) {
                    fprintf(stderr, "Clearing %d:%d (was %d,%d)\n", I.n, g, I.allele[g * 2], I.allele[g * 2 + 1]);
                    I.allele[g * 2] = I.allele[g * 2 + 1] = UNKNOWN;
                    I.sure[g * 2] = I.sure[g * 2 + 1] = 0.0;
                }
        }
        // ... and to its parents: an interpretation that stands alone hands the allele on that side to the parent's
        // markervals (value -> count, product of the children's error odds).  Gathered per parent from its children in
        // ascending order -- the order in which the reference's loop visits them -- and only where it is consumed
        // (markers at which the parent lacks an allele), so nothing of size individuals x markers is held.
        std::vector<std::vector<std::pair<int, int>>> kids(R);      // (child record, which parent slot of the child)
        for (int r : recs)
            for (int k = 0; k < 2; k++)
                if (P.inds[r].pars[k] >= 0) kids[P.inds[r].pars[k]].push_back(std::make_pair(r, k));
        auto gather = [&](int r, int g, ValMap& vm) {
            for (auto& ck : kids[r]) {
                const Individual& K = P.inds[ck.first];
                const size_t      q = qof[ck.first];
                const bool ok0 = ok[(q * M + g) * 2] != 0, ok1 = ok[(q * M + g) * 2 + 1] != 0;
                if (ok0 == ok1) continue;                      // only an interpretation that stands alone is passed on
                const int flag2 = ok1 ? 1 : 0;
                const int u = (ck.second ^ flag2) & 1;
                const int value = K.allele[g * 2 + u];
                if (value == UNKNOWN) continue;
                int    oldcount = 0;
                double oldodds = 1;
                const int it = vm.find(value);
                if (it >= 0) {
                    oldcount = vm.val[it].first;
                    oldodds  = vm.val[it].second;
                }
                double probit = K.sure[g * 2] + K.sure[g * 2 + 1];
                probit /= (1 - probit);
                vm.set(value, std::make_pair(oldcount + 1, oldodds * probit));
            }
        };
        // corrections (cnF2freq.cpp:3281-3366, latephase is never set).  The contributions were all made before this
        // loop starts in the reference; a correction below changes a child's alleles only after its parents (lower or
        // higher numbers alike) have been given the old ones, so the children's state is frozen first.
        any = 0;
        anyrem = 0;
        struct Fix { int r, g; uint8_t a0, a1; double s0, s1; };
        // every (record, marker) reads the frozen state only: records in parallel, each thread's corrections appended in
        // record order afterwards (the messages of a run that is not quiet keep their order by running it on one thread)
        std::vector<std::vector<Fix>> fixes_of(recs.size());
        int any_sum = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : any_sum) num_threads(host_threads()) if (opt.quiet)
        for (size_t q_ = 0; q_ < recs.size(); q_++) {
            const int r = recs[q_];
            std::vector<Fix>& fixes = fixes_of[q_];
            int& any = any_sum;
            Individual& I = P.inds[r];
            for (int g = 0; g < M; g++) {
                const int known = (I.allele[g * 2] != UNKNOWN) + (I.allele[g * 2 + 1] != UNKNOWN);
                if (known == 2) continue;
                ValMap vm;
                gather(r, g, vm);
                vm.erase(UNKNOWN);
                if (I.allele[g * 2] != UNKNOWN) vm.insert((int)I.allele[g * 2], std::make_pair(children_[r], I.sure[g * 2]));
                if (I.allele[g * 2 + 1] != UNKNOWN) vm.insert((int)I.allele[g * 2 + 1], std::make_pair(children_[r], I.sure[g * 2 + 1]));
                if (vm.size() >= 3) fprintf(stderr, "Error, too many matches: %d\t%d\n", I.n, g);
                Fix fx = {r, g, I.allele[g * 2], I.allele[g * 2 + 1], I.sure[g * 2], I.sure[g * 2 + 1]};
                bool changed = false;
                if (vm.size() == 2) {
                    const int knowncount = vm.val[0].first + vm.val[1].first;
                    fx.a0 = (uint8_t)vm.key[0];
                    fx.a1 = (uint8_t)vm.key[1];
                    fx.s0 = sureval_from(knowncount, vm.val[0].first, vm.val[0].second);
                    fx.s1 = sureval_from(knowncount, vm.val[1].first, vm.val[1].second);
                    changed = true;
                } else if (vm.size() == 1 && known == 0) {
                    fx.a0 = (uint8_t)vm.key[0];
                    fx.a1 = UNKNOWN;
                    fx.s0 = sureval_from(vm.val[0].first, vm.val[0].first, vm.val[0].second);
                    fx.s1 = 0.0;
                    changed = true;
                }
                if (changed) {
                    any++;
                    fixes.push_back(fx);
                    if (!opt.quiet)
                        printf("Correction at %d, marker %d (%d;%d) (%lf;%lf)\n", I.n, g, fx.a0, fx.a1, fx.s0, fx.s1);
                }
            }
        }
        any = any_sum;
        // a record's corrections touch that record only, and every gather above is done: applied record by record, by all
        // threads (config 5: 57 million of them)
#pragma omp parallel for schedule(dynamic, 16) num_threads(host_threads())
        for (size_t q = 0; q < recs.size(); q++) {
            Individual& I = P.inds[recs[q]];
            for (const Fix& fx : fixes_of[q]) {
                I.allele[fx.g * 2] = fx.a0;
                I.allele[fx.g * 2 + 1] = fx.a1;
                I.sure[fx.g * 2] = fx.s0;
                I.sure[fx.g * 2 + 1] = fx.s1;
            }
            std::vector<Fix>().swap(fixes_of[q]);
            for (int g = 0; g < M; g++)
                if (I.allele[g * 2] == SEXMARKER) std::swap(I.allele[g * 2], I.allele[g * 2 + 1]);
        }
        fprintf(stderr, "Number of corrected genotypes: %d\n", any);
        lap("corrections");
    } while (any > anyrem);

    // variances with the record's own window, founder flags as fixtrees has assigned them so far (cnF2freq.cpp:3373-3389)
    push_rows();
    if (!recs.empty()) {
        // (2 GB at config 5's size: not value-initialised, every element is written by the call)
        std::unique_ptr<double[]> var(new double[recs.size() * (size_t)M]);
        check(cnf2_variances(ctx, recs.data(), (int)recs.size(), 1, var.get()), "cnf2_variances");
#pragma omp parallel for schedule(static) num_threads(host_threads())
        for (size_t q = 0; q < recs.size(); q++)
            for (int g = 0; g < M; g++) {
                const double v = var[q * M + g];
                if (v == v) variances_[(size_t)recs[q] * M + g] = v;       // NaN: the reference leaves the entry alone
            }
    }
    lap("variances (GPU)");
    // lockhaplos (cnF2freq.cpp:3045-3081): per chromosome, lock the phase at the first marker of STRICTLY largest variance.
    // Markers whose genotype configurations are the same have the same variance to the bit in the reference, and markers whose
    // configurations are mirror images of each other have variances that are equal in exact arithmetic and differ there by the
    // rounding of the reference's sums: which one it locks is decided by those bits.  The closed form agrees with the
    // reference's number to ~1e-9, so it says which markers CAN win (those within 1e-8 of the largest); when they hold more
    // than one configuration, the first marker of each is evaluated once more with the reference's own additions in the
    // reference's order (cnf2_variances_exact: its bits) and the reference's comparison is made on those.
    struct Choice {
        int bestpos;           // decided by the closed form alone (one configuration can win), -1: nothing to lock
        int rep_first, rep_n;  // else: its entries among the exact evaluations
    };
    std::vector<Choice>               choice(recs.size() * (size_t)C);
    std::vector<std::vector<int32_t>> reps_of(recs.size());
#pragma omp parallel for schedule(dynamic, 16) num_threads(host_threads())
    for (size_t q = 0; q < recs.size(); q++) {
        const int r = recs[q];
        // the window's members: a configuration is what they hold at the marker
        int rel[7] = {r, P.inds[r].pars[0], -1, -1, P.inds[r].pars[1], -1, -1};
        for (int p = 0; p < 2; p++)
            if (rel[1 + 3 * p] >= 0) {
                rel[2 + 3 * p] = P.inds[rel[1 + 3 * p]].pars[0];
                rel[3 + 3 * p] = P.inds[rel[1 + 3 * p]].pars[1];
            }
        auto same_configuration = [&](int a, int b) {
            for (int k = 0; k < 7; k++) {
                if (rel[k] < 0) continue;
                const Individual& X = P.inds[rel[k]];
                if (X.allele[a * 2] != X.allele[b * 2] || X.allele[a * 2 + 1] != X.allele[b * 2 + 1] ||
                    memcmp(&X.sure[a * 2], &X.sure[b * 2], 2 * sizeof(double)) != 0)
                    return false;
            }
            return true;
        };
        const Individual& I = P.inds[r];
        for (int c = 0; c < C; c++) {
            int& ls = lockstart_[(size_t)r * C + c];
            if (ls >= P.chromstarts[c + 1]) ls = 0;
            Choice&      ch = choice[q * C + c];
            const int    j0 = std::max(P.chromstarts[c], ls), j1 = P.chromstarts[c + 1];
            const double* v = &variances_[(size_t)r * M];
            double        best = 0;
            for (int j = j0; j < j1; j++) best = std::max(best, v[j]);
            const double can_win = best * (1.0 - 1e-8) - 1e-18;
            ch.bestpos = -1;
            ch.rep_first = (int)reps_of[q].size();
            ch.rep_n = 0;
            for (int j = j0; j < j1; j++) {
                if (!(v[j] >= can_win)) continue;
                // both alleles alike: the two terms of every path cancel exactly, there as here (0 never wins)
                if (I.allele[j * 2] == I.allele[j * 2 + 1] && I.sure[j * 2] == I.sure[j * 2 + 1]) continue;
                bool seen = false;
                for (int k = 0; k < ch.rep_n && !seen; k++) seen = same_configuration(reps_of[q][ch.rep_first + k], j);
                if (!seen) {
                    reps_of[q].push_back(j);
                    ch.rep_n++;
                }
            }
            if (ch.rep_n == 1 && best > 1e-18) {            // one configuration can win: its first marker
                ch.bestpos = reps_of[q][ch.rep_first];
                reps_of[q].pop_back();
                ch.rep_n = 0;
            }
        }
    }
    std::vector<int32_t> ex_rec, ex_marker;
    std::vector<size_t>  ex_of(recs.size() + 1, 0);
    for (size_t q = 0; q < recs.size(); q++) {
        ex_of[q] = ex_rec.size();
        ex_rec.insert(ex_rec.end(), reps_of[q].size(), recs[q]);
        ex_marker.insert(ex_marker.end(), reps_of[q].begin(), reps_of[q].end());
    }
    std::vector<double> ex_var(ex_rec.size());
    if (!ex_rec.empty())
        check(cnf2_variances_exact(ctx, ex_rec.data(), ex_marker.data(), (int)ex_rec.size(), 1, ex_var.data()), "cnf2_variances_exact");
    exact_variances_ = (long)ex_rec.size();
    // (records are independent; the messages of a run that is not quiet keep their order by running it on one thread)
    long by_bits = 0;
#pragma omp parallel for schedule(static) num_threads(host_threads()) reduction(+ : by_bits) if (opt.quiet)
    for (size_t q = 0; q < recs.size(); q++) {
        const int   r = recs[q];
        Individual& I = P.inds[r];
        for (int c = 0; c < C; c++) {
            const Choice& ch = choice[q * C + c];
            int           bestpos = ch.bestpos;
            double        bestvar = 0;
            for (int k = 0; k < ch.rep_n; k++) {
                const size_t e = ex_of[q] + ch.rep_first + k;
                // NaN: the reference leaves the entry as it was
                const double x = ex_var[e] == ex_var[e] ? ex_var[e] : variances_[(size_t)r * M + ex_marker[e]];
                if (x > bestvar) {                          // cnF2freq.cpp:3058
                    bestpos = ex_marker[e];
                    bestvar = x;
                }
            }
            if (ch.rep_n > 1) by_bits++;
            if (bestpos == -1) continue;
            if (!opt.quiet) printf("Fixing point: %d %d %d\n", I.n, c + 1, bestpos);
            I.hw[bestpos] = I.hw[bestpos] <= 0.5 ? 0 : 1;
            lockstart_[(size_t)r * C + c] = bestpos + 1;
        }
    }
    locked_by_bits_ = by_bits;
    if (timing)
        fprintf(stderr, "  [lockhaplos] %ld of %zu (record, chromosome) pairs held more than one configuration that could win; %ld entries summed in the reference's order\n",
                locked_by_bits_, recs.size() * (size_t)C, exact_variances_);
    push_rows();
    lap("lockhaplos, rows to the device");
}

bool Engine::deserialize(const char* path)
{
    FILE* f = fopen(path, "rt");
    if (!f) return false;
    sync_rows();
    printf("deserialize started.\n");
    std::vector<char> buf(1 << 16);
    auto getline = [&](std::string& line) -> bool {
        if (!fgets(buf.data(), (int)buf.size(), f)) return false;
        line = buf.data();
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        return true;
    };
    std::string line;
    while (getline(line)) {
        // a header is exactly "<number> <name>"
        int  n = 0, used = 0;
        char name[256], extra[8];
        if (sscanf(line.c_str(), "%d %255s%n %1s", &n, name, &used, extra) != 2) continue;
        int r = -1;
        auto it = P.index.find(name);
        if (it != P.index.end()) r = it->second;
        if (r < 0 || P.inds[r].n != n) {
            fprintf(stderr, "Supposed individual header not a header: %s\n", line.c_str());
            continue;
        }
        Individual& I = P.inds[r];
        int oldphase = 0, switches = 0;
        for (int i = 0; i < M; i++) {
            if (!getline(line)) break;
            double hw, negshift, s1, s2;
            int    a, b;
            if (parse_dump_line(line.c_str(), &hw, &a, &b, &negshift, &s1, &s2) != 6) {       // = sscanf "%lf %d %d %lf %lf %lf"
                fprintf(stderr, "Reading haplotype for marker %d for individual %s failed: %s\n", i, I.name.c_str(), line.c_str());
                continue;
            }
            I.hw[i] = hw;
            bool inv = false, match = true;
            if (!(a == I.allele[i * 2] && b == I.allele[i * 2 + 1])) {
                if (!(b == I.allele[i * 2] && a == I.allele[i * 2 + 1])) {
                    fprintf(stderr, "Genotype mismatch for marker %d for individual %s (%d,%d) to  (%d,%d)\n", i, I.name.c_str(),
                            I.allele[i * 2], I.allele[i * 2 + 1], a, b);
                    match = false;
                } else inv = true;
            }
            I.allele[i * 2]     = (uint8_t)a;
            I.allele[i * 2 + 1] = (uint8_t)b;
            I.sure[i * 2]       = s1;
            I.sure[i * 2 + 1]   = s2;
            if (hw == 0.5 || a == b || !match) continue;
            const int newphase = 1 + ((hw > 0.5) ^ inv);
            if (oldphase && oldphase != newphase) switches++;
            oldphase = newphase;
        }
        const bool par_data = (I.pars[0] >= 0 && !P.inds[I.pars[0]].empty) || (I.pars[1] >= 0 && !P.inds[I.pars[1]].empty);
        if (children_[r] || par_data) printf("Switches %d %s\t%d\n", I.n, I.name.c_str(), switches);
    }
    fclose(f);
    printf("deserialize finished.\n");
    push_rows();
    return true;
}

// Text output by blocks of individuals: `fill(i, buffer)` forms the text of individual i (any thread), the buffers are
// written in order.
template <class F>
static void write_blocks(FILE* out, int n, F&& fill)
{
    const int            T = host_threads(), K = std::max(1, T * 4);
    std::vector<TextBuf> bufs((size_t)std::min(K, std::max(n, 1)));
    for (int i0 = 0; i0 < n; i0 += K) {
        const int nk = std::min(K, n - i0);
#pragma omp parallel for num_threads(T) schedule(dynamic, 1)
        for (int k = 0; k < nk; k++) {
            bufs[k].clear();
            fill(i0 + k, bufs[k]);
        }
        for (int k = 0; k < nk; k++) fwrite(bufs[k].s.data(), 1, bufs[k].n, out);
    }
}

void Engine::iteration(FILE* out)
{
    // CNF2_TIMING=1: wall-clock of the steps of an iteration on stderr (tuning aid)
    const bool timing = getenv("CNF2_TIMING") != nullptr;
    auto       t_prev = std::chrono::steady_clock::now();
    // where the wall time of the iteration went (last_timing): [0] sweep + accumulators, [1] exchanges, [2] update passes,
    // [3] everything else on the host (bookkeeping, likelihood lines, rows), [4] total
    for (double& t : last_timing_) t = 0.0;
    const auto t_start = t_prev;
    auto lap = [&](const char* what, int slot = 3) {
        const auto   now = std::chrono::steady_clock::now();
        const double dt = std::chrono::duration<double>(now - t_prev).count();
        last_timing_[slot] += dt;
        last_timing_[4] = std::chrono::duration<double>(now - t_start).count();
        if (timing) fprintf(stderr, "  [iteration] %-28s %.3f s\n", what, dt);
        t_prev = now;
    };
    const int R = (int)P.inds.size();
    // children = analysed children of every individual (cnF2freq.cpp:5222-5260)
    std::fill(children_.begin(), children_.end(), 0);
    for (int j = 0; j < N; j++)
        for (int k = 0; k < 2; k++)
            if (P.inds[T.dous[j]].pars[k] >= 0) children_[P.inds[T.dous[j]].pars[k]]++;
    std::vector<int32_t> desc(descendants_.begin(), descendants_.end());
    for (auto& d : desc)
        if (d == 0) d = 1;     // postmarkerdata leaves no zero; a run without it counts every individual once
    // this rank's block of analysed individuals (all of them in a single-process run)
    const int b0 = block_end_ < 0 ? 0 : block_begin_, b1 = block_end_ < 0 ? N : block_end_, nb = b1 - b0;
    std::vector<double> factors((size_t)nb * C * 8), loglik((size_t)nb * C), dosage(opt.print_rows ? (size_t)nb * M * 3 : 0);
    const uint32_t rowflag = (opt.normalise ? 0 : CNF2_RAW_DOSAGE) | (deterministic_ ? CNF2_DETERMINISTIC : 0);
    lap("children, descendants");
    if (N > 0) {
        if (opt.update)
            check(cnf2_sweep_accumulate(ctx, b0, b1, desc.data(), factors.data(), loglik.data(),
                                        opt.print_rows ? dosage.data() : nullptr, nullptr, nullptr, nullptr, nullptr, rowflag),
                  "cnf2_sweep_accumulate");
        else if (nb == 0) {}
        else if (!opt.print_rows)
            check(cnf2_sweep(ctx, b0, b1, factors.data(), loglik.data(), nullptr,
                             (opt.merge_modes ? CNF2_MERGE_MODES : 0) | CNF2_NO_DOSAGE),
                  "cnf2_sweep");
        else
            check(cnf2_sweep(ctx, b0, b1, factors.data(), loglik.data(), dosage.data(),
                             (opt.merge_modes ? CNF2_MERGE_MODES : 0) | rowflag),
                  "cnf2_sweep");
    }
    lap("sweep + accumulators", 0);
    const bool multi = part_.world > 1 && exchange_ != nullptr;
    const size_t S = cnf2_packed_accumulator_doubles(ctx), B = cnf2_packed_row_bytes(ctx);
    if (opt.update && N > 0 && multi && part_.seg_shared > 0) {
        // the records several ranks' windows touch hold partial sums: packed by owner, one reduce-scatter, the owner keeps
        // the total (cnF2freq.cpp:6245-6254 reduces per individual; private records need nothing)
        void* buf = nullptr;
        check(cnf2_exchange_buffer(ctx, (size_t)part_.world * part_.seg_shared * std::max(S * sizeof(double), B), &buf), "cnf2_exchange_buffer");
        double* q = (double*)buf;
        for (int k = 0; k < part_.world; k++)
            check(cnf2_pack_accumulators(ctx, part_.shared_of[k].data(), (int)part_.shared_of[k].size(), q + (size_t)k * part_.seg_shared * S),
                  "cnf2_pack_accumulators");
        exchange(X_SUM_SEGMENTS, buf, (size_t)part_.world * part_.seg_shared * S, part_.seg_shared * S, "the sum of the shared accumulators");
        check(cnf2_unpack_accumulators(ctx, part_.shared_of[part_.rank].data(), (int)part_.shared_of[part_.rank].size(),
                                       q + (size_t)part_.rank * part_.seg_shared * S),
              "cnf2_unpack_accumulators");
        lap("exchange of the accumulators", 1);
    }
    pass_hits_.assign(C, 0);
    std::vector<int32_t> wtab;                  // shiftignore / flag2ignore of every window for the likelihood lines: one call
    if (!opt.quiet && N > 0) {
        wtab.resize((size_t)N * 17);
        check(cnf2_window_table(ctx, wtab.data()), "cnf2_window_table");
    }
    for (int c = 0; c < C; c++) {
        // multi-process runs with a spool directory: ranks > 0 write the likelihood lines and the rows of their block to
        // files, rank 0 appends the files behind its own lines / rows in rank order -- the order of a single-process run --
        // and removes them.  The names carry the run's tag (the parent's pid): two runs may share a --tmppath.
        auto spool_name = [&](const char* what, int rank) {
            return opt.spool_dir + "/cnf2_" + what + (opt.spool_tag.empty() ? "" : "_" + opt.spool_tag) + "_it" + std::to_string(iteration_no_) +
                   "_chrom" + std::to_string(c + 1) + "_rank" + std::to_string(rank) + ".txt";
        };
        auto collect = [&](const char* what, FILE* to) {
            exchange(X_BARRIER, nullptr, 0, 0, "the barrier behind the spooled text");
            if (part_.rank != 0) return;
            std::vector<char> buf(1 << 20);
            for (int r = 1; r < part_.world; r++) {
                FILE* f = fopen(spool_name(what, r).c_str(), "r");
                if (!f) throw EngineError(CNF2_ERR_STATE, "cannot read " + spool_name(what, r));
                size_t k;
                while ((k = fread(buf.data(), 1, buf.size(), f)) > 0) fwrite(buf.data(), 1, k, to);
                fclose(f);
                remove(spool_name(what, r).c_str());
            }
        };
        const bool spool_lines = multi && !opt.spool_dir.empty() && !opt.quiet;
        if (!opt.quiet) {
            FILE* lines_out = stdout;                // the two printf of cnF2freq.cpp:5399-5401 go to stdout
            if (spool_lines && part_.rank > 0) {
                lines_out = fopen(spool_name("lines", part_.rank).c_str(), "w");
                if (!lines_out) throw EngineError(CNF2_ERR_STATE, "cannot write " + spool_name("lines", part_.rank));
            }
            for (int j = 0; j < nb; j++) {
                const int32_t* w = &wtab[(size_t)(b0 + j) * 17];
                double mx = -1e15;
                for (int s = 0; s < 8; s++) mx = std::max(mx, factors[((size_t)j * C + c) * 8 + s]);
                fprintf(lines_out, "%d,%03d,%03d: %lf\t%lf %d\n", P.inds[T.dous[b0 + j]].n, w[1], w[0], mx, loglik[(size_t)j * C + c],
                        P.inds[T.dous[b0 + j]].gen < 2 ? 2 : 8);
            }
            if (spool_lines) {
                if (part_.rank > 0) fclose(lines_out);
                collect("lines", stdout);
            }
        }
        const bool spool = multi && !opt.spool_dir.empty() && opt.print_rows;
        FILE* rows_out = out;
        if (spool && part_.rank > 0) {
            rows_out = fopen(spool_name("rows", part_.rank).c_str(), "w");
            if (!rows_out) throw EngineError(CNF2_ERR_STATE, "cannot write " + spool_name("rows", part_.rank));
        }
        // cnF2freq.cpp:6183-6188: "%s:%d\n", a line "%.5lf\t%.5lf\t%.5lf\n" per marker, an empty line (the text is formed
        // by the host's threads, an individual each, and written in order: cnf2_format.h)
        if (opt.print_rows)
            write_blocks(rows_out, nb, [&](int j, TextBuf& tb) {
                const double ll = loglik[(size_t)j * C + c];
                const bool skipped = (ll != ll) || ll < (double)CNF2_MINFACTOR;     // cnF2freq.cpp:5403
                rows_text(P.inds[T.dous[b0 + j]].name, c + 1, &dosage[(size_t)j * M * 3], P.chromstarts[c], P.chromstarts[c + 1], skipped, tb);
            });
        if (spool) {
            if (part_.rank > 0) fclose(rows_out);
            collect("rows", out);
        }
        fflush(out);
        lap("likelihood lines / rows");
        if (!opt.update || N == 0) continue;
        // cnF2freq.cpp:6232-6392: the update pass after this chromosome
        if (opt.print_rows)
            for (int r = 0; r < R; r++) fprintf(out, "FIRST PASS: %d\n", P.inds[r].n);
        int hits = 0;
        // CNF2_DETERMINISTIC runs stay pinned to the bit-exact update form (both certainty flows of a side) unless the caller
        // chose a form itself
        const uint32_t uflags = update_flags_set_ ? update_flags_ : (deterministic_ ? (uint32_t)CNF2_UPDATE_BOTH_FLOWS : 0u);
        if (multi) {
            // this rank's records only; the step-size control needs the hits of all of them (cnF2freq.cpp:6373-6392)
            check(cnf2_update_pass_records(ctx, c, part_.owned.data(), (int)part_.owned.size(), children_.data(), desc.data(),
                                           scalefactor_, entropyfactor_, &hits, uflags),
                  "cnf2_update_pass_records");
            int32_t h = hits;
            exchange(X_SUM_HITS, &h, 1, 1, "the sum of the hit counters");
            hits = h;
        } else
            check(cnf2_update_pass(ctx, c, children_.data(), desc.data(), nullptr, nullptr, nullptr, scalefactor_, entropyfactor_,
                                   &hits, uflags),
                  "cnf2_update_pass");
        if (opt.print_rows)
            for (int r = 0; r < R; r++) fprintf(out, "SKEWNESS PASS: %d\n", P.inds[r].n);
        const int  mx = std::max(oldhits_, oldhits2_), mn = std::min(oldhits_, oldhits2_);
        const bool bad = hits > mx;
        if (bad) scalefactor_ /= 1.1;
        const bool good = hits < std::max(mn, N / 7) * 0.99;
        if (good) scalefactor_ *= 1.21;
        scalefactor_ *= 0.997;
        oldhits2_ = oldhits_;
        oldhits_  = hits;
        last_hits_ = hits;
        pass_hits_[c] = hits;
        fprintf(stdout, "Scale factor now %lf, entropy %lf, hitnnn %d\n", scalefactor_, entropyfactor_, oldhits_);
        lap("update pass", 2);
    }
    if (opt.update && N > 0 && multi) {
        // the new rows of the shared records from their owners to everybody (their next sweep reads them); the private
        // records' rows stay where they are until somebody asks for the whole state (sync_rows)
        if (part_.seg_shared > 0) {
            void* buf = nullptr;
            check(cnf2_exchange_buffer(ctx, (size_t)part_.world * part_.seg_shared * std::max(S * sizeof(double), B), &buf), "cnf2_exchange_buffer");
            uint8_t* q = (uint8_t*)buf;
            check(cnf2_pack_rows(ctx, part_.shared_of[part_.rank].data(), (int)part_.shared_of[part_.rank].size(),
                                 q + (size_t)part_.rank * part_.seg_shared * B),
                  "cnf2_pack_rows");
            exchange(X_GATHER_SEGMENTS, buf, (size_t)part_.world * part_.seg_shared * B, part_.seg_shared * B, "the gather of the shared rows");
            for (int k = 0; k < part_.world; k++)
                if (k != part_.rank)
                    check(cnf2_unpack_rows(ctx, part_.shared_of[k].data(), (int)part_.shared_of[k].size(), q + (size_t)k * part_.seg_shared * B),
                          "cnf2_unpack_rows");
            lap("exchange of the shared rows", 1);
        }
        rows_partial_ = true;
    }
    if (opt.update && N > 0) rows_stale_ = true;
    iteration_no_++;
}

// The dump of cnF2freq.cpp:8157-8192 (cnf2_text.h dump_text), formed by the host's threads
void Engine::dump(FILE* out, int limit)
{
    sync_rows();
    if (part_.world > 1 && part_.rank > 0) return;      // the gather above is the part every rank owes; the text is rank 0's
    std::vector<int> who;
    for (size_t r = 0; r < P.inds.size(); r++)
        if (P.inds[r].n <= limit) who.push_back((int)r);
    write_blocks(out, (int)who.size(), [&](int k, TextBuf& tb) { dump_text(P.inds[who[k]], M, tb); });
}

}  // namespace cnf2host
