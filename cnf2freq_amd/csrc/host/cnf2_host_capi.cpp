// cnf2_host_capi.cpp -- implementation of include/cnf2host.h: the engine of the `cnF2freq` command line
// (cnf2_engine.cpp) driven from arrays.
#include "../../../include/cnf2host.h"

#include <stdio.h>

#include <string>

#include "cnf2_engine.h"

using namespace cnf2host;

struct cnf2h_run {
    Pedigree  P;
    cnf2_ctx* ctx = nullptr;
    Engine*   E = nullptr;
};

static std::string g_err;

// runs f(); an EngineError becomes its code and g_err
template <class F>
static int guarded(F&& f)
{
    try {
        f();
        return 0;
    } catch (const EngineError& e) {
        g_err = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_err = e.what();
        return CNF2_ERR_STATE;
    }
}

extern "C" {

const char* cnf2h_last_error(void) { return g_err.c_str(); }

cnf2h_run* cnf2h_create(int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen, const uint8_t* has_prior,
                        const uint8_t* allele, const double* sure, const double* hw, const double* pos, int n_markers,
                        const int32_t* chromstarts, int n_chrom, const int32_t* dous, int n_dous, int quiet)
{
    return cnf2h_create_on(0, n_rec, par, empty, gen, has_prior, allele, sure, hw, pos, n_markers, chromstarts, n_chrom, dous,
                           n_dous, quiet);
}

cnf2h_run* cnf2h_create_on(int device, int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
                           const uint8_t* has_prior, const uint8_t* allele, const double* sure, const double* hw, const double* pos,
                           int n_markers, const int32_t* chromstarts, int n_chrom, const int32_t* dous, int n_dous, int quiet)
{
    if (n_rec <= 0 || !par || !empty || !gen || !has_prior || !allele || !sure || !hw || !pos || n_markers <= 0 || !chromstarts ||
        n_chrom <= 0 || !dous || n_dous < 0) {
        g_err = "bad arguments";
        return nullptr;
    }
    cnf2h_run* run = new cnf2h_run();
    Pedigree&  P = run->P;
    P.pos.assign(pos, pos + n_markers);
    P.chromstarts.assign(chromstarts, chromstarts + n_chrom + 1);
    P.index["0"] = -1;
    const size_t M = (size_t)n_markers;
    P.inds.resize(n_rec);
#pragma omp parallel for schedule(static) num_threads(host_threads())
    for (int r = 0; r < n_rec; r++) {
        Individual& I = P.inds[r];
        I.n = r + 1;
        I.name = "r" + std::to_string(r);
        I.gen = gen[r];
        I.empty = empty[r] != 0;
        I.pars[0] = par[r * 2];
        I.pars[1] = par[r * 2 + 1];
        I.allele.assign(allele + (size_t)r * M * 2, allele + (size_t)(r + 1) * M * 2);
        I.sure.assign(sure + (size_t)r * M * 2, sure + (size_t)(r + 1) * M * 2);
        I.hw.assign(hw + (size_t)r * M, hw + (size_t)(r + 1) * M);
        I.has_prior = has_prior[r] != 0;
        if (I.has_prior) {
            I.prior_allele = I.allele;
            I.prior_sure = I.sure;
        }
    }
    for (int r = 0; r < n_rec; r++) P.index[P.inds[r].name] = r;
    P.dous.assign(dous, dous + n_dous);
    if (cnf2_ctx_create(device, &run->ctx) != CNF2_OK) {
        g_err = cnf2_last_error(nullptr);
        delete run;
        return nullptr;
    }
    EngineOptions eo;
    eo.quiet = quiet != 0;
    run->E = new Engine(P, run->ctx, eo);
    if (guarded([&] { run->E->upload(); }) != 0) {
        cnf2h_destroy(run);
        return nullptr;
    }
    return run;
}

void cnf2h_destroy(cnf2h_run* run)
{
    if (!run) return;
    delete run->E;
    cnf2_ctx_destroy(run->ctx);
    delete run;
}

int cnf2h_postmarkerdata(cnf2h_run* run, int indcount)
{
    if (!run) return -2;
    return guarded([&] { run->E->postmarkerdata(indcount); });
}

int cnf2h_iteration(cnf2h_run* run, const char* rows_path, int update)
{
    if (!run) return -2;
    FILE* f = fopen(rows_path ? rows_path : "/dev/null", "a");
    if (!f) {
        g_err = "cannot open rows file";
        return -2;
    }
    run->E->set_update(update != 0);
    run->E->set_print_rows(rows_path != nullptr);
    const int rc = guarded([&] { run->E->iteration(f); });
    fclose(f);
    return rc;
}

int cnf2h_dump(cnf2h_run* run, const char* path, int limit)
{
    if (!run || !path) return -2;
    FILE* f = fopen(path, "a");
    if (!f) return -2;
    const int rc = guarded([&] { run->E->dump(f, limit); });
    fclose(f);
    return rc;
}

int cnf2h_deserialize(cnf2h_run* run, const char* path)
{
    if (!run || !path) return -2;
    bool ok = false;
    const int rc = guarded([&] { ok = run->E->deserialize(path); });
    return rc ? rc : (ok ? 0 : -2);
}

int cnf2h_get_state(cnf2h_run* run, uint8_t* allele, double* sure, double* hw, int32_t* descendants, int32_t* children,
                    double* variances, double* scalefactor, int32_t* last_hits)
{
    if (!run) return -2;
    const int rc = guarded([&] { run->E->sync_rows(); });
    if (rc) return rc;
    const Pedigree& P = run->P;
    const size_t    M = P.pos.size();
    for (size_t r = 0; r < P.inds.size(); r++) {
        const Individual& I = P.inds[r];
        if (allele) std::copy(I.allele.begin(), I.allele.end(), allele + r * M * 2);
        if (sure) std::copy(I.sure.begin(), I.sure.end(), sure + r * M * 2);
        if (hw) std::copy(I.hw.begin(), I.hw.end(), hw + r * M);
        if (descendants) descendants[r] = run->E->descendants()[r];
        if (children) children[r] = run->E->children()[r];
    }
    if (variances) std::copy(run->E->variances().begin(), run->E->variances().end(), variances);
    if (scalefactor) *scalefactor = run->E->scalefactor();
    if (last_hits) *last_hits = run->E->last_hits();
    return 0;
}

int cnf2h_set_block(cnf2h_run* run, int begin, int end)
{
    if (!run) return -2;
    return guarded([&] { run->E->set_block(begin, end); });
}

int cnf2h_balanced_block(cnf2h_run* run, int rank, int world, int32_t* begin, int32_t* end)
{
    if (!run || !begin || !end) return -2;
    int b = 0, e = 0;
    const int rc = guarded([&] { run->E->balanced_block(rank, world, &b, &e); });
    *begin = b;
    *end = e;
    return rc;
}

int cnf2h_set_partition(cnf2h_run* run, int rank, int world, cnf2h_exchange_fn fn, void* user)
{
    if (!run) return -2;
    return guarded([&] { run->E->set_partition(rank, world, fn, user); });
}

int cnf2h_get_partition(cnf2h_run* run, int64_t* info10, int32_t* owned)
{
    if (!run || !info10) return -2;
    const Partition& Q = run->E->partition();
    size_t bytes[4];
    run->E->exchange_bytes(bytes);
    const bool set = !Q.bounds.empty();
    info10[0] = set ? Q.bounds[Q.rank] : 0;
    info10[1] = set ? Q.bounds[Q.rank + 1] : (int64_t)run->P.dous.size();
    info10[2] = (int64_t)Q.owned.size();
    info10[3] = (int64_t)Q.n_shared;
    info10[4] = (int64_t)Q.seg_shared;
    for (int i = 0; i < 4; i++) info10[5 + i] = (int64_t)bytes[i];
    info10[9] = set ? (int64_t)Q.private_of[Q.rank].size() : 0;
    if (owned) std::copy(Q.owned.begin(), Q.owned.end(), owned);
    return 0;
}

int cnf2h_reserve(cnf2h_run* run)
{
    if (!run) return -2;
    return guarded([&] { run->E->reserve(); });
}

int cnf2h_get_timing(cnf2h_run* run, double* out5)
{
    if (!run || !out5) return -2;
    std::copy(run->E->last_timing(), run->E->last_timing() + 5, out5);
    return 0;
}

int cnf2h_set_update_flags(cnf2h_run* run, uint32_t flags)
{
    if (!run) return -2;
    run->E->set_update_flags(flags);
    return 0;
}

int cnf2h_set_deterministic(cnf2h_run* run, int on)
{
    if (!run) return -2;
    run->E->set_deterministic(on != 0);
    return 0;
}

void* cnf2h_context(cnf2h_run* run) { return run ? run->ctx : nullptr; }

int cnf2h_get_passes(cnf2h_run* run, int32_t* hits, double* haplobase, double* haplocount)
{
    if (!run) return -2;
    if (hits) std::copy(run->E->pass_hits().begin(), run->E->pass_hits().end(), hits);
    if (haplobase || haplocount) return guarded([&] { run->E->accumulators(haplobase, haplocount); });
    return 0;
}

}  // extern "C"
