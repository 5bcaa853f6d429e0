// cnf2_capi.hip -- implementation of the C ABI declared in include/cnf2hip.h.
// Host side only: owns device memory, derives the window tables, builds the job list and
// launches the kernels of cnf2_kernels.hip on the context's stream.  There is no CPU
// compute path here: without a usable HIP device every entry point fails.
#include "../../include/cnf2hip.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <chrono>
#include <stdio.h>
#include <string.h>

#include <string>
#include <algorithm>
#include <vector>

#include "cnf2_device.h"
#include "cnf2_emission.h"

using namespace cnf2;

struct cnf2_ctx {
    int         device = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // tied windows (general kernel) run beside the fast kernel
    hipEvent_t  ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    bool        timed = false;
    std::string err;
    int         n_cu = 0;
    int         blocks_per_cu = 1;

    // map
    int                  n_markers = 0, n_chrom = 0;
    std::vector<int32_t> chromstarts;
    double               genrec[3] = {-0.02, -0.02, -0.02};
    double2*             d_rho = nullptr;
    double2*             d_tq = nullptr;
    double*              d_logk = nullptr;

    // rows
    int      n_rows = 0;
    uint8_t* d_allele8 = nullptr;
    double2* d_sure = nullptr;
    double*  d_hw = nullptr;

    // pedigree
    HostPedigree        ped;
    std::vector<Window> windows;     // one per analysed individual
    Window*             d_windows = nullptr;
    bool                windows_dirty = true;   // rows or pedigree changed since the last derivation
    uint8_t*            d_rowflags = nullptr;
    int                 fast_blocks_per_cu = 1;
    int                 reserve_blocks = 0;     // workgroup slots left free for concurrent kernels (RCCL)
    int                 batch_jobs = 0;         // cap on the jobs per batch of the batched consumers (0 = what memory allows)

    // workspace
    Job*    d_jobs = nullptr;
    size_t  jobs_cap = 0;
    PackedJob* d_pjobs = nullptr;
    size_t  pjobs_cap = 0;
    double* d_spill = nullptr;
    size_t  spill_bytes = 0;
    double *d_factors = nullptr, *d_loglik = nullptr, *d_dosage = nullptr;
    size_t  factors_cap = 0, loglik_cap = 0, dosage_cap = 0;
    int32_t* d_lexp = nullptr;                 // binary exponents of the fast kernel's likelihoods: [n][C][8] then [n][C]
    size_t   lexp_cap = 0;
    unsigned long long* d_clock = nullptr;     // [4] clock stamps of the last plain fast-kernel launch
    int*     d_jobnext = nullptr;              // [4] job counters of the fast-kernel launches in flight (KernelParams::job_next)
    size_t   jobnext_cap = 0;
    size_t   clock_cap = 0;
    double* d_scratch = nullptr;     // small parity buffers
    size_t  scratch_cap = 0;

    // batched HOT LOOP 2 (cnf2_sweep_accumulate)
    std::vector<int32_t> slot_rec;   // [n_dous][7] record per window slot (derive_window), -1 none
    int32_t* d_slot_rec = nullptr;
    int32_t* d_desc = nullptr;
    uint8_t* d_rec_empty = nullptr;
    size_t   rec_cap = 0;
    double*  d_wbuf = nullptr;
    size_t   wbuf_cap = 0;
    double * d_acc_inf = nullptr, *d_acc_hb = nullptr, *d_acc_hc = nullptr, *d_acc_hz = nullptr;
    size_t   acc_inf_cap = 0, acc_hb_cap = 0, acc_hc_cap = 0, acc_hz_cap = 0;

    // per-iteration updates (cnf2_update_pass) and pre-processing scans
    uint8_t* d_prior_allele8 = nullptr;
    double2* d_prior_sure = nullptr;
    uint8_t* d_has_prior = nullptr;      // [n_rec]
    bool     priors_set = false;
    int32_t* d_row_of = nullptr;
    int32_t* d_children = nullptr;
    int32_t* d_chromstarts = nullptr;
    size_t   upd_rec_cap = 0;
    uint8_t* d_anyinfo = nullptr;
    size_t   anyinfo_cap = 0;
    double * d_fw = nullptr, *d_ratio = nullptr;
    size_t   fw_cap = 0, ratio_cap = 0;
    int*     d_hits = nullptr;
    unsigned long long* d_flow_next = nullptr;
    double*  d_flow_out = nullptr;
    double*  d_todo = nullptr;            // flows set aside by the scouts (3 doubles each)
    size_t   todo_cap = 0;
    double*  d_part = nullptr;            // CNF2_DETERMINISTIC rows
    size_t   part_cap = 0;
    int32_t* d_gather = nullptr;          // rec_start [n_rec + 1] followed by the list
    size_t   gather_cap = 0;
    size_t   flow_out_cap = 0;
    int32_t* d_pathlog = nullptr;
    size_t   pathlog_cap = 0;
    int      pathlog_n = 0;
    int32_t* d_updrecs = nullptr;         // the records an update pass is restricted to
    size_t   updrecs_cap = 0;
    int32_t* d_xidx = nullptr;            // index lists of the exchange packers
    size_t   xidx_cap = 0;
    uint8_t* d_xbuf = nullptr;            // staging buffer of the exchanges (cnf2_exchange_buffer)
    size_t   xbuf_cap = 0;
    Window*  d_scanwin = nullptr;
    size_t   scanwin_cap = 0;
    uint8_t* d_okout = nullptr;
    size_t   okout_cap = 0;
};

static std::string g_create_error;

static int fail(cnf2_ctx* ctx, int code, const char* fmt, ...)
{
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, call)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, e_ == hipErrorOutOfMemory ? CNF2_ERR_NOMEM : CNF2_ERR_HIP,        \
                        "%s failed: %s", #call, hipGetErrorString(e_));                        \
    } while (0)

template <class T>
static int ensure(cnf2_ctx* ctx, T** ptr, size_t* cap, size_t count)
{
    if (*cap >= count && *ptr) return CNF2_OK;
    if (*ptr) HIP_TRY(ctx, hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)ptr, count * sizeof(T)));
    *cap = count;
    return CNF2_OK;
}

extern "C" {

const char* cnf2_version(void) { return "cnf2hip 0.1 (gfx950)"; }

int cnf2_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* cnf2_last_error(const cnf2_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int cnf2_ctx_create(int device, cnf2_ctx** out)
{
    if (!out) return fail(nullptr, CNF2_ERR_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, CNF2_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(nullptr, CNF2_ERR_ARG, "device %d out of range (%d devices)", device, n);
    cnf2_ctx* ctx = new cnf2_ctx();
    ctx->device   = device;
    hipError_t e  = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreate(&ctx->stream);
    if (e == hipSuccess) e = hipStreamCreate(&ctx->stream2);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev2, hipEventDisableTiming);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        fail(nullptr, CNF2_ERR_HIP, "context creation failed: %s", hipGetErrorString(e));
        delete ctx;
        return CNF2_ERR_HIP;
    }
    ctx->n_cu          = prop.multiProcessorCount;
    ctx->blocks_per_cu = fb_blocks_per_cu();
    ctx->fast_blocks_per_cu = fb_fast_blocks_per_cu();
    *out = ctx;
    return CNF2_OK;
}

void cnf2_ctx_destroy(cnf2_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->d_rho);
    (void)hipFree(ctx->d_tq);
    (void)hipFree(ctx->d_logk);
    (void)hipFree(ctx->d_allele8);
    (void)hipFree(ctx->d_sure);
    (void)hipFree(ctx->d_hw);
    (void)hipFree(ctx->d_windows);
    (void)hipFree(ctx->d_rowflags);
    (void)hipFree(ctx->d_jobs);
    (void)hipFree(ctx->d_pjobs);
    (void)hipFree(ctx->d_spill);
    (void)hipFree(ctx->d_factors);
    (void)hipFree(ctx->d_loglik);
    (void)hipFree(ctx->d_lexp);
    (void)hipFree(ctx->d_clock);
    (void)hipFree(ctx->d_jobnext);
    (void)hipFree(ctx->d_dosage);
    (void)hipFree(ctx->d_scratch);
    (void)hipFree(ctx->d_slot_rec);
    (void)hipFree(ctx->d_desc);
    (void)hipFree(ctx->d_rec_empty);
    (void)hipFree(ctx->d_wbuf);
    (void)hipFree(ctx->d_acc_inf);
    (void)hipFree(ctx->d_acc_hb);
    (void)hipFree(ctx->d_acc_hc);
    (void)hipFree(ctx->d_acc_hz);
    (void)hipFree(ctx->d_prior_allele8);
    (void)hipFree(ctx->d_prior_sure);
    (void)hipFree(ctx->d_has_prior);
    (void)hipFree(ctx->d_row_of);
    (void)hipFree(ctx->d_children);
    (void)hipFree(ctx->d_chromstarts);
    (void)hipFree(ctx->d_anyinfo);
    (void)hipFree(ctx->d_fw);
    (void)hipFree(ctx->d_ratio);
    (void)hipFree(ctx->d_hits);
    (void)hipFree(ctx->d_flow_next);
    (void)hipFree(ctx->d_flow_out);
    (void)hipFree(ctx->d_todo);
    (void)hipFree(ctx->d_part);
    (void)hipFree(ctx->d_gather);
    (void)hipFree(ctx->d_updrecs);
    (void)hipFree(ctx->d_xidx);
    (void)hipFree(ctx->d_xbuf);
    (void)hipFree(ctx->d_scanwin);
    (void)hipFree(ctx->d_pathlog);
    (void)hipFree(ctx->d_okout);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipEventDestroy(ctx->ev2);
    (void)hipStreamDestroy(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void* cnf2_stream(cnf2_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int cnf2_set_grid_reserve(cnf2_ctx* ctx, int blocks)
{
    if (!ctx || blocks < 0) return CNF2_ERR_ARG;
    ctx->reserve_blocks = blocks;
    return CNF2_OK;
}

int cnf2_set_batch_jobs(cnf2_ctx* ctx, int jobs)
{
    if (!ctx || jobs < 0) return CNF2_ERR_ARG;
    ctx->batch_jobs = jobs;
    return CNF2_OK;
}

int cnf2_sync(cnf2_ctx* ctx)
{
    if (!ctx) return CNF2_ERR_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_upload_map(cnf2_ctx* ctx, const double* pos, int n_markers, const int32_t* chromstarts, int n_chrom,
                    const double* genrec)
{
    if (!ctx || !pos || !chromstarts || n_markers <= 0 || n_chrom <= 0) return fail(ctx, CNF2_ERR_ARG, "bad map arguments");
    if (chromstarts[0] != 0 || chromstarts[n_chrom] != n_markers)
        return fail(ctx, CNF2_ERR_ARG, "chromstarts must run from 0 to n_markers");
    for (int c = 0; c < n_chrom; c++)
        if (chromstarts[c + 1] <= chromstarts[c]) return fail(ctx, CNF2_ERR_ARG, "empty chromosome %d", c);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->n_markers != n_markers && (ctx->d_allele8 || ctx->d_sure || ctx->d_hw))
        return fail(ctx, CNF2_ERR_STATE, "marker count changed after rows were uploaded");
    ctx->n_markers = n_markers;
    ctx->n_chrom   = n_chrom;
    ctx->chromstarts.assign(chromstarts, chromstarts + n_chrom + 1);
    if (genrec) memcpy(ctx->genrec, genrec, sizeof(ctx->genrec));
    // recombination fraction per gap, exactly as realanalyze forms it (cnF2freq.cpp:2270-2286);
    // a gap with dist <= 0 performs no transition (cnF2freq.cpp:2273) == rho 0
    std::vector<double2> rho(n_markers);
    for (int m = 0; m < n_markers; m++) {
        double2 r = make_double2(0.0, 0.0);
        if (m + 1 < n_markers) {
            double dist = pos[m + 1] - pos[m];
            if (dist > 0) {
                r.x = 0.5 * (1.0 - exp(ctx->genrec[0] * dist));
                r.y = 0.5 * (1.0 - exp(ctx->genrec[1] * dist));
            }
        }
        rho[m] = r;
    }
    if (ctx->d_rho) HIP_TRY(ctx, hipFree(ctx->d_rho));
    ctx->d_rho = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_rho, sizeof(double2) * n_markers));
    HIP_TRY(ctx, hipMemcpy(ctx->d_rho, rho.data(), sizeof(double2) * n_markers, hipMemcpyHostToDevice));
    // fast kernel: butterflies x' = x + t * partner with t = r / (1 - r); the dropped scalar
    // (1-r0)^4 (1-r1)^2 per gap (bits with TYPEGENS 0: four, TYPEGENS 1: two, settings.h:23) is
    // accounted for in the log-likelihoods through its logarithm summed over each chromosome
    std::vector<double2> tq(n_markers);
    std::vector<double>  logk(n_chrom, 0.0);
    for (int c = 0; c < n_chrom; c++)
        for (int m = chromstarts[c]; m < chromstarts[c + 1]; m++) {
            tq[m] = make_double2(rho[m].x / (1.0 - rho[m].x), rho[m].y / (1.0 - rho[m].y));
            if (m + 1 < chromstarts[c + 1]) logk[c] += 4.0 * log1p(-rho[m].x) + 2.0 * log1p(-rho[m].y);
        }
    if (ctx->d_tq) HIP_TRY(ctx, hipFree(ctx->d_tq));
    if (ctx->d_logk) HIP_TRY(ctx, hipFree(ctx->d_logk));
    ctx->d_tq = nullptr;
    ctx->d_logk = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_tq, sizeof(double2) * n_markers));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_logk, sizeof(double) * n_chrom));
    HIP_TRY(ctx, hipMemcpy(ctx->d_tq, tq.data(), sizeof(double2) * n_markers, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_logk, logk.data(), sizeof(double) * n_chrom, hipMemcpyHostToDevice));
    return CNF2_OK;
}

static int copy_rows(cnf2_ctx* ctx, int row0, int n, const uint8_t* allele, const double* sure, const double* hw)
{
    const size_t M = ctx->n_markers;
    const size_t cnt = (size_t)n * M;
    // pack (a0, a1) into one byte on the host, in slabs to bound the staging buffer
    const size_t slab = 1u << 24;
    std::vector<uint8_t> packed(cnt < slab ? cnt : slab);
    for (size_t off = 0; off < cnt; off += slab) {
        size_t k = cnt - off < slab ? cnt - off : slab;
        for (size_t i = 0; i < k; i++) {
            uint8_t a0 = allele[(off + i) * 2], a1 = allele[(off + i) * 2 + 1];
            if (a0 > 15 || a1 > 15) return fail(ctx, CNF2_ERR_ARG, "allele value out of range");
            packed[i] = (uint8_t)(a0 | (a1 << 4));
        }
        HIP_TRY(ctx, hipMemcpy(ctx->d_allele8 + (size_t)row0 * M + off, packed.data(), k, hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, hipMemcpy(ctx->d_sure + (size_t)row0 * M, sure, cnt * sizeof(double2), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_hw + (size_t)row0 * M, hw, cnt * sizeof(double), hipMemcpyHostToDevice));
    return CNF2_OK;
}

static int blank_rows(cnf2_ctx* ctx)
{
    const size_t cnt = (size_t)ctx->n_rows * ctx->n_markers;
    HIP_TRY(ctx, hipMemset(ctx->d_allele8, 0, cnt));
    HIP_TRY(ctx, hipMemset(ctx->d_sure, 0, cnt * sizeof(double2)));
    // haploweight of an individual without data is 0.5 (getind, cnF2freq.cpp:2491)
    // (one row from the host, then the filled part copied onto the rest, doubling: ~log2(rows) copies instead of one per row)
    const size_t        M = (size_t)ctx->n_markers;
    std::vector<double> half(M, 0.5);
    HIP_TRY(ctx, hipMemcpy(ctx->d_hw, half.data(), sizeof(double) * M, hipMemcpyHostToDevice));
    for (size_t filled = 1; filled < (size_t)ctx->n_rows; filled *= 2) {
        const size_t k = std::min(filled, (size_t)ctx->n_rows - filled);
        HIP_TRY(ctx, hipMemcpy(ctx->d_hw + filled * M, ctx->d_hw, sizeof(double) * M * k, hipMemcpyDeviceToDevice));
    }
    return CNF2_OK;
}

int cnf2_upload_rows(cnf2_ctx* ctx, int n_rows, const uint8_t* allele, const double* sure, const double* hw)
{
    const bool blank = !allele && !sure && !hw;
    if (!ctx || n_rows <= 0 || (!blank && (!allele || !sure || !hw))) return fail(ctx, CNF2_ERR_ARG, "bad row arguments");
    if (ctx->n_markers <= 0) return fail(ctx, CNF2_ERR_STATE, "upload the map before the rows");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_allele8) HIP_TRY(ctx, hipFree(ctx->d_allele8));
    if (ctx->d_sure) HIP_TRY(ctx, hipFree(ctx->d_sure));
    if (ctx->d_hw) HIP_TRY(ctx, hipFree(ctx->d_hw));
    ctx->d_allele8 = nullptr;
    ctx->d_sure = nullptr;
    ctx->d_hw = nullptr;
    ctx->n_rows = 0;
    size_t cnt = (size_t)n_rows * ctx->n_markers;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_allele8, cnt));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_sure, cnt * sizeof(double2)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_hw, cnt * sizeof(double)));
    ctx->n_rows = n_rows;
    ctx->windows_dirty = true;
    ctx->priors_set = false;
    // a pedigree uploaded against a larger table would index past the new one: drop it, it must be uploaded again
    for (int32_t r : ctx->ped.row_of)
        if (r >= n_rows) {
            ctx->ped = HostPedigree();
            ctx->windows.clear();
            break;
        }
    if (blank) return blank_rows(ctx);
    return copy_rows(ctx, 0, n_rows, allele, sure, hw);
}

int cnf2_update_rows_device(cnf2_ctx* ctx, int row0, int n, const uint8_t* d_allele8, const double* d_sure,
                            const double* d_hw)
{
    if (!ctx || !d_allele8 || !d_sure || !d_hw) return fail(ctx, CNF2_ERR_ARG, "bad row arguments");
    if (!ctx->d_allele8) return fail(ctx, CNF2_ERR_STATE, "no rows uploaded");
    if (row0 < 0 || n < 0 || row0 + n > ctx->n_rows) return fail(ctx, CNF2_ERR_ARG, "row range out of bounds");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t M = ctx->n_markers, cnt = (size_t)n * M;
    ctx->windows_dirty = true;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_allele8 + (size_t)row0 * M, d_allele8, cnt, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_sure + (size_t)row0 * M, d_sure, cnt * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_hw + (size_t)row0 * M, d_hw, cnt * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_update_rows(cnf2_ctx* ctx, int row0, int n, const uint8_t* allele, const double* sure, const double* hw)
{
    if (!ctx || !allele || !sure || !hw) return fail(ctx, CNF2_ERR_ARG, "bad row arguments");
    if (!ctx->d_allele8) return fail(ctx, CNF2_ERR_STATE, "no rows uploaded");
    if (row0 < 0 || n < 0 || row0 + n > ctx->n_rows) return fail(ctx, CNF2_ERR_ARG, "row range out of bounds");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (n == 0) return CNF2_OK;
    ctx->windows_dirty = true;
    return copy_rows(ctx, row0, n, allele, sure, hw);
}

int cnf2_upload_pedigree(cnf2_ctx* ctx, int n_rec, const int32_t* par, const uint8_t* empty, const int32_t* gen,
                         const int32_t* row_of, const int32_t* dous, int n_dous)
{
    if (!ctx || n_rec <= 0 || !par || !empty || !gen || !row_of || !dous || n_dous < 0)
        return fail(ctx, CNF2_ERR_ARG, "bad pedigree arguments");
    if (ctx->n_rows <= 0) return fail(ctx, CNF2_ERR_STATE, "upload the rows before the pedigree");
    for (int r = 0; r < n_rec; r++) {
        for (int k = 0; k < 2; k++)
            if (par[r * 2 + k] < -1 || par[r * 2 + k] >= n_rec) return fail(ctx, CNF2_ERR_ARG, "parent index out of range at record %d", r);
        if (row_of[r] < 0 || row_of[r] >= ctx->n_rows) return fail(ctx, CNF2_ERR_ARG, "row index out of range at record %d", r);
    }
    for (int j = 0; j < n_dous; j++)
        if (dous[j] < 0 || dous[j] >= n_rec) return fail(ctx, CNF2_ERR_ARG, "analysed record out of range at %d", j);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HostPedigree& P = ctx->ped;
    P.n_rec = n_rec;
    P.par.assign(par, par + 2 * (size_t)n_rec);
    P.empty.assign(empty, empty + n_rec);
    P.gen.assign(gen, gen + n_rec);
    P.row_of.assign(row_of, row_of + n_rec);
    P.dous.assign(dous, dous + n_dous);
    derive_founders(P);
    P.row_hom.clear();
    ctx->windows.assign(n_dous, Window());
    ctx->windows_dirty = true;
    ctx->priors_set = false;         // the per-record prior flags belong to the pedigree that was replaced
    return CNF2_OK;
}

// (Re)derive the window tables: per-row "always homozygous" flags from the device rows, then
// fixtrees per analysed individual.  Runs lazily before a sweep after rows or pedigree changed.
static int prepare_windows(cnf2_ctx* ctx)
{
    if (!ctx->windows_dirty) return CNF2_OK;
    HostPedigree& P = ctx->ped;
    const int n_dous = (int)P.dous.size();
    if (ctx->d_rowflags) HIP_TRY(ctx, hipFree(ctx->d_rowflags));
    ctx->d_rowflags = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_rowflags, ctx->n_rows));
    launch_row_flags(ctx->d_allele8, ctx->d_sure, ctx->n_rows, ctx->n_markers, ctx->d_rowflags, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    P.row_hom.assign(ctx->n_rows, 0);
    HIP_TRY(ctx, hipMemcpyAsync(P.row_hom.data(), ctx->d_rowflags, ctx->n_rows, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->windows.resize(n_dous);
    ctx->slot_rec.assign((size_t)n_dous * 7, -1);
    for (int j = 0; j < n_dous; j++) derive_window(P, P.dous[j], &ctx->windows[j], ctx->slot_rec.data() + (size_t)j * 7);
    if (ctx->d_windows) HIP_TRY(ctx, hipFree(ctx->d_windows));
    if (ctx->d_slot_rec) HIP_TRY(ctx, hipFree(ctx->d_slot_rec));
    ctx->d_windows = nullptr;
    ctx->d_slot_rec = nullptr;
    if (n_dous > 0) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_windows, sizeof(Window) * n_dous));
        HIP_TRY(ctx, hipMemcpy(ctx->d_windows, ctx->windows.data(), sizeof(Window) * n_dous, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_slot_rec, sizeof(int32_t) * 7 * n_dous));
        HIP_TRY(ctx, hipMemcpy(ctx->d_slot_rec, ctx->slot_rec.data(), sizeof(int32_t) * 7 * n_dous, hipMemcpyHostToDevice));
    }
    ctx->windows_dirty = false;
    return CNF2_OK;
}

int cnf2_window_info(cnf2_ctx* ctx, int ind, int32_t* out17)
{
    if (!ctx || !out17) return CNF2_ERR_ARG;
    if (ind < 0 || ind >= (int)ctx->windows.size()) return fail(ctx, CNF2_ERR_ARG, "individual out of range");
    Window  w;
    int32_t slot_rec[7];
    {
        // topology exactly as fixtrees leaves it: tie groups are not pruned by row content here
        std::vector<uint8_t> keep;
        keep.swap(ctx->ped.row_hom);
        derive_window(ctx->ped, ctx->ped.dous[ind], &w, slot_rec);
        keep.swap(ctx->ped.row_hom);
    }
    out17[0] = w.shiftignore;
    out17[1] = w.flag2ignore;
    out17[2] = ctx->ped.founder[ctx->ped.dous[ind]];
    for (int i = 0; i < 7; i++) {
        out17[3 + i]  = slot_rec[i];
        out17[10 + i] = w.tie[i];
    }
    return CNF2_OK;
}

int cnf2_window_table(cnf2_ctx* ctx, int32_t* out17_all)
{
    if (!ctx || !out17_all) return CNF2_ERR_ARG;
    const int n = (int)ctx->windows.size();
    for (int j = 0; j < n; j++) {
        const int rc = cnf2_window_info(ctx, j, out17_all + (size_t)j * 17);
        if (rc) return rc;
    }
    return CNF2_OK;
}

static int ready(cnf2_ctx* ctx)
{
    if (!ctx) return CNF2_ERR_ARG;
    if (!ctx->d_rho || !ctx->d_allele8 || ctx->ped.n_rec == 0)
        return fail(ctx, CNF2_ERR_STATE, "map, rows and pedigree must be uploaded first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = prepare_windows(ctx);
    if (rc) return rc;
    const size_t ne = ctx->windows.size() * (size_t)ctx->n_chrom;
    if ((rc = ensure(ctx, &ctx->d_lexp, &ctx->lexp_cap, ne * 9 + 1))) return rc;
    if (!ctx->d_clock) {
        if ((rc = ensure(ctx, &ctx->d_clock, &ctx->clock_cap, (size_t)4))) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_clock, 0, 4 * sizeof(unsigned long long), ctx->stream));
    }
    if ((rc = ensure(ctx, &ctx->d_jobnext, &ctx->jobnext_cap, (size_t)4))) return rc;
    return CNF2_OK;
}

static void base_params(cnf2_ctx* ctx, KernelParams* p)
{
    memset(p, 0, sizeof(*p));
    p->windows   = ctx->d_windows;
    p->allele8   = ctx->d_allele8;
    p->sure      = ctx->d_sure;
    p->hw        = ctx->d_hw;
    p->rho       = ctx->d_rho;
    p->tq        = ctx->d_tq;
    p->chrom_logk = ctx->d_logk;
    p->n_markers = ctx->n_markers;
    p->n_chrom   = ctx->n_chrom;
    p->fexp      = ctx->d_lexp;
    p->lexp      = ctx->d_lexp + ctx->windows.size() * (size_t)ctx->n_chrom * 8;
    p->job_next  = ctx->d_jobnext;
}

static int max_chrom_len(const cnf2_ctx* ctx)
{
    int mx = 0;
    for (int c = 0; c < ctx->n_chrom; c++) {
        int l = ctx->chromstarts[c + 1] - ctx->chromstarts[c];
        if (l > mx) mx = l;
    }
    return mx;
}

// A batch of the batched consumers is swept by the resident waves in rounds (a wave takes the next job when it has finished
// one): 6.1 rounds take the time of 7.  When the jobs do not fit one batch, a batch is a whole number of rounds.
static size_t whole_rounds(size_t batch, size_t n_jobs, int grid_cap)
{
    const size_t waves = (size_t)grid_cap * CNF2_WAVES_PER_BLOCK;
    if (batch >= n_jobs || batch < waves) return batch;
    return batch / waves * waves;
}

// The chromosomes in the order their jobs are listed: longest first (ties in map order).  The waves of a launch take the
// jobs in list order (KernelParams::job_next), so the long jobs start first and a launch ends on the short ones; the jobs of
// one chromosome keep the order of the individuals, so nothing that adds up over individuals sees a difference.
static std::vector<int> chrom_order(const cnf2_ctx* ctx)
{
    std::vector<int> o(ctx->n_chrom);
    for (int c = 0; c < ctx->n_chrom; c++) o[c] = c;
    std::stable_sort(o.begin(), o.end(), [&](int a, int b) {
        return ctx->chromstarts[a + 1] - ctx->chromstarts[a] > ctx->chromstarts[b + 1] - ctx->chromstarts[b];
    });
    return o;
}

int cnf2_sweep(cnf2_ctx* ctx, int ind_begin, int ind_end, double* factors_out, double* loglik_out, double* dosage_out,
               uint32_t flags)
{
    int rc = ready(ctx);
    if (rc) return rc;
    const int n_all = (int)ctx->windows.size();
    if (ind_begin < 0 || ind_end > n_all || ind_begin > ind_end) return fail(ctx, CNF2_ERR_ARG, "individual range out of bounds");
    const bool want_dosage = !(flags & CNF2_NO_DOSAGE);
    if (!factors_out || !loglik_out || (want_dosage && !dosage_out)) return fail(ctx, CNF2_ERR_ARG, "output pointer is NULL");
    const int n = ind_end - ind_begin;
    if (n == 0) return CNF2_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    const size_t n_jobs = (size_t)n * ctx->n_chrom;
    if (n_jobs > 0x7fffffff) return fail(ctx, CNF2_ERR_ARG, "too many jobs in one call");
    // job list: individuals x chromosomes (the loops at cnF2freq.cpp:5283 and 5294), windows
    // without an active tie group first (fast kernel), tied windows after (general kernel)
    // CNF2_MERGE_MODES: windows whose two parents are homozygous with equal sure everywhere go four to a
    // wavefront (fb_packed_kernel); groups are formed per chromosome, what does not fill a group of four
    // stays with the ordinary kernel
    const bool merge = (flags & CNF2_MERGE_MODES) && !(flags & (CNF2_FULL_SPILL | CNF2_FLUSH_TINY));
    auto mergeable = [&](const Window& w) {
        if (w.n_groups > 0 && !(flags & CNF2_NO_TIES)) return false;
        if (w.shiftignore != 0 || w.shiftend != 8 || (w.flags[0] & SLOT_FOUNDER)) return false;
        for (int k = 1; k <= 4; k += 3) {
            if (!(w.flags[k] & SLOT_PRESENT) || w.row[k] < 0) return false;
            if (!ctx->ped.row_hom[w.row[k]]) return false;
        }
        return true;
    };
    std::vector<uint8_t>   packed(merge ? n : 0, 0);
    std::vector<PackedJob> pjobs;
    if (merge) {
        // a group shares the producer's instantiation: windows whose grandparents are all present and
        // homozygous everywhere (SLOT_HOM) are grouped apart from the others
        auto homleaf = [&](const Window& w) {
            const int gp = w.flags[2] & w.flags[3] & w.flags[5] & w.flags[6];
            return (gp & SLOT_HOM) && (gp & SLOT_PRESENT);
        };
        for (int cls = 0; cls < 2; cls++) {
            std::vector<int> el;
            for (int j = 0; j < n; j++) {
                const Window& w = ctx->windows[ind_begin + j];
                if (mergeable(w) && (homleaf(w) ? 1 : 0) == cls) el.push_back(j);
            }
            const size_t full = el.size() / 4 * 4;
            for (size_t k = 0; k < full; k++) packed[el[k]] = 1;
            for (int c : chrom_order(ctx))
                for (size_t k = 0; k < full; k += 4) {
                    PackedJob pj;
                    for (int i = 0; i < 4; i++) pj.ind[i] = el[k + i];
                    pj.first   = ctx->chromstarts[c];
                    pj.last    = ctx->chromstarts[c + 1] - 1;
                    pj.chrom   = c;
                    pj.homleaf = cls;
                    pjobs.push_back(pj);
                }
        }
    }
    const size_t n_packed = pjobs.size();
    std::vector<Job> jobs;
    jobs.reserve(n_jobs);
    size_t n_fast = 0;
    for (int pass = 0; pass < 2; pass++) {
        for (int c : chrom_order(ctx))
            for (int j = 0; j < n; j++) {
                if (merge && packed[j]) continue;
                // (CNF2_FLUSH_TINY: every window takes the general kernel's route, the second list)
                const bool tied = (ctx->windows[ind_begin + j].n_groups > 0 && !(flags & CNF2_NO_TIES)) || (flags & CNF2_FLUSH_TINY);
                if (tied != (pass == 1)) continue;
                Job jb;
                jb.ind   = j;
                jb.first = ctx->chromstarts[c];
                jb.last  = ctx->chromstarts[c + 1] - 1;
                jb.chrom = c;
                jobs.push_back(jb);
            }
        if (pass == 0) n_fast = jobs.size();
    }
    const size_t n_general = jobs.size() - n_fast;
    rc = ensure(ctx, &ctx->d_jobs, &ctx->jobs_cap, jobs.size() + 1);
    if (rc) return rc;
    if (!jobs.empty())
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_jobs, jobs.data(), sizeof(Job) * jobs.size(), hipMemcpyHostToDevice, ctx->stream));
    if (n_packed > 0) {
        rc = ensure(ctx, &ctx->d_pjobs, &ctx->pjobs_cap, n_packed);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_pjobs, pjobs.data(), sizeof(PackedJob) * n_packed, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // the job vectors go out of scope

    // grids: one wave per job in flight, capped at what is resident so that the spill stays small
    auto grid_for = [&](size_t nj, int per_cu) {
        int g   = (int)((nj + CNF2_WAVES_PER_BLOCK - 1) / CNF2_WAVES_PER_BLOCK);
        int cap = ctx->n_cu * per_cu - ctx->reserve_blocks;
        if (cap < 1) cap = 1;
        return g > cap ? cap : g;
    };
    // the packed kernel runs before the ordinary fast kernel on the same stream and in the same spill slots
    int grid_fast = grid_for(n_fast > n_packed ? n_fast : n_packed, ctx->fast_blocks_per_cu);
    int grid_gen  = grid_for(n_general, ctx->blocks_per_cu);
    const size_t stride = (size_t)max_chrom_len(ctx) * 528;   // covers every layout: 520 or 528 doubles per (pair of) marker(s), 512 in the general kernel
    {
        // One spill slot per resident wave.  Long chromosomes make slots big: keep the spill within
        // ~60 % of what is free (plus what the context already holds) by running fewer waves.
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        const size_t budget   = (size_t)((double)(free_b + ctx->spill_bytes) * 0.6);
        const size_t per_blk  = (size_t)CNF2_WAVES_PER_BLOCK * stride * sizeof(double);
        const size_t max_blks = budget / per_blk;
        if (max_blks < 1)
            return fail(ctx, CNF2_ERR_NOMEM, "a chromosome of %d markers needs %zu MB of spill per block, %zu MB free",
                        max_chrom_len(ctx), per_blk >> 20, free_b >> 20);
        if ((size_t)grid_fast > max_blks) grid_fast = (int)max_blks;
        if ((size_t)grid_gen > max_blks) grid_gen = (int)max_blks;
    }
    if ((n_fast > 0 || n_packed > 0) && n_general > 0) {
        // both kernels run side by side on two streams with disjoint spill slots: share the budget
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        const size_t budget   = (size_t)((double)(free_b + ctx->spill_bytes) * 0.6);
        const size_t per_blk  = (size_t)CNF2_WAVES_PER_BLOCK * stride * sizeof(double);
        while ((size_t)(grid_fast + grid_gen) * per_blk > budget && grid_fast + grid_gen > 2) {
            if (grid_fast > 1) grid_fast--;
            if (grid_gen > 1 && (size_t)(grid_fast + grid_gen) * per_blk > budget) grid_gen--;
        }
    }
    const int    grid = grid_fast + grid_gen;
    const size_t need = (size_t)grid * CNF2_WAVES_PER_BLOCK * stride;
    {
        size_t capd = ctx->spill_bytes / sizeof(double);
        rc = ensure(ctx, &ctx->d_spill, &capd, need);
        ctx->spill_bytes = capd * sizeof(double);
        if (rc) return rc;
    }

    double *d_f, *d_l, *d_d = nullptr;
    const size_t nf = (size_t)n * ctx->n_chrom * 8, nl = (size_t)n * ctx->n_chrom, nd = (size_t)n * ctx->n_markers * 3;
    if (flags & CNF2_OUT_DEVICE) {
        d_f = factors_out;
        d_l = loglik_out;
        d_d = dosage_out;
    } else {
        if ((rc = ensure(ctx, &ctx->d_factors, &ctx->factors_cap, nf))) return rc;
        if ((rc = ensure(ctx, &ctx->d_loglik, &ctx->loglik_cap, nl))) return rc;
        if (want_dosage && (rc = ensure(ctx, &ctx->d_dosage, &ctx->dosage_cap, nd))) return rc;
        d_f = ctx->d_factors;
        d_l = ctx->d_loglik;
        d_d = ctx->d_dosage;
    }

    KernelParams p;
    base_params(ctx, &p);
    p.windows      = ctx->d_windows + ind_begin;
    p.jobs         = ctx->d_jobs;
    p.n_jobs       = (int)n_fast;
    p.spill        = ctx->d_spill;
    p.spill_stride = stride;
    p.factors      = d_f;
    p.loglik       = d_l;
    p.dosage       = d_d;
    p.flags        = (want_dosage ? 0 : KP_NO_DOSAGE) | ((flags & CNF2_RAW_DOSAGE) ? KP_RAW_DOSAGE : 0) |
              ((flags & CNF2_NO_TIES) ? KP_NO_TIES : 0);
    if (flags & CNF2_STATIC_JOBS) p.job_next = nullptr;
    if (flags & CNF2_LOG_PATHS) {
        if ((rc = ensure(ctx, &ctx->d_pathlog, &ctx->pathlog_cap, nl))) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_pathlog, 0xff, nl * sizeof(int32_t), ctx->stream));
        p.path_log     = ctx->d_pathlog;
        ctx->pathlog_n = (int)nl;
    }

    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    if (n_general > 0) {
        // the tied windows' kernel on a second stream with its own spill slots (behind the fast kernel's) and its own job
        // counter: no ordering between the two kernels.  It is launched FIRST: its jobs are the long ones (a backward pass
        // per tie combination), and blocks of either kernel that find no room wait and take jobs from their launch's counter
        // once they get on -- the sweep then ends on the short jobs of the untied windows
        KernelParams pt = p;
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev0, 0));
        pt.jobs   = ctx->d_jobs + n_fast;
        pt.n_jobs = (int)n_general;
        if (pt.job_next) pt.job_next = ctx->d_jobnext + 1;
        pt.spill  = ctx->d_spill + (size_t)((n_fast > 0 || n_packed > 0) ? grid_fast : 0) * CNF2_WAVES_PER_BLOCK * stride;
        // the tile-producer kernel with a pass per tie combination; the general kernel (one lane per table entry, per-marker
        // producer) with the full spill and where asked for
        if (flags & CNF2_FLUSH_TINY) pt.flags |= KP_FLUSH_TINY;
        if ((flags & CNF2_FULL_SPILL) || (flags & (CNF2_TIES_GENERAL | CNF2_FLUSH_TINY))) launch_fb(pt, grid_gen, false, ctx->stream2);
        else launch_fb_fast_tied(pt, grid_gen, ctx->stream2);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ctx->ev2, ctx->stream2));
    }
    if (n_packed > 0) {
        p.pjobs   = ctx->d_pjobs;
        p.n_pjobs = (int)n_packed;
        int gp    = (int)((n_packed + CNF2_WAVES_PER_BLOCK - 1) / CNF2_WAVES_PER_BLOCK);
        KernelParams pp = p;
        if (pp.job_next) pp.job_next = ctx->d_jobnext + 2;    // (the fast kernel behind it on the stream zeroes its own)
        launch_fb_packed(pp, gp < grid_fast ? gp : grid_fast, ctx->stream);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (n_fast > 0) {
        int gf = (int)((n_fast + CNF2_WAVES_PER_BLOCK - 1) / CNF2_WAVES_PER_BLOCK);
        p.clock_out = ctx->d_clock;
        if ((flags & CNF2_XPOSE) && !(flags & CNF2_FULL_SPILL)) launch_fb_fast_xpose(p, gf < grid_fast ? gf : grid_fast, ctx->stream);
        else launch_fb_fast(p, gf < grid_fast ? gf : grid_fast, !(flags & CNF2_FULL_SPILL), ctx->stream);
        p.clock_out = nullptr;
        HIP_TRY(ctx, hipGetLastError());
    }
    if (n_general > 0) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev2, 0));
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;

    if (!(flags & CNF2_OUT_DEVICE)) {
        HIP_TRY(ctx, hipMemcpyAsync(factors_out, d_f, nf * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(loglik_out, d_l, nl * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (want_dosage)
            HIP_TRY(ctx, hipMemcpyAsync(dosage_out, d_d, nd * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CNF2_OK;
}

int cnf2_last_paths(cnf2_ctx* ctx, int32_t* paths_out, int n)
{
    if (!ctx || !paths_out || n < 0) return fail(ctx, CNF2_ERR_ARG, "bad path arguments");
    if (!ctx->d_pathlog || n > ctx->pathlog_n) return fail(ctx, CNF2_ERR_STATE, "no sweep with CNF2_LOG_PATHS covers %d jobs", n);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(paths_out, ctx->d_pathlog, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return CNF2_OK;
}

int cnf2_last_kernel_ms(cnf2_ctx* ctx, float* kernel_ms, int n)
{
    if (!ctx || !kernel_ms || n < 1) return CNF2_ERR_ARG;
    if (!ctx->timed) return fail(ctx, CNF2_ERR_STATE, "no sweep has been launched");
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    kernel_ms[0] = ms;
    for (int i = 1; i < n; i++) kernel_ms[i] = 0;
    return CNF2_OK;
}

int cnf2_sweep_clock(cnf2_ctx* ctx, double* mhz_out)
{
    if (!ctx || !mhz_out) return fail(ctx, CNF2_ERR_ARG, "bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *mhz_out = 0.0;
    if (!ctx->d_clock) return CNF2_OK;
    unsigned long long t[2] = {0, 0};
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(t, ctx->d_clock, sizeof(t), hipMemcpyDeviceToHost));
    int khz = 0;
    HIP_TRY(ctx, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    if (t[1] > 0 && khz > 0) *mhz_out = (double)t[0] / (double)t[1] * (double)khz * 1e-3;
    return CNF2_OK;
}

int cnf2_clock_probe(cnf2_ctx* ctx, double* mhz_out)
{
    if (!ctx || !mhz_out) return fail(ctx, CNF2_ERR_ARG, "bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, &ctx->d_scratch, &ctx->scratch_cap, (size_t)8);
    if (rc) return rc;
    const int iters = 1 << 20;
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    launch_clock_probe(ctx->n_cu, 1 << 14, ctx->d_scratch, ctx->stream);            // warm up
    HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    launch_clock_probe(ctx->n_cu, iters, ctx->d_scratch, ctx->stream);
    HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // per SIMD: 4 waves x 8 FMAs x iters, one wave-wide f64 FMA per 4 cycles
    *mhz_out = 4.0 * 8.0 * (double)iters * 4.0 / ((double)ms * 1e-3) / 1e6;
    return CNF2_OK;
}

size_t cnf2_workspace_bytes(cnf2_ctx* ctx)
{
    if (!ctx) return 0;
    return ctx->spill_bytes + ctx->jobs_cap * sizeof(Job) +
           (ctx->factors_cap + ctx->loglik_cap + ctx->dosage_cap + ctx->scratch_cap + ctx->wbuf_cap) * sizeof(double);
}

// Runs fb_kernel<true> for one individual x chromosome and leaves the reference-layout store in the
// context's scratch buffer; fills the Stage2Params view of it.  extra = doubles reserved after it.
static int run_store(cnf2_ctx* ctx, int ind, int chrom, Stage2Params* q, size_t extra, double** extra_ptr)
{
    int rc = ready(ctx);
    if (rc) return rc;
    if (ind < 0 || ind >= (int)ctx->windows.size() || chrom < 0 || chrom >= ctx->n_chrom)
        return fail(ctx, CNF2_ERR_ARG, "individual or chromosome out of range");
    const int    first = ctx->chromstarts[chrom], last = ctx->chromstarts[chrom + 1] - 1, len = last - first + 1;
    const size_t nfw = (size_t)8 * len * 3 * 64, nff = (size_t)8 * len * 3;
    const size_t stride = (size_t)len * 512;
    // scratch: fwbw | factors | spill(4 waves) | out factors(8) | loglik(8) | dosage(n_markers*3) | job(8) | extra
    const size_t total = nfw + nff + stride * CNF2_WAVES_PER_BLOCK + 16 + (size_t)ctx->n_markers * 3 + 8 + extra;
    if ((rc = ensure(ctx, &ctx->d_scratch, &ctx->scratch_cap, total))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_scratch, 0, (nfw + nff) * sizeof(double), ctx->stream));
    double* d_fw  = ctx->d_scratch;
    double* d_ff  = d_fw + nfw;
    double* d_sp  = d_ff + nff;
    double* d_f   = d_sp + stride * CNF2_WAVES_PER_BLOCK;
    double* d_l   = d_f + 8;
    double* d_d   = d_l + 8;
    Job*    d_job = (Job*)(d_d + (size_t)ctx->n_markers * 3);
    if (extra_ptr) *extra_ptr = d_d + (size_t)ctx->n_markers * 3 + 8;
    Job     jb;
    jb.ind = 0;
    jb.first = first;
    jb.last = last;
    jb.chrom = 0;
    HIP_TRY(ctx, hipMemcpyAsync(d_job, &jb, sizeof(jb), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    KernelParams p;
    base_params(ctx, &p);
    p.windows      = ctx->d_windows + ind;
    p.n_chrom      = 1;
    p.jobs         = d_job;
    p.n_jobs       = 1;
    p.spill        = d_sp;
    p.spill_stride = stride;
    p.factors      = d_f;
    p.loglik       = d_l;
    p.dosage       = d_d;   // rows of this individual, indexed by global marker
    p.flags        = KP_NO_DOSAGE;
    p.dbg_fwbw     = d_fw;
    p.dbg_factors  = d_ff;
    launch_fb(p, 1, true, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    q->kp          = p;
    q->fwbw        = d_fw;
    q->fwbwfactors = d_ff;
    q->factors     = d_f;
    q->loglik      = d_l;
    q->first       = first;
    q->len         = len;
    return CNF2_OK;
}

int cnf2_fwbw_store(cnf2_ctx* ctx, int ind, int chrom, double* fwbw_out, double* fwbwfactors_out)
{
    if (!ctx || !fwbw_out || !fwbwfactors_out) return fail(ctx, CNF2_ERR_ARG, "bad fwbw_store arguments");
    Stage2Params q;
    int rc = run_store(ctx, ind, chrom, &q, 0, nullptr);
    if (rc) return rc;
    const size_t nfw = (size_t)8 * q.len * 3 * 64, nff = (size_t)8 * q.len * 3;
    HIP_TRY(ctx, hipMemcpyAsync(fwbw_out, q.fwbw, nfw * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(fwbwfactors_out, q.fwbwfactors, nff * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_locked_query(cnf2_ctx* ctx, int ind, int chrom, int marker, double* val_out)
{
    if (!ctx || !val_out) return fail(ctx, CNF2_ERR_ARG, "bad locked_query arguments");
    Stage2Params q;
    double*      d_out = nullptr;
    const size_t n = (size_t)8 * 64 * 128;
    int rc = run_store(ctx, ind, chrom, &q, n, &d_out);
    if (rc) return rc;
    if (marker < q.first || marker >= q.first + q.len) return fail(ctx, CNF2_ERR_ARG, "marker not on this chromosome");
    launch_locked_query(q, marker, d_out, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(val_out, d_out, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_turn_scan(cnf2_ctx* ctx, int ind, int chrom, int marker, double* rawervals_out)
{
    if (!ctx || !rawervals_out) return fail(ctx, CNF2_ERR_ARG, "bad turn_scan arguments");
    Stage2Params q;
    double*      d_out = nullptr;
    const size_t n = 128 * 8;
    int rc = run_store(ctx, ind, chrom, &q, n, &d_out);
    if (rc) return rc;
    if (marker < q.first || marker >= q.first + q.len) return fail(ctx, CNF2_ERR_ARG, "marker not on this chromosome");
    launch_turn_scan(q, marker, d_out, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(rawervals_out, d_out, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_turn_scan_rows(cnf2_ctx* ctx, int ind, int chrom, double* rows_out)
{
    if (!ctx || !rows_out) return fail(ctx, CNF2_ERR_ARG, "bad turn_scan_rows arguments");
    Stage2Params q;
    double*      d_out = nullptr;
    int rc = ready(ctx);
    if (rc) return rc;
    if (chrom < 0 || chrom >= ctx->n_chrom) return fail(ctx, CNF2_ERR_ARG, "chromosome out of range");
    const size_t n = (size_t)(ctx->chromstarts[chrom + 1] - ctx->chromstarts[chrom]) * 1024;
    rc = run_store(ctx, ind, chrom, &q, n, &d_out);
    if (rc) return rc;
    launch_turn_scan_rows(q, d_out, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(rows_out, d_out, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_state_posterior(cnf2_ctx* ctx, int ind, int chrom, double* rows_out, uint32_t flags)
{
    if (!ctx || !rows_out) return fail(ctx, CNF2_ERR_ARG, "bad state_posterior arguments");
    Stage2Params q;
    double*      d_out = nullptr;
    int rc = ready(ctx);
    if (rc) return rc;
    if (chrom < 0 || chrom >= ctx->n_chrom) return fail(ctx, CNF2_ERR_ARG, "chromosome out of range");
    const size_t n = (size_t)(ctx->chromstarts[chrom + 1] - ctx->chromstarts[chrom]) * 64;
    rc = run_store(ctx, ind, chrom, &q, n, &d_out);
    if (rc) return rc;
    launch_state_rows(q, (flags & CNF2_NO_TIES) ? KP_NO_TIES : 0, d_out, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(rows_out, d_out, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_haplos(cnf2_ctx* ctx, int ind, int chrom, double* rows_out, uint32_t flags)
{
    if (!ctx || !rows_out) return fail(ctx, CNF2_ERR_ARG, "bad haplos arguments");
    Stage2Params q;
    double*      d_out = nullptr;
    int rc = ready(ctx);
    if (rc) return rc;
    if (chrom < 0 || chrom >= ctx->n_chrom) return fail(ctx, CNF2_ERR_ARG, "chromosome out of range");
    const size_t n = (size_t)(ctx->chromstarts[chrom + 1] - ctx->chromstarts[chrom]) * 14;
    rc = run_store(ctx, ind, chrom, &q, n, &d_out);
    if (rc) return rc;
    launch_haplos_rows(q, (flags & CNF2_NO_TIES) ? KP_NO_TIES : 0, d_out, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(rows_out, d_out, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_infprobs(cnf2_ctx* ctx, int ind, int chrom, int marker, double* inf_out, double* hz_out, uint32_t flags)
{
    if (!ctx || !inf_out || !hz_out) return fail(ctx, CNF2_ERR_ARG, "bad infprobs arguments");
    Stage2Params q;
    double*      d_out = nullptr;
    int rc = ready(ctx);
    if (rc) return rc;
    if (chrom < 0 || chrom >= ctx->n_chrom) return fail(ctx, CNF2_ERR_ARG, "chromosome out of range");
    if (marker < ctx->chromstarts[chrom] || marker >= ctx->chromstarts[chrom + 1])
        return fail(ctx, CNF2_ERR_ARG, "marker not on the chromosome");
    rc = run_store(ctx, ind, chrom, &q, 32, &d_out);
    if (rc) return rc;
    launch_infprobs(q, marker, (flags & CNF2_NO_TIES) ? KP_NO_TIES : 0, d_out, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    double h[30];
    HIP_TRY(ctx, hipMemcpyAsync(h, d_out, 30 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 28; k++) inf_out[k] = h[k];
    hz_out[0] = h[28];
    hz_out[1] = h[29];
    return CNF2_OK;
}

int cnf2_infprobs_rows(cnf2_ctx* ctx, int ind, int chrom, double* rows_out, uint32_t flags)
{
    if (!ctx || !rows_out) return fail(ctx, CNF2_ERR_ARG, "bad infprobs_rows arguments");
    Stage2Params q;
    double*      d_out = nullptr;
    int rc = ready(ctx);
    if (rc) return rc;
    if (chrom < 0 || chrom >= ctx->n_chrom) return fail(ctx, CNF2_ERR_ARG, "chromosome out of range");
    const size_t n = (size_t)(ctx->chromstarts[chrom + 1] - ctx->chromstarts[chrom]) * 30;
    rc = run_store(ctx, ind, chrom, &q, n, &d_out);
    if (rc) return rc;
    launch_infprobs_rows(q, (flags & CNF2_NO_TIES) ? KP_NO_TIES : 0, d_out, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(rows_out, d_out, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_descendants(cnf2_ctx* ctx, int32_t* desc_out)
{
    if (!ctx || !desc_out) return fail(ctx, CNF2_ERR_ARG, "bad descendants arguments");
    if (ctx->ped.n_rec == 0) return fail(ctx, CNF2_ERR_STATE, "no pedigree uploaded");
    derive_descendants(ctx->ped, desc_out);
    return CNF2_OK;
}

// Entries of the list the update scouts set flows aside on: a chunk of the pass's flows, at most 2^26
static size_t todo_chunk(size_t n_rec, size_t chrom_len, size_t markers_upto)
{
    const size_t n1 = n_rec * chrom_len * 4, n3 = n_rec * markers_upto;
    size_t       want = n1 > n3 ? n1 : n3;
    if (want < 4096) want = 4096;
    const size_t cap = (size_t)1 << 26;        // x 48 B for the two lists: 3.2 GB
    return want < cap ? want : cap;
}

// ... in doubles: two lists of 24-byte entries (the scouts' and the packed one of the guided kernel; a third for the experiment with
// a second lock-step kernel) and the packing's counts
#ifdef CNF2_X_GUIDED_ROUNDS
#define TODO_LISTS 3
#else
#define TODO_LISTS 2
#endif
static size_t todo_doubles(size_t chunk) { return chunk * 3 * TODO_LISTS + chunk / 256 + 8; }

enum : uint32_t { ACC_RESERVE_ONLY = 1u << 31 };     // internal flag of cnf2_sweep_accumulate (not in the header)

// Batched HOT LOOP 2 with its reductions (cnF2freq.cpp:5416-5577, 5876-5902 with moveinfprobs / movehaplos
// 3577-3616) for the analysed individuals [ind_begin, ind_end): the sweep kernels run in their accumulate
// instantiation (they leave the posterior weights wg(s, g) of every marker in a batch buffer next to the usual
// outputs), acc_rows_kernel turns them into the per-record accumulators on the device.  Jobs go in batches sized
// to the memory that is free (4 KB per individual x marker of weights).
int cnf2_sweep_accumulate(cnf2_ctx* ctx, int ind_begin, int ind_end, const int32_t* descendants, double* factors_out,
                          double* loglik_out, double* dosage_out, double* infprobs, double* haplobase,
                          double* haplocount, double* homozyg, uint32_t flags)
{
    int rc = ready(ctx);
    if (rc) return rc;
    const int n_all = (int)ctx->windows.size();
    const bool out_dev = (flags & CNF2_OUT_DEVICE) != 0, acc_dev = (flags & CNF2_ACC_DEVICE) != 0;
    const bool acc_given = infprobs && haplobase && haplocount && homozyg;
    const bool acc_none = !infprobs && !haplobase && !haplocount && !homozyg;    // keep them in the context
    const bool reserve_only = (flags & ACC_RESERVE_ONLY) != 0;      // cnf2_reserve_accumulate: allocations only
    if ((!descendants && !reserve_only) || (!acc_given && !(acc_none && !acc_dev)) || ind_begin < 0 || ind_end > n_all || ind_begin > ind_end)
        return fail(ctx, CNF2_ERR_ARG, "bad accumulate arguments");
    if (out_dev && (!factors_out || !loglik_out || !dosage_out)) return fail(ctx, CNF2_ERR_ARG, "output pointer is NULL");
    const HostPedigree& P = ctx->ped;
    const int    n = ind_end - ind_begin;
    const size_t M = (size_t)ctx->n_markers, R = (size_t)P.n_rec;
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // per-record tables
    if (ctx->rec_cap < R) {
        if (ctx->d_desc) HIP_TRY(ctx, hipFree(ctx->d_desc));
        if (ctx->d_rec_empty) HIP_TRY(ctx, hipFree(ctx->d_rec_empty));
        ctx->d_desc = nullptr;
        ctx->d_rec_empty = nullptr;
        ctx->rec_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_desc, sizeof(int32_t) * R));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_rec_empty, R));
        ctx->rec_cap = R;
    }
    if (descendants) HIP_TRY(ctx, hipMemcpy(ctx->d_desc, descendants, sizeof(int32_t) * R, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_rec_empty, P.empty.data(), R, hipMemcpyHostToDevice));

    // accumulators and sweep outputs
    double *a_inf = infprobs, *a_hb = haplobase, *a_hc = haplocount, *a_hz = homozyg;
    if (!acc_dev) {
        if ((rc = ensure(ctx, &ctx->d_acc_inf, &ctx->acc_inf_cap, R * M * 4))) return rc;
        if ((rc = ensure(ctx, &ctx->d_acc_hb, &ctx->acc_hb_cap, R * M))) return rc;
        if ((rc = ensure(ctx, &ctx->d_acc_hc, &ctx->acc_hc_cap, R * M))) return rc;
        if ((rc = ensure(ctx, &ctx->d_acc_hz, &ctx->acc_hz_cap, (size_t)(n > 0 ? n : 1) * M * 2))) return rc;
        a_inf = ctx->d_acc_inf;
        a_hb  = ctx->d_acc_hb;
        a_hc  = ctx->d_acc_hc;
        a_hz  = ctx->d_acc_hz;
    }
    if (!(flags & CNF2_ACC_KEEP)) {
        HIP_TRY(ctx, hipMemsetAsync(a_inf, 0, R * M * 4 * sizeof(double), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(a_hb, 0, R * M * sizeof(double), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(a_hc, 0, R * M * sizeof(double), ctx->stream));
    }
    HIP_TRY(ctx, hipMemsetAsync(a_hz, 0, (size_t)n * M * 2 * sizeof(double), ctx->stream));
    const size_t nf = (size_t)n * ctx->n_chrom * 8, nl = (size_t)n * ctx->n_chrom, nd = (size_t)n * M * 3;
    double *d_f = factors_out, *d_l = loglik_out, *d_d = dosage_out;
    if (!out_dev) {
        if ((rc = ensure(ctx, &ctx->d_factors, &ctx->factors_cap, nf ? nf : 1))) return rc;
        if ((rc = ensure(ctx, &ctx->d_loglik, &ctx->loglik_cap, nl ? nl : 1))) return rc;
        if ((rc = ensure(ctx, &ctx->d_dosage, &ctx->dosage_cap, nd ? nd : 1))) return rc;
        d_f = ctx->d_factors;
        d_l = ctx->d_loglik;
        d_d = ctx->d_dosage;
    }
    // A caller that passes no dosage pointer gets no per-locus rows (an iteration that prints none: all but the last of a
    // run): the sweep then runs in the instantiation that forms none -- no class sums, no restricted tables, no tile
    // epilogue -- and, since the posterior weights do not see the tie rule, the sweep of the windows with tie groups takes it
    // as well (their accumulators do see the rule: they stay a pass of their own).
    const bool want_rows = dosage_out != nullptr;
    if (n > 0) {
        // job list: untied windows (fast kernel) first, tied ones (general kernel) after
        std::vector<Job> jobs;
        size_t           n_fast = 0;
        for (int pass = 0; pass < 2; pass++) {
            for (int c : chrom_order(ctx))
                for (int j = 0; j < n; j++) {
                    const bool tied = ctx->windows[ind_begin + j].n_groups > 0 && !(flags & CNF2_NO_TIES);
                    if (tied != (pass == 1)) continue;
                    Job jb;
                    jb.ind = j;
                    jb.first = ctx->chromstarts[c];
                    jb.last = ctx->chromstarts[c + 1] - 1;
                    jb.chrom = c;
                    jobs.push_back(jb);
                }
            if (pass == 0) n_fast = jobs.size();
        }
        if ((rc = ensure(ctx, &ctx->d_jobs, &ctx->jobs_cap, jobs.size() + 1))) return rc;
        HIP_TRY(ctx, hipMemcpy(ctx->d_jobs, jobs.data(), sizeof(Job) * jobs.size(), hipMemcpyHostToDevice));

        const int    mlen = max_chrom_len(ctx);
        const size_t stride = (size_t)mlen * 528;
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        free_b += ctx->spill_bytes + ctx->wbuf_cap * sizeof(double);
        // spill: one slot per resident wave, at most a quarter of what is free
        int          grid_cap = ctx->n_cu * ctx->fast_blocks_per_cu - ctx->reserve_blocks;
        if (grid_cap < 1) grid_cap = 1;
        const size_t per_blk = (size_t)CNF2_WAVES_PER_BLOCK * stride * sizeof(double);
        if ((size_t)grid_cap * per_blk > free_b / 4) grid_cap = (int)(free_b / 4 / per_blk);
        if (grid_cap < 1) return fail(ctx, CNF2_ERR_NOMEM, "not enough memory for the spill of one block");
        {
            size_t capd = ctx->spill_bytes / sizeof(double);
            rc = ensure(ctx, &ctx->d_spill, &capd, (size_t)grid_cap * CNF2_WAVES_PER_BLOCK * stride);
            ctx->spill_bytes = capd * sizeof(double);
            if (rc) return rc;
        }
        // CNF2_DETERMINISTIC: the per-individual rows (336 B per individual x marker) are taken out of what is free BEFORE the
        // batch buffer is sized -- and allocated now, also by cnf2_reserve_accumulate -- so that the batch buffer cannot
        // leave them without memory
        const size_t part_need = (flags & CNF2_DETERMINISTIC) ? (size_t)n * M * 42 : 0;
        if (part_need) {
            const size_t have = ctx->part_cap * sizeof(double);
            if (part_need * sizeof(double) > free_b / 2 + have)
                return fail(ctx, CNF2_ERR_NOMEM, "CNF2_DETERMINISTIC needs %zu MB for the per-individual rows", (part_need * 8) >> 20);
            if ((rc = ensure(ctx, &ctx->d_part, &ctx->part_cap, part_need))) return rc;
            free_b -= (part_need * sizeof(double) > have) ? part_need * sizeof(double) - have : 0;
        }
        // weights: 512 doubles per (job, marker); batch = what fits in half of the rest
        const size_t per_job = (size_t)mlen * 512;
        size_t       batch = (free_b - (size_t)grid_cap * per_blk) / 2 / (per_job * sizeof(double));
        if (batch < 1) return fail(ctx, CNF2_ERR_NOMEM, "not enough memory for the weights of one job (%zu MB)", per_job >> 17);
        if (batch > jobs.size()) batch = jobs.size();
        if (batch > 1000000) batch = 1000000;
        batch = whole_rounds(batch, jobs.size(), grid_cap);
        if (ctx->batch_jobs > 0 && batch > (size_t)ctx->batch_jobs) batch = (size_t)ctx->batch_jobs;
        {
            // CNF2_TIMING: the first call of a run allocates the batch buffer (up to half the free memory) -- seconds
            const bool timing = getenv("CNF2_TIMING") != nullptr && ctx->wbuf_cap < batch * per_job;
            const auto t0 = std::chrono::steady_clock::now();
            if ((rc = ensure(ctx, &ctx->d_wbuf, &ctx->wbuf_cap, batch * per_job))) return rc;
            if (timing)
                fprintf(stderr, "  [sweep_accumulate] batch buffer of %.1f GB allocated in %.3f s (%zu jobs per batch of %zu)\n",
                        batch * per_job * 8 / 1e9, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), batch,
                        jobs.size());
        }
        if (reserve_only) {
            // ... and the buffers of the update passes (cnf2_update_pass): results of a chromosome's flows, the scouts' list
            if (!ctx->d_flow_next) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_flow_next, 32 * sizeof(unsigned long long)));
            if ((rc = ensure(ctx, &ctx->d_flow_out, &ctx->flow_out_cap, R * (size_t)mlen * 4))) return rc;
            if ((rc = ensure(ctx, &ctx->d_todo, &ctx->todo_cap, todo_doubles(todo_chunk(R, (size_t)mlen, M))))) return rc;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            return CNF2_OK;
        }

        KernelParams p;
        base_params(ctx, &p);
        if (flags & CNF2_STATIC_JOBS) p.job_next = nullptr;
        p.windows      = ctx->d_windows + ind_begin;
        p.spill        = ctx->d_spill;
        p.spill_stride = stride;
        p.factors      = d_f;
        p.loglik       = d_l;
        p.dosage       = d_d;
        p.flags        = ((flags & CNF2_RAW_DOSAGE) ? KP_RAW_DOSAGE : 0) | ((flags & CNF2_NO_TIES) ? KP_NO_TIES : 0);
        p.wbuf         = ctx->d_wbuf;
        p.wstride      = (size_t)mlen;
        AccParams q;
        memset(&q, 0, sizeof(q));
        q.flags     = ((flags & CNF2_NO_TIES) ? KP_NO_TIES : 0) | ((flags & CNF2_ACC_TABLE) ? KP_ACC_TABLE : 0);
        for (int j = 0; j < n; j++)
            if (ctx->windows[ind_begin + j].flags[0] & SLOT_FOUNDER) q.flags |= KP_ACC_ATTOP;
        q.slot_rec  = ctx->d_slot_rec + (size_t)ind_begin * 7;
        q.desc      = ctx->d_desc;
        q.rec_empty = ctx->d_rec_empty;
        q.acc_inf   = a_inf;
        q.acc_hb    = a_hb;
        q.acc_hc    = a_hc;
        q.acc_hz    = a_hz;
        q.max_len   = mlen;
        std::vector<int32_t> gather;               // CNF2_DETERMINISTIC: rec_start[R + 1], then ind * 8 + slot per record
        if (flags & CNF2_DETERMINISTIC) {
            const size_t need = part_need;                    // allocated above, before the batch buffer was sized
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_part, 0, need * sizeof(double), ctx->stream));
            q.part = ctx->d_part;
            std::vector<int32_t> count(R + 1, 0);
            auto first_slots = [&](int j, auto&& f) {         // the slots that emit: first occurrence of their record
                const int32_t* sr = &ctx->slot_rec[(size_t)(ind_begin + j) * 7];
                for (int k = 0; k < 7; k++) {
                    if (sr[k] < 0) continue;
                    bool first = true;
                    for (int k2 = 0; k2 < k; k2++) first = first && sr[k2] != sr[k];
                    if (first) f(sr[k], k);
                }
            };
            for (int j = 0; j < n; j++) first_slots(j, [&](int r, int) { count[r + 1]++; });
            for (size_t r = 0; r < R; r++) count[r + 1] += count[r];
            gather.assign(R + 1 + (size_t)count[R], 0);
            std::copy(count.begin(), count.end(), gather.begin());
            std::vector<int32_t> fill(count.begin(), count.end() - 1);
            for (int j = 0; j < n; j++) first_slots(j, [&](int r, int k) { gather[R + 1 + fill[r]++] = j * 8 + k; });
            if ((rc = ensure(ctx, &ctx->d_gather, &ctx->gather_cap, gather.size()))) return rc;
            HIP_TRY(ctx, hipMemcpyAsync(ctx->d_gather, gather.data(), gather.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        for (int pass = 0; pass < 2; pass++) {
            const size_t lo = pass ? n_fast : 0, hi = pass ? jobs.size() : n_fast;
            for (size_t b0 = lo; b0 < hi; b0 += batch) {
                const size_t nb = (hi - b0 < batch) ? hi - b0 : batch;
                p.jobs   = ctx->d_jobs + b0;
                p.n_jobs = (int)nb;
                int grid = (int)((nb + CNF2_WAVES_PER_BLOCK - 1) / CNF2_WAVES_PER_BLOCK);
                if (grid > grid_cap) grid = grid_cap;
                if (pass == 0) launch_fb_fast_w(p, grid, ctx->stream, want_rows);
                else if (flags & CNF2_TIES_GENERAL) launch_fb_w(p, grid, ctx->stream);
                else if (!want_rows) launch_fb_fast_w(p, grid, ctx->stream, false);
                else launch_fb_fast_tied_w(p, grid, ctx->stream);
                HIP_TRY(ctx, hipGetLastError());
                q.kp     = p;
                q.n_jobs = (int)nb;
                q.max_len = jobs[b0].last - jobs[b0].first + 1;      // the batch's longest job is its first (chrom_order)
                q.flags  = (q.flags & ~(uint32_t)KP_ACC_LANES) | ((pass == 1 || (flags & CNF2_ACC_LANES)) ? KP_ACC_LANES : 0);
                launch_acc_rows(q, ctx->stream);
                HIP_TRY(ctx, hipGetLastError());
            }
        }
        if (q.part) {
            launch_acc_gather(q, ctx->d_gather, ctx->d_gather + R + 1, (int)R, ctx->stream);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // `gather` (host staging of the lists) goes out of scope
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        ctx->timed = true;
    }
    if (!out_dev) {
        if (factors_out) HIP_TRY(ctx, hipMemcpyAsync(factors_out, d_f, nf * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (loglik_out) HIP_TRY(ctx, hipMemcpyAsync(loglik_out, d_l, nl * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (dosage_out) HIP_TRY(ctx, hipMemcpyAsync(dosage_out, d_d, nd * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!acc_dev && acc_given) {
        HIP_TRY(ctx, hipMemcpyAsync(infprobs, a_inf, R * M * 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(haplobase, a_hb, R * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(haplocount, a_hc, R * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(homozyg, a_hz, (size_t)n * M * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!out_dev || !acc_dev) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

// The allocations of cnf2_sweep_accumulate for [ind_begin, ind_end) without the sweep: accumulators, outputs, spill and the
// batch buffer of posterior weights (up to half of the free memory: a first hipMalloc of that size takes seconds on a
// fresh device, 4.3 s for 125 GB measured).  A run calls this once after its uploads, so that its first iteration costs
// what the others cost.
int cnf2_reserve_accumulate(cnf2_ctx* ctx, int ind_begin, int ind_end, uint32_t flags)
{
    const uint32_t keep = flags & (CNF2_NO_TIES | CNF2_DETERMINISTIC | CNF2_TIES_GENERAL);
    return cnf2_sweep_accumulate(ctx, ind_begin, ind_end, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                 keep | ACC_RESERVE_ONLY);
}

// Batched turn scan (HOT LOOP 3, cnF2freq.cpp:5686-5752) for the analysed individuals [ind_begin, ind_end): the sweep
// kernels run in their turn-scan instantiation (alpha after emission and beta of every marker, with scales, into a
// batch buffer), turn_rows_kernel forms rawervals[turn][s] for all 128 turns and 8 shift modes of every marker.
int cnf2_sweep_turn_scan(cnf2_ctx* ctx, int ind_begin, int ind_end, double* rawervals_out, double* turn_lse_out,
                         uint32_t flags)
{
    int rc = ready(ctx);
    if (rc) return rc;
    const int n_all = (int)ctx->windows.size();
    if ((!rawervals_out && !turn_lse_out) || ind_begin < 0 || ind_end > n_all || ind_begin > ind_end)
        return fail(ctx, CNF2_ERR_ARG, "bad turn scan arguments");
    const bool out_dev = (flags & CNF2_OUT_DEVICE) != 0;
    const int    n = ind_end - ind_begin;
    const size_t M = (size_t)ctx->n_markers;
    if (n == 0) return CNF2_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nf = (size_t)n * ctx->n_chrom * 8, nl = (size_t)n * ctx->n_chrom, nd = (size_t)n * M * 3;
    if ((rc = ensure(ctx, &ctx->d_factors, &ctx->factors_cap, nf))) return rc;
    if ((rc = ensure(ctx, &ctx->d_loglik, &ctx->loglik_cap, nl))) return rc;
    if ((rc = ensure(ctx, &ctx->d_dosage, &ctx->dosage_cap, nd))) return rc;
    double *d_full = rawervals_out, *d_lse = turn_lse_out;
    if (!out_dev) {
        // staging for the whole range: 9 KB per individual x marker -- callers with large ranges use device buffers
        d_full = d_lse = nullptr;
        const size_t need = (rawervals_out ? (size_t)n * M * 1024 : 0) + (turn_lse_out ? (size_t)n * M * 128 : 0);
        if ((rc = ensure(ctx, &ctx->d_scratch, &ctx->scratch_cap, need))) return rc;
        double* q0 = ctx->d_scratch;
        if (rawervals_out) {
            d_full = q0;
            q0 += (size_t)n * M * 1024;
        }
        if (turn_lse_out) d_lse = q0;
    }
    std::vector<Job> jobs;
    size_t           n_fast = 0;
    for (int pass = 0; pass < 2; pass++) {
        for (int c : chrom_order(ctx))
            for (int j = 0; j < n; j++) {
                const bool tied = ctx->windows[ind_begin + j].n_groups > 0 && !(flags & CNF2_NO_TIES);
                if (tied != (pass == 1)) continue;
                Job jb;
                jb.ind = j;
                jb.first = ctx->chromstarts[c];
                jb.last = ctx->chromstarts[c + 1] - 1;
                jb.chrom = c;
                jobs.push_back(jb);
            }
        if (pass == 0) n_fast = jobs.size();
    }
    if ((rc = ensure(ctx, &ctx->d_jobs, &ctx->jobs_cap, jobs.size() + 1))) return rc;
    HIP_TRY(ctx, hipMemcpy(ctx->d_jobs, jobs.data(), sizeof(Job) * jobs.size(), hipMemcpyHostToDevice));
    const int    mlen = max_chrom_len(ctx);
    const size_t stride = (size_t)mlen * 528;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
    free_b += ctx->spill_bytes + ctx->wbuf_cap * sizeof(double);
    int          grid_cap = ctx->n_cu * ctx->fast_blocks_per_cu - ctx->reserve_blocks;
    if (grid_cap < 1) grid_cap = 1;
    const size_t per_blk = (size_t)CNF2_WAVES_PER_BLOCK * stride * sizeof(double);
    if ((size_t)grid_cap * per_blk > free_b / 4) grid_cap = (int)(free_b / 4 / per_blk);
    if (grid_cap < 1) return fail(ctx, CNF2_ERR_NOMEM, "not enough memory for the spill of one block");
    {
        size_t capd = ctx->spill_bytes / sizeof(double);
        rc = ensure(ctx, &ctx->d_spill, &capd, (size_t)grid_cap * CNF2_WAVES_PER_BLOCK * stride);
        ctx->spill_bytes = capd * sizeof(double);
        if (rc) return rc;
    }
    const size_t per_job = (size_t)mlen * CNF2_TURN_ROW;
    size_t       batch = (free_b - (size_t)grid_cap * per_blk) / 2 / (per_job * sizeof(double));
    if (batch < 1) return fail(ctx, CNF2_ERR_NOMEM, "not enough memory for the alpha/beta rows of one job");
    if (batch > jobs.size()) batch = jobs.size();
    if (batch > 1000000) batch = 1000000;
    batch = whole_rounds(batch, jobs.size(), grid_cap);
    if (ctx->batch_jobs > 0 && batch > (size_t)ctx->batch_jobs) batch = (size_t)ctx->batch_jobs;
    if ((rc = ensure(ctx, &ctx->d_wbuf, &ctx->wbuf_cap, batch * per_job))) return rc;
    KernelParams p;
    base_params(ctx, &p);
    if (flags & CNF2_STATIC_JOBS) p.job_next = nullptr;
    p.windows      = ctx->d_windows + ind_begin;
    p.spill        = ctx->d_spill;
    p.spill_stride = stride;
    p.factors      = ctx->d_factors;
    p.loglik       = ctx->d_loglik;
    p.dosage       = ctx->d_dosage;
    p.flags        = (flags & CNF2_NO_TIES) ? KP_NO_TIES : 0;
    p.wbuf         = ctx->d_wbuf;
    p.wstride      = (size_t)mlen;
    TurnParams q;
    memset(&q, 0, sizeof(q));
    q.max_len   = mlen;
    q.rawervals = d_full;
    q.turn_lse  = d_lse;
    q.valu_form = (flags & CNF2_TURN_VALU) ? 1 : 0;
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int pass = 0; pass < 2; pass++) {
        const size_t lo = pass ? n_fast : 0, hi = pass ? jobs.size() : n_fast;
        for (size_t b0 = lo; b0 < hi; b0 += batch) {
            const size_t nb = (hi - b0 < batch) ? hi - b0 : batch;
            p.jobs   = ctx->d_jobs + b0;
            p.n_jobs = (int)nb;
            int grid = (int)((nb + CNF2_WAVES_PER_BLOCK - 1) / CNF2_WAVES_PER_BLOCK);
            if (grid > grid_cap) grid = grid_cap;
            // alpha and beta do not see the tie rule: tied windows take the tile-producer kernel too (its rows, which
            // would need the rule, go to the context's scratch and are not an output of this call)
            const bool fast = pass == 0 || !(flags & CNF2_TIES_GENERAL);
            if (fast) launch_fb_fast_ab(p, grid, ctx->stream);
            else launch_fb_ab(p, grid, ctx->stream);
            HIP_TRY(ctx, hipGetLastError());
            q.kp     = p;
            q.n_jobs = (int)nb;
            q.max_len = jobs[b0].last - jobs[b0].first + 1;          // the batch's longest job is its first (chrom_order)
            q.scaled_transitions = fast;
            launch_turn_rows(q, ctx->stream);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    if (!out_dev) {
        if (rawervals_out) HIP_TRY(ctx, hipMemcpyAsync(rawervals_out, d_full, (size_t)n * M * 1024 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (turn_lse_out) HIP_TRY(ctx, hipMemcpyAsync(turn_lse_out, d_lse, (size_t)n * M * 128 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CNF2_OK;
}

// The accumulators alone, into host arrays (zeroed first): the form the parity tests use.
int cnf2_accumulate(cnf2_ctx* ctx, int ind_begin, int ind_end, const int32_t* descendants, double* infprobs_out,
                    double* haplobase_out, double* haplocount_out, double* homozyg_out, uint32_t flags)
{
    return cnf2_sweep_accumulate(ctx, ind_begin, ind_end, descendants, nullptr, nullptr, nullptr, infprobs_out,
                                 haplobase_out, haplocount_out, homozyg_out,
                                 flags & ~(uint32_t)(CNF2_OUT_DEVICE | CNF2_ACC_DEVICE | CNF2_ACC_KEEP));
}

// ------------------------------------------------------------------------------------------------
// Per-iteration parameter updates on the device
// ------------------------------------------------------------------------------------------------
int cnf2_snapshot_priors(cnf2_ctx* ctx, const uint8_t* has_prior)
{
    if (!ctx || !has_prior) return fail(ctx, CNF2_ERR_ARG, "bad prior arguments");
    if (!ctx->d_allele8 || ctx->ped.n_rec == 0) return fail(ctx, CNF2_ERR_STATE, "rows and pedigree must be uploaded first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t cnt = (size_t)ctx->n_rows * ctx->n_markers;
    if (ctx->d_prior_allele8) HIP_TRY(ctx, hipFree(ctx->d_prior_allele8));
    if (ctx->d_prior_sure) HIP_TRY(ctx, hipFree(ctx->d_prior_sure));
    if (ctx->d_has_prior) HIP_TRY(ctx, hipFree(ctx->d_has_prior));
    ctx->d_prior_allele8 = nullptr;
    ctx->d_prior_sure = nullptr;
    ctx->d_has_prior = nullptr;
    ctx->priors_set = false;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_prior_allele8, cnt));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_prior_sure, cnt * sizeof(double2)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_has_prior, ctx->ped.n_rec));
    HIP_TRY(ctx, hipMemcpy(ctx->d_prior_allele8, ctx->d_allele8, cnt, hipMemcpyDeviceToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_prior_sure, ctx->d_sure, cnt * sizeof(double2), hipMemcpyDeviceToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_has_prior, has_prior, ctx->ped.n_rec, hipMemcpyHostToDevice));
    ctx->priors_set = true;
    return CNF2_OK;
}

static int update_pass_impl(cnf2_ctx* ctx, int chrom, const int32_t* recs, int n_recs, const int32_t* children,
                            const int32_t* descendants, double* infprobs, double* haplobase, double* haplocount, double scalefactor,
                            double entropyfactor, int* hits_out, uint32_t flags)
{
    if (!ctx || !children || !descendants || !hits_out) return fail(ctx, CNF2_ERR_ARG, "bad update arguments");
    if (recs) {
        if (n_recs < 0) return fail(ctx, CNF2_ERR_ARG, "bad record list");
        for (int i = 0; i < n_recs; i++)
            if (recs[i] < 0 || recs[i] >= ctx->ped.n_rec || (i > 0 && recs[i] <= recs[i - 1]))
                return fail(ctx, CNF2_ERR_ARG, "the record list must be ascending and within the pedigree (position %d)", i);
    }
    if (!ctx->d_allele8 || ctx->ped.n_rec == 0 || !ctx->d_rho) return fail(ctx, CNF2_ERR_STATE, "map, rows and pedigree must be uploaded first");
    if (!ctx->priors_set) return fail(ctx, CNF2_ERR_STATE, "cnf2_snapshot_priors must be called after the rows were uploaded");
    if (chrom < 0 || chrom >= ctx->n_chrom) return fail(ctx, CNF2_ERR_ARG, "chromosome out of range");
    const bool acc_dev = (flags & CNF2_ACC_DEVICE) != 0;
    const bool given = infprobs && haplobase && haplocount;
    if (acc_dev && !given) return fail(ctx, CNF2_ERR_ARG, "accumulator pointers are NULL");
    const HostPedigree& P = ctx->ped;
    const size_t R = (size_t)P.n_rec, M = (size_t)ctx->n_markers;
    // a row that is written must belong to one record.  Row 0 is the shared blank row of a de-duplicated upload: the
    // kernels never write it (a record on it keeps haplotype weight 1/2); every other row has exactly one owner, empty
    // records included (updatehaploweights moves their weights too).
    {
        std::vector<int32_t> owner(ctx->n_rows, -1);
        for (int r = 0; r < P.n_rec; r++) {
            if (P.row_of[r] == 0) {
                if (!P.empty[r]) return fail(ctx, CNF2_ERR_ARG, "record %d has data but sits on the blank row 0", r);
                continue;
            }
            if (owner[P.row_of[r]] >= 0) return fail(ctx, CNF2_ERR_ARG, "records %d and %d share genotype row %d: updates need one row per record (or the blank row 0)", owner[P.row_of[r]], r, P.row_of[r]);
            owner[P.row_of[r]] = r;
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if (ctx->upd_rec_cap < R) {
        if (ctx->d_row_of) HIP_TRY(ctx, hipFree(ctx->d_row_of));
        if (ctx->d_children) HIP_TRY(ctx, hipFree(ctx->d_children));
        ctx->d_row_of = ctx->d_children = nullptr;
        ctx->upd_rec_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_row_of, sizeof(int32_t) * R));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_children, sizeof(int32_t) * R));
        ctx->upd_rec_cap = R;
    }
    if (ctx->rec_cap < R) {
        if (ctx->d_desc) HIP_TRY(ctx, hipFree(ctx->d_desc));
        if (ctx->d_rec_empty) HIP_TRY(ctx, hipFree(ctx->d_rec_empty));
        ctx->d_desc = nullptr;
        ctx->d_rec_empty = nullptr;
        ctx->rec_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_desc, sizeof(int32_t) * R));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_rec_empty, R));
        ctx->rec_cap = R;
    }
    if (!ctx->d_chromstarts) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_chromstarts, sizeof(int32_t) * 65536));
    if (ctx->n_chrom + 1 > 65536) return fail(ctx, CNF2_ERR_ARG, "too many chromosomes");
    if (!ctx->d_hits) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_hits, sizeof(int)));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_row_of, P.row_of.data(), sizeof(int32_t) * R, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_children, children, sizeof(int32_t) * R, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_desc, descendants, sizeof(int32_t) * R, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_rec_empty, P.empty.data(), R, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_chromstarts, ctx->chromstarts.data(), sizeof(int32_t) * (ctx->n_chrom + 1),
                                hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_hits, 0, sizeof(int), ctx->stream));
    if ((rc = ensure(ctx, &ctx->d_anyinfo, &ctx->anyinfo_cap, R * ctx->n_chrom))) return rc;
    if ((rc = ensure(ctx, &ctx->d_fw, &ctx->fw_cap, R * M * 2))) return rc;
    if ((rc = ensure(ctx, &ctx->d_ratio, &ctx->ratio_cap, R * M))) return rc;
    double *a_inf = infprobs, *a_hb = haplobase, *a_hc = haplocount;
    if (!acc_dev) {
        if ((rc = ensure(ctx, &ctx->d_acc_inf, &ctx->acc_inf_cap, R * M * 4))) return rc;
        if ((rc = ensure(ctx, &ctx->d_acc_hb, &ctx->acc_hb_cap, R * M))) return rc;
        if ((rc = ensure(ctx, &ctx->d_acc_hc, &ctx->acc_hc_cap, R * M))) return rc;
        a_inf = ctx->d_acc_inf;
        a_hb  = ctx->d_acc_hb;
        a_hc  = ctx->d_acc_hc;
        if (given) {
            HIP_TRY(ctx, hipMemcpyAsync(a_inf, infprobs, R * M * 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(a_hb, haplobase, R * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(a_hc, haplocount, R * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        }
    }
    UpdateParams u;
    memset(&u, 0, sizeof(u));
    u.n_rec = P.n_rec;
    if (recs) {
        if ((rc = ensure(ctx, &ctx->d_updrecs, &ctx->updrecs_cap, (size_t)(n_recs > 0 ? n_recs : 1)))) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_updrecs, recs, sizeof(int32_t) * n_recs, hipMemcpyHostToDevice, ctx->stream));
        u.n_rec = n_recs;
        u.rec_list = ctx->d_updrecs;
    }
    const size_t RU = (size_t)u.n_rec;         // records this pass updates
    u.n_markers = ctx->n_markers;
    u.n_chrom = ctx->n_chrom;
    u.chrom = chrom;
    u.first = ctx->chromstarts[chrom];
    u.last = ctx->chromstarts[chrom + 1] - 1;
    u.chromstarts_host_upto = ctx->chromstarts[chrom + 1];
    u.chromstarts = ctx->d_chromstarts;
    u.row_of = ctx->d_row_of;
    u.rec_empty = ctx->d_rec_empty;
    u.has_prior = ctx->d_has_prior;
    u.children = ctx->d_children;
    u.descendants = ctx->d_desc;
    u.allele8 = ctx->d_allele8;
    u.sure = ctx->d_sure;
    u.hw = ctx->d_hw;
    u.prior_allele8 = ctx->d_prior_allele8;
    u.prior_sure = ctx->d_prior_sure;
    u.acc_inf = a_inf;
    u.acc_hb = a_hb;
    u.acc_hc = a_hc;
    u.anyinfo = ctx->d_anyinfo;
    u.fw = ctx->d_fw;
    u.ratio = ctx->d_ratio;
    u.relhaplo = 0.5;
    u.scalefactor = scalefactor;
    u.entropyfactor = entropyfactor;
    u.hits = ctx->d_hits;
    if (!(flags & CNF2_UPDATE_PLAIN)) {
        if (!ctx->d_flow_next) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_flow_next, 32 * sizeof(unsigned long long)));
        if ((rc = ensure(ctx, &ctx->d_flow_out, &ctx->flow_out_cap, (RU ? RU : 1) * (size_t)(u.last - u.first + 1) * 4))) return rc;
        // the scouts work through their flows in chunks; a chunk's worth of set-aside entries (24 bytes each), no more
        // than the pass has flows (certainties: 4 per record and marker of the chromosome; weights: 1 per record and marker
        // of the chromosomes so far)
        const size_t chunk = todo_chunk(RU, (size_t)(u.last - u.first + 1), (size_t)u.chromstarts_host_upto);
        if ((rc = ensure(ctx, &ctx->d_todo, &ctx->todo_cap, todo_doubles(chunk)))) return rc;
        u.flow_next = ctx->d_flow_next;
        u.flow_out = ctx->d_flow_out;
        u.stats = getenv("CNF2_UPDATE_STATS") ? ctx->d_flow_next + 2 : nullptr;     // diagnostics only (no effect on results): a few atomics per wavefront
        u.todo = ctx->d_todo;
        u.todo_cap = chunk;
        u.todo2 = ctx->d_todo + chunk * 3;
        u.todo3 = TODO_LISTS > 2 ? ctx->d_todo + chunk * 6 : nullptr;
        u.todo_counts = (unsigned long long*)(ctx->d_todo + chunk * 3 * TODO_LISTS);
        u.scout_passes = (flags & CNF2_UPDATE_ONE_SCOUT) ? 1 : 2;
        u.mirror = (flags & CNF2_UPDATE_BOTH_FLOWS) ? 0 : 1;
        u.literal_finish = (flags & CNF2_UPDATE_LITERAL_FINISH) ? 1 : 0;
    }
    if (u.n_rec > 0) launch_update_pass(u, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    ctx->windows_dirty = true;          // rows changed: the "homozygous everywhere" flags must be derived again
    HIP_TRY(ctx, hipMemcpyAsync(hits_out, ctx->d_hits, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (!acc_dev && given) {
        HIP_TRY(ctx, hipMemcpyAsync(infprobs, a_inf, R * M * 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(haplobase, a_hb, R * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(haplocount, a_hc, R * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_update_pass(cnf2_ctx* ctx, int chrom, const int32_t* children, const int32_t* descendants, double* infprobs,
                     double* haplobase, double* haplocount, double scalefactor, double entropyfactor, int* hits_out,
                     uint32_t flags)
{
    return update_pass_impl(ctx, chrom, nullptr, 0, children, descendants, infprobs, haplobase, haplocount, scalefactor, entropyfactor,
                            hits_out, flags);
}

int cnf2_update_pass_records(cnf2_ctx* ctx, int chrom, const int32_t* recs, int n_recs, const int32_t* children,
                             const int32_t* descendants, double scalefactor, double entropyfactor, int* hits_out, uint32_t flags)
{
    if (!recs && n_recs != 0) return fail(ctx, CNF2_ERR_ARG, "bad record list");
    static const int32_t none = 0;
    return update_pass_impl(ctx, chrom, recs ? recs : &none, n_recs, children, descendants, nullptr, nullptr, nullptr, scalefactor,
                            entropyfactor, hits_out, flags & ~(uint32_t)CNF2_ACC_DEVICE);
}

// ------------------------------------------------------------------------------------------------
// Exchange support of multi-process runs: listed records' accumulators / rows to and from a packed device buffer
// ------------------------------------------------------------------------------------------------
int cnf2_exchange_buffer(cnf2_ctx* ctx, size_t bytes, void** d_buf)
{
    if (!ctx || !d_buf) return fail(ctx, CNF2_ERR_ARG, "bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int rc = ensure(ctx, &ctx->d_xbuf, &ctx->xbuf_cap, bytes ? bytes : 1);
    if (rc) return rc;
    *d_buf = ctx->d_xbuf;
    return CNF2_OK;
}

int cnf2_exchange_read(cnf2_ctx* ctx, size_t offset, void* host_dst, size_t bytes)
{
    if (!ctx || (!host_dst && bytes)) return fail(ctx, CNF2_ERR_ARG, "bad arguments");
    if (offset + bytes > ctx->xbuf_cap || !ctx->d_xbuf) return fail(ctx, CNF2_ERR_ARG, "beyond the exchange buffer");
    if (!bytes) return CNF2_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(host_dst, ctx->d_xbuf + offset, bytes, hipMemcpyDeviceToHost));
    return CNF2_OK;
}

int cnf2_exchange_write(cnf2_ctx* ctx, size_t offset, const void* host_src, size_t bytes)
{
    if (!ctx || (!host_src && bytes)) return fail(ctx, CNF2_ERR_ARG, "bad arguments");
    if (offset + bytes > ctx->xbuf_cap || !ctx->d_xbuf) return fail(ctx, CNF2_ERR_ARG, "beyond the exchange buffer");
    if (!bytes) return CNF2_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->d_xbuf + offset, host_src, bytes, hipMemcpyHostToDevice));
    return CNF2_OK;
}

int cnf2_exchange_download(cnf2_ctx* ctx, void* host_dst, size_t bytes) { return cnf2_exchange_read(ctx, 0, host_dst, bytes); }
int cnf2_exchange_upload(cnf2_ctx* ctx, const void* host_src, size_t bytes) { return cnf2_exchange_write(ctx, 0, host_src, bytes); }

size_t cnf2_packed_accumulator_doubles(const cnf2_ctx* ctx) { return ctx ? (size_t)ctx->n_markers * 6 : 0; }
size_t cnf2_packed_row_bytes(const cnf2_ctx* ctx) { return ctx ? (((size_t)ctx->n_markers * 25 + 7) & ~(size_t)7) : 0; }

// uploads recs (and, with rows, the rows they sit on) as index lists: d_xidx = [recs | rows]
static int exchange_lists(cnf2_ctx* ctx, const int32_t* recs, int n, bool rows)
{
    if (!ctx || (!recs && n > 0) || n < 0) return fail(ctx, CNF2_ERR_ARG, "bad record list");
    if (ctx->ped.n_rec == 0 || !ctx->d_allele8) return fail(ctx, CNF2_ERR_STATE, "rows and pedigree must be uploaded first");
    std::vector<int32_t> idx((size_t)n * 2);
    for (int i = 0; i < n; i++) {
        if (recs[i] < 0 || recs[i] >= ctx->ped.n_rec) return fail(ctx, CNF2_ERR_ARG, "record out of range at %d", i);
        idx[i] = recs[i];
        idx[(size_t)n + i] = ctx->ped.row_of[recs[i]];
        if (rows && idx[(size_t)n + i] == 0) return fail(ctx, CNF2_ERR_ARG, "record %d sits on the shared blank row", recs[i]);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, &ctx->d_xidx, &ctx->xidx_cap, (size_t)(n > 0 ? n : 1) * 2);
    if (rc) return rc;
    if (n > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_xidx, idx.data(), sizeof(int32_t) * idx.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // idx goes out of scope
    }
    return CNF2_OK;
}

static int have_accumulators(cnf2_ctx* ctx)
{
    const size_t R = (size_t)ctx->ped.n_rec, M = (size_t)ctx->n_markers;
    if (!ctx->d_acc_inf || !ctx->d_acc_hb || !ctx->d_acc_hc || ctx->acc_inf_cap < R * M * 4 || ctx->acc_hb_cap < R * M ||
        ctx->acc_hc_cap < R * M)
        return fail(ctx, CNF2_ERR_STATE, "the context holds no accumulators (cnf2_sweep_accumulate with NULL accumulator pointers first)");
    return CNF2_OK;
}

int cnf2_pack_accumulators(cnf2_ctx* ctx, const int32_t* recs, int n, double* d_packed)
{
    int rc = exchange_lists(ctx, recs, n, false);
    if (rc) return rc;
    if (n == 0) return CNF2_OK;
    if (!d_packed) return fail(ctx, CNF2_ERR_ARG, "packed buffer is NULL");
    if ((rc = have_accumulators(ctx))) return rc;
    const size_t M = (size_t)ctx->n_markers, S = M * 6;
    launch_copy_rows_f64(ctx->d_acc_inf, M * 4, ctx->d_xidx, d_packed, S, nullptr, n, M * 4, ctx->stream);
    launch_copy_rows_f64(ctx->d_acc_hb, M, ctx->d_xidx, d_packed + M * 4, S, nullptr, n, M, ctx->stream);
    launch_copy_rows_f64(ctx->d_acc_hc, M, ctx->d_xidx, d_packed + M * 5, S, nullptr, n, M, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_unpack_accumulators(cnf2_ctx* ctx, const int32_t* recs, int n, const double* d_packed)
{
    int rc = exchange_lists(ctx, recs, n, false);
    if (rc) return rc;
    if (n == 0) return CNF2_OK;
    if (!d_packed) return fail(ctx, CNF2_ERR_ARG, "packed buffer is NULL");
    if ((rc = have_accumulators(ctx))) return rc;
    const size_t M = (size_t)ctx->n_markers, S = M * 6;
    launch_copy_rows_f64(d_packed, S, nullptr, ctx->d_acc_inf, M * 4, ctx->d_xidx, n, M * 4, ctx->stream);
    launch_copy_rows_f64(d_packed + M * 4, S, nullptr, ctx->d_acc_hb, M, ctx->d_xidx, n, M, ctx->stream);
    launch_copy_rows_f64(d_packed + M * 5, S, nullptr, ctx->d_acc_hc, M, ctx->d_xidx, n, M, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_pack_rows(cnf2_ctx* ctx, const int32_t* recs, int n, void* d_packed)
{
    int rc = exchange_lists(ctx, recs, n, true);
    if (rc) return rc;
    if (n == 0) return CNF2_OK;
    if (!d_packed) return fail(ctx, CNF2_ERR_ARG, "packed buffer is NULL");
    const size_t   M = (size_t)ctx->n_markers, B = cnf2_packed_row_bytes(ctx);
    const int32_t* rows = ctx->d_xidx + n;
    uint8_t*       q = (uint8_t*)d_packed;
    launch_copy_rows_f64((const double*)ctx->d_sure, M * 2, rows, (double*)q, B / 8, nullptr, n, M * 2, ctx->stream);
    launch_copy_rows_f64(ctx->d_hw, M, rows, (double*)(q + M * 16), B / 8, nullptr, n, M, ctx->stream);
    launch_copy_rows_u8(ctx->d_allele8, M, rows, q + M * 24, B, nullptr, n, M, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_unpack_rows(cnf2_ctx* ctx, const int32_t* recs, int n, const void* d_packed)
{
    int rc = exchange_lists(ctx, recs, n, true);
    if (rc) return rc;
    if (n == 0) return CNF2_OK;
    if (!d_packed) return fail(ctx, CNF2_ERR_ARG, "packed buffer is NULL");
    const size_t   M = (size_t)ctx->n_markers, B = cnf2_packed_row_bytes(ctx);
    const int32_t* rows = ctx->d_xidx + n;
    const uint8_t* q = (const uint8_t*)d_packed;
    launch_copy_rows_f64((const double*)q, B / 8, nullptr, (double*)ctx->d_sure, M * 2, rows, n, M * 2, ctx->stream);
    launch_copy_rows_f64((const double*)(q + M * 16), B / 8, nullptr, ctx->d_hw, M, rows, n, M, ctx->stream);
    launch_copy_rows_u8(q + M * 24, B, nullptr, ctx->d_allele8, M, rows, n, M, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->windows_dirty = true;          // rows changed: the "homozygous everywhere" flags must be derived again
    return CNF2_OK;
}

int cnf2_download_accumulators(cnf2_ctx* ctx, double* infprobs, double* haplobase, double* haplocount)
{
    if (!ctx) return fail(ctx, CNF2_ERR_ARG, "ctx is NULL");
    const size_t R = (size_t)ctx->ped.n_rec, M = (size_t)ctx->n_markers;
    if (!ctx->d_acc_inf || !ctx->d_acc_hb || !ctx->d_acc_hc || ctx->acc_inf_cap < R * M * 4 || ctx->acc_hb_cap < R * M ||
        ctx->acc_hc_cap < R * M)
        return fail(ctx, CNF2_ERR_STATE, "the context holds no accumulators (cnf2_sweep_accumulate with NULL accumulator pointers first)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (infprobs) HIP_TRY(ctx, hipMemcpy(infprobs, ctx->d_acc_inf, R * M * 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (haplobase) HIP_TRY(ctx, hipMemcpy(haplobase, ctx->d_acc_hb, R * M * sizeof(double), hipMemcpyDeviceToHost));
    if (haplocount) HIP_TRY(ctx, hipMemcpy(haplocount, ctx->d_acc_hc, R * M * sizeof(double), hipMemcpyDeviceToHost));
    return CNF2_OK;
}

int cnf2_accumulator_ptrs(cnf2_ctx* ctx, double** infprobs, double** haplobase, double** haplocount)
{
    if (!ctx || !infprobs || !haplobase || !haplocount) return fail(ctx, CNF2_ERR_ARG, "bad arguments");
    const size_t R = (size_t)ctx->ped.n_rec, M = (size_t)ctx->n_markers;
    if (!ctx->d_acc_inf || !ctx->d_acc_hb || !ctx->d_acc_hc || ctx->acc_inf_cap < R * M * 4 || ctx->acc_hb_cap < R * M ||
        ctx->acc_hc_cap < R * M)
        return fail(ctx, CNF2_ERR_STATE, "the context holds no accumulators (cnf2_sweep_accumulate with NULL accumulator pointers first)");
    *infprobs = ctx->d_acc_inf;
    *haplobase = ctx->d_acc_hb;
    *haplocount = ctx->d_acc_hc;
    return CNF2_OK;
}

int cnf2_upload_accumulators(cnf2_ctx* ctx, const double* infprobs, const double* haplobase, const double* haplocount)
{
    if (!ctx) return fail(ctx, CNF2_ERR_ARG, "ctx is NULL");
    const size_t R = (size_t)ctx->ped.n_rec, M = (size_t)ctx->n_markers;
    if (R == 0 || M == 0) return fail(ctx, CNF2_ERR_STATE, "map and pedigree must be uploaded first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, &ctx->d_acc_inf, &ctx->acc_inf_cap, R * M * 4))) return rc;
    if ((rc = ensure(ctx, &ctx->d_acc_hb, &ctx->acc_hb_cap, R * M))) return rc;
    if ((rc = ensure(ctx, &ctx->d_acc_hc, &ctx->acc_hc_cap, R * M))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (infprobs) HIP_TRY(ctx, hipMemcpy(ctx->d_acc_inf, infprobs, R * M * 4 * sizeof(double), hipMemcpyHostToDevice));
    if (haplobase) HIP_TRY(ctx, hipMemcpy(ctx->d_acc_hb, haplobase, R * M * sizeof(double), hipMemcpyHostToDevice));
    if (haplocount) HIP_TRY(ctx, hipMemcpy(ctx->d_acc_hc, haplocount, R * M * sizeof(double), hipMemcpyHostToDevice));
    return CNF2_OK;
}

int cnf2_update_stats(cnf2_ctx* ctx, uint64_t* out16)
{
    if (!ctx || !out16) return fail(ctx, CNF2_ERR_ARG, "bad arguments");
    if (!ctx->d_flow_next) return fail(ctx, CNF2_ERR_STATE, "no update pass has run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out16, ctx->d_flow_next + 2, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return CNF2_OK;
}

int cnf2_update_stats_guided(cnf2_ctx* ctx, uint64_t* out8)
{
    if (!ctx || !out8) return fail(ctx, CNF2_ERR_ARG, "bad arguments");
    if (!ctx->d_flow_next) return fail(ctx, CNF2_ERR_STATE, "no update pass has run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out8, ctx->d_flow_next + 18, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return CNF2_OK;
}

int cnf2_download_rows(cnf2_ctx* ctx, int row0, int n, uint8_t* allele, double* sure, double* hw)
{
    if (!ctx || !allele || !sure || !hw) return fail(ctx, CNF2_ERR_ARG, "bad row arguments");
    if (!ctx->d_allele8) return fail(ctx, CNF2_ERR_STATE, "no rows uploaded");
    if (row0 < 0 || n < 0 || row0 + n > ctx->n_rows) return fail(ctx, CNF2_ERR_ARG, "row range out of bounds");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t M = ctx->n_markers, cnt = (size_t)n * M;
    std::vector<uint8_t> packed(cnt);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(packed.data(), ctx->d_allele8 + (size_t)row0 * M, cnt, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < cnt; i++) {
        allele[i * 2]     = packed[i] & 15;
        allele[i * 2 + 1] = packed[i] >> 4;
    }
    HIP_TRY(ctx, hipMemcpy(sure, ctx->d_sure + (size_t)row0 * M, cnt * sizeof(double2), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(hw, ctx->d_hw + (size_t)row0 * M, cnt * sizeof(double), hipMemcpyDeviceToHost));
    return CNF2_OK;
}

// ------------------------------------------------------------------------------------------------
// Pre-processing users of the emission for ARBITRARY records (postmarkerdata, cnF2freq.cpp:3190-3412)
// ------------------------------------------------------------------------------------------------
// windows of the records recs[0..n): mode 0 = no founder flag anywhere (the state in which main() calls
// postmarkerdata: fixtrees has not run), mode 1 = the flags fixtrees has left when the records are processed in
// ascending order (a member's flag counts if its record index is <= the record's own, cnF2freq.cpp:3373-3389),
// mode 2 = every flag (fixtrees has run on everybody)
static int scan_windows(cnf2_ctx* ctx, const int32_t* recs, int n, int mode)
{
    const HostPedigree& P = ctx->ped;
    std::vector<Window> ws(n);
    for (int i = 0; i < n; i++) {
        if (recs[i] < 0 || recs[i] >= P.n_rec) return fail(ctx, CNF2_ERR_ARG, "record out of range at %d", i);
        int32_t slot_rec[7];
        derive_window(P, recs[i], &ws[i], slot_rec);
        for (int k = 0; k < 7; k++) {
            if (slot_rec[k] < 0) continue;
            if (mode == 0 || (mode == 1 && slot_rec[k] > recs[i])) ws[i].flags[k] &= (uint8_t)~SLOT_FOUNDER;
        }
    }
    int rc = ensure(ctx, &ctx->d_scanwin, &ctx->scanwin_cap, (size_t)n);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(ctx->d_scanwin, ws.data(), sizeof(Window) * n, hipMemcpyHostToDevice));
    return CNF2_OK;
}

int cnf2_fixparents_scan(cnf2_ctx* ctx, const int32_t* recs, int n, uint8_t* ok_out)
{
    if (!ctx || !recs || !ok_out || n < 0) return fail(ctx, CNF2_ERR_ARG, "bad scan arguments");
    if (!ctx->d_rho || !ctx->d_allele8 || ctx->ped.n_rec == 0) return fail(ctx, CNF2_ERR_STATE, "map, rows and pedigree must be uploaded first");
    if (n == 0) return CNF2_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // grid.y holds the record index: slabs of at most 65 535 records (a pedigree of config 4's size has ~300 000)
    const int    slab = 65535;
    const size_t per = (size_t)ctx->n_markers * 2;
    int rc;
    if ((rc = ensure(ctx, &ctx->d_okout, &ctx->okout_cap, (size_t)(n < slab ? n : slab) * per))) return rc;
    for (int i0 = 0; i0 < n; i0 += slab) {
        const int k = n - i0 < slab ? n - i0 : slab;
        if ((rc = scan_windows(ctx, recs + i0, k, 0))) return rc;
        KernelParams p;
        base_params(ctx, &p);
        p.windows = ctx->d_scanwin;
        launch_okvals(p, k, ctx->d_okout, ctx->stream);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(ok_out + (size_t)i0 * per, ctx->d_okout, (size_t)k * per, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CNF2_OK;
}

// ordered: bit 0 = founder flags as fixtrees has assigned them in ascending order; bit 1 = evaluate by brute force
// (the 65 536 emission calls of the reference's loops) instead of the closed form -- cross-check only
int cnf2_variances(cnf2_ctx* ctx, const int32_t* recs, int n, int ordered, double* var_out)
{
    if (!ctx || !recs || !var_out || n < 0) return fail(ctx, CNF2_ERR_ARG, "bad scan arguments");
    if (!ctx->d_rho || !ctx->d_allele8 || ctx->ped.n_rec == 0) return fail(ctx, CNF2_ERR_STATE, "map, rows and pedigree must be uploaded first");
    if (n == 0) return CNF2_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t M = ctx->n_markers;
    const bool brute = (ordered & 2) != 0;
    ordered &= 1;
    const int slab = brute ? 4096 : 65535;       // grid.y
    int rc;
    if ((rc = ensure(ctx, &ctx->d_scratch, &ctx->scratch_cap, (size_t)(n < slab ? n : slab) * M))) return rc;
    for (int i0 = 0; i0 < n; i0 += slab) {
        const int k = n - i0 < slab ? n - i0 : slab;
        if ((rc = scan_windows(ctx, recs + i0, k, ordered ? 1 : 2))) return rc;
        KernelParams p;
        base_params(ctx, &p);
        p.windows = ctx->d_scanwin;
        if (brute) launch_addvariance_batch(p, k, ctx->d_scratch, ctx->stream);
        else launch_variance_closed(p, k, ctx->d_scratch, ctx->stream);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(var_out + (size_t)i0 * M, ctx->d_scratch, (size_t)k * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CNF2_OK;
}

// var_out[q] = addvariance of record recs[q] at marker markers[q] with the reference's own rounding (variance_exact)
int cnf2_variances_exact(cnf2_ctx* ctx, const int32_t* recs, const int32_t* markers, int n, int ordered, double* var_out)
{
    if (!ctx || !recs || !markers || !var_out || n < 0) return fail(ctx, CNF2_ERR_ARG, "bad scan arguments");
    if (!ctx->d_rho || !ctx->d_allele8 || ctx->ped.n_rec == 0) return fail(ctx, CNF2_ERR_STATE, "map, rows and pedigree must be uploaded first");
    if (n == 0) return CNF2_OK;
    for (int q = 0; q < n; q++)
        if (markers[q] < 0 || markers[q] >= ctx->n_markers) return fail(ctx, CNF2_ERR_ARG, "marker out of range at %d", q);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int slab = 1 << 20;
    int       rc;
    // per entry: the result, then (behind all results) its marker
    if ((rc = ensure(ctx, &ctx->d_scratch, &ctx->scratch_cap, (size_t)(n < slab ? n : slab) * 2))) return rc;
    for (int i0 = 0; i0 < n; i0 += slab) {
        const int k = n - i0 < slab ? n - i0 : slab;
        if ((rc = scan_windows(ctx, recs + i0, k, (ordered & 1) ? 1 : 2))) return rc;
        int32_t* d_markers = (int32_t*)(ctx->d_scratch + k);
        HIP_TRY(ctx, hipMemcpyAsync(d_markers, markers + i0, (size_t)k * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        KernelParams p;
        base_params(ctx, &p);
        p.windows = ctx->d_scanwin;
        launch_variance_exact(p, d_markers, k, ctx->d_scratch, ctx->stream);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(var_out + i0, ctx->d_scratch, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CNF2_OK;
}

int cnf2_addvariance(cnf2_ctx* ctx, int ind, int chrom, double* var_out)
{
    int rc = ready(ctx);
    if (rc) return rc;
    if (!var_out || ind < 0 || ind >= (int)ctx->windows.size() || chrom < 0 || chrom >= ctx->n_chrom)
        return fail(ctx, CNF2_ERR_ARG, "bad addvariance arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int first = ctx->chromstarts[chrom], len = ctx->chromstarts[chrom + 1] - first;
    if ((rc = ensure(ctx, &ctx->d_scratch, &ctx->scratch_cap, (size_t)len))) return rc;
    KernelParams p;
    base_params(ctx, &p);
    p.windows = ctx->d_windows + ind;
    launch_addvariance(p, first, len, ctx->d_scratch, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(var_out, ctx->d_scratch, (size_t)len * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_emission(cnf2_ctx* ctx, int ind, int marker, double* e_out)
{
    int rc = ready(ctx);
    if (rc) return rc;
    if (ind < 0 || ind >= (int)ctx->windows.size() || marker < 0 || marker >= ctx->n_markers || !e_out)
        return fail(ctx, CNF2_ERR_ARG, "bad emission arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((rc = ensure(ctx, &ctx->d_scratch, &ctx->scratch_cap, (size_t)512))) return rc;
    KernelParams p;
    base_params(ctx, &p);
    launch_emission(p, ind, marker, ctx->d_scratch, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(e_out, ctx->d_scratch, 512 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_emission_paths(cnf2_ctx* ctx, int ind, int marker, double* e_out)
{
    int rc = ready(ctx);
    if (rc) return rc;
    if (ind < 0 || ind >= (int)ctx->windows.size() || marker < 0 || marker >= ctx->n_markers || !e_out)
        return fail(ctx, CNF2_ERR_ARG, "bad emission arguments");
    const size_t n = (size_t)8 * 64 * 128;
    if ((rc = ensure(ctx, &ctx->d_scratch, &ctx->scratch_cap, n))) return rc;
    KernelParams p;
    base_params(ctx, &p);
    p.windows = ctx->d_windows + ind;
    launch_emission_paths(p, marker, ctx->d_scratch, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(e_out, ctx->d_scratch, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

int cnf2_selftest_lane_xor(cnf2_ctx* ctx, double* out384)
{
    if (!ctx || !out384) return CNF2_ERR_ARG;
    int rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((rc = ensure(ctx, &ctx->d_scratch, &ctx->scratch_cap, (size_t)512))) return rc;
    launch_xor_selftest(ctx->d_scratch, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out384, ctx->d_scratch, 384 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CNF2_OK;
}

} // extern "C"
