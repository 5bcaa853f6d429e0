// cnf2_shm_transport.h -- the transport of `cnF2freq --gpus N` (ExchangeFn of cnf2_engine.h): N ranks forked from one
// process, one GPU each, exchanging through a shared memory region on the host.
//
// What ranks exchange in a haplotyping run is small by construction (cnf2_partition.h: the records their windows SHARE --
// nothing at all when the families are disjoint -- and a hit counter per chromosome pass), so the collectives are staged
// through the host: every rank copies its part of the engine's device buffer into its slot of the region, a
// process-shared barrier, every rank reads what it needs of the others' slots.  The whole state is gathered the same way
// once, at the end of a run.  (A pedigree whose ranks share thousands of records would want RCCL on the device buffer in
// place: that is cnf2freq_amd/dist.py's transport; the engine does not know the difference.)
//
// The region is created BEFORE the ranks are forked and before anything has touched the GPU (a process that has
// initialised HIP must not fork workers that use it): the parent only reads the input files, forks, and waits.
#ifndef CNF2_SHM_TRANSPORT_H
#define CNF2_SHM_TRANSPORT_H

#include <pthread.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>

#include <algorithm>
#include <vector>

#include "cnf2_engine.h"
#include "cnf2hip.h"

namespace cnf2host {

struct ShmRegion {
    pthread_barrier_t barrier;
    int               world;
    size_t            slot_bytes;          // staging bytes per rank
    int32_t           hits[64 * 16];       // [world][16] counters of X_SUM_HITS
    // followed by world slots of slot_bytes
    unsigned char*    slot(int r) { return (unsigned char*)(this + 1) + (size_t)r * slot_bytes; }
};

// maps the region for `world` ranks (anonymous, shared with the children a later fork() creates); nullptr on failure
inline ShmRegion* shm_region_create(int world, size_t slot_bytes)
{
    if (world < 1 || world > 64) return nullptr;
    const size_t total = sizeof(ShmRegion) + (size_t)world * slot_bytes;
    void*        p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    ShmRegion* R = (ShmRegion*)p;
    memset(R, 0, sizeof(ShmRegion));
    R->world = world;
    R->slot_bytes = slot_bytes;
    pthread_barrierattr_t a;
    pthread_barrierattr_init(&a);
    pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
    const int rc = pthread_barrier_init(&R->barrier, &a, (unsigned)world);
    pthread_barrierattr_destroy(&a);
    return rc == 0 ? R : nullptr;
}

struct ShmTransport {
    ShmRegion* R = nullptr;
    cnf2_ctx*  ctx = nullptr;
    int        rank = 0;
    size_t     bytes_moved = 0;
    // where the engine's buffer is: the context's exchange buffer on the device, or (host_buf: the CPU self-test of
    // tests/shim) plain host memory
    unsigned char* host_buf = nullptr;

    void wait() { pthread_barrier_wait(&R->barrier); }
    int  xread(size_t offset, void* dst, size_t bytes)
    {
        if (host_buf) {
            memcpy(dst, host_buf + offset, bytes);
            return CNF2_OK;
        }
        return cnf2_exchange_read(ctx, offset, dst, bytes);
    }
    int xwrite(size_t offset, const void* src, size_t bytes)
    {
        if (host_buf) {
            memcpy(host_buf + offset, src, bytes);
            return CNF2_OK;
        }
        return cnf2_exchange_write(ctx, offset, src, bytes);
    }

    // every collective works on the engine's exchange buffer (cnf2_exchange_buffer): `buf` is its device address, which
    // cnf2_exchange_read / _write address by offset
    int sum_segments(size_t count, size_t seg)
    {
        const int    W = R->world;
        const size_t chunk = R->slot_bytes / sizeof(double);
        std::vector<double> acc;
        // the part [off, off + n) of THIS rank's segment: every rank publishes its partial sums of it, this rank adds them up
        for (size_t off = 0; off < seg; off += chunk) {
            const size_t n = std::min(chunk, seg - off);
            for (int q = 0; q < W; q++) {
                // round q: all ranks publish their values of rank q's segment part; rank q sums
                if (xread(((size_t)q * seg + off) * sizeof(double), R->slot(rank), n * sizeof(double)) != CNF2_OK) return -1;
                wait();
                if (q == rank) {
                    acc.assign(n, 0.0);
                    for (int r = 0; r < W; r++) {
                        const double* v = (const double*)R->slot(r);
                        for (size_t i = 0; i < n; i++) acc[i] += v[i];
                    }
                    if (xwrite(((size_t)q * seg + off) * sizeof(double), acc.data(), n * sizeof(double)) != CNF2_OK) return -1;
                }
                wait();
                bytes_moved += n * sizeof(double);
            }
        }
        (void)count;
        return 0;
    }
    int gather_segments(size_t count, size_t seg)
    {
        const int W = R->world;
        for (size_t off = 0; off < seg; off += R->slot_bytes) {
            const size_t n = std::min(R->slot_bytes, seg - off);
            if (xread((size_t)rank * seg + off, R->slot(rank), n) != CNF2_OK) return -1;
            wait();
            for (int r = 0; r < W; r++)
                if (r != rank && xwrite((size_t)r * seg + off, R->slot(r), n) != CNF2_OK) return -1;
            wait();
            bytes_moved += n;
        }
        (void)count;
        return 0;
    }
    int sum_hits(int32_t* h, size_t count)
    {
        if (count > 16) return -1;
        for (size_t i = 0; i < count; i++) R->hits[rank * 16 + i] = h[i];
        wait();
        for (size_t i = 0; i < count; i++) {
            int32_t s = 0;
            for (int r = 0; r < R->world; r++) s += R->hits[r * 16 + i];
            h[i] = s;
        }
        wait();
        return 0;
    }

    int bcast_host(unsigned char* h, size_t count)
    {
        for (size_t off = 0; off < count; off += R->slot_bytes) {
            const size_t n = std::min(R->slot_bytes, count - off);
            if (rank == 0) memcpy(R->slot(0), h + off, n);
            wait();
            if (rank != 0) memcpy(h + off, R->slot(0), n);
            wait();
        }
        return 0;
    }

    static int call(void* user, int op, void* buf, size_t count, size_t seg)
    {
        ShmTransport* T = (ShmTransport*)user;
        switch (op) {
        case X_BCAST_HOST: return T->bcast_host((unsigned char*)buf, count);
        case X_SUM_SEGMENTS: return T->sum_segments(count, seg);
        case X_SUM_HITS: return T->sum_hits((int32_t*)buf, count);
        case X_GATHER_SEGMENTS: return T->gather_segments(count, seg);
        case X_BARRIER: T->wait(); return 0;
        }
        return -1;
    }
};

}  // namespace cnf2host
#endif
