// cnf2_rccl_transport.h -- the transport of `cnF2freq --gpus N` when its ranks sit on distinct GPUs: the collectives of a
// haplotyping run (ExchangeFn of cnf2_engine.h) as RCCL calls IN PLACE on the context's exchange buffer, on the context's
// own stream -- what the ranks' windows share goes GPU to GPU over xGMI and never through the host.
//   X_SUM_SEGMENTS     ncclReduceScatter  (the reference's reduce per individual, cnF2freq.cpp:6245-6254: here only the records
//                                          several ranks' windows touch, packed by owner; the owner receives the total)
//   X_GATHER_SEGMENTS  ncclAllGather      (the owners' new rows of those records / the state's gather, in slices)
//   X_SUM_HITS, X_BARRIER, X_BCAST_HOST   host values (a counter per chromosome pass; rank 0's postmarkerdata result, host
//                                          memory): through the shared region of cnf2_shm_transport.h, which also carries
//                                          the communicator's unique id from rank 0 to the others
// One communicator per rank, made after the fork and after the rank has taken its GPU (a forked process never inherits a HIP
// or RCCL state here: the parent has touched neither).  `--single-device` (every rank on GPU 0: the rehearsal on a one-GPU
// box) keeps the shared-memory transport: RCCL refuses two ranks on one device.
#ifndef CNF2_RCCL_TRANSPORT_H
#define CNF2_RCCL_TRANSPORT_H

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "cnf2_shm_transport.h"

namespace cnf2host {

struct RcclTransport {
    ShmTransport shm;                 // host-side operations and the rendezvous
    ncclComm_t   comm = nullptr;
    hipStream_t  stream = nullptr;
    int          rank = 0, world = 1;
    size_t       bytes_moved = 0;
    std::string  error;

    // every rank calls this (a collective): rank 0's unique id through the shared region, then ncclCommInitRank
    int init(ShmRegion* R, cnf2_ctx* ctx, int rank_, int world_)
    {
        shm.R = R;
        shm.ctx = ctx;
        shm.rank = rank_;
        rank = rank_;
        world = world_;
        stream = (hipStream_t)cnf2_stream(ctx);
        if (!R || R->slot_bytes < sizeof(ncclUniqueId)) return fail("the shared region is too small for the communicator's id");
        ncclUniqueId id;
        if (rank == 0) {
            if (ncclGetUniqueId(&id) != ncclSuccess) return fail("ncclGetUniqueId");
            memcpy(R->slot(0), &id, sizeof(id));
        }
        shm.wait();
        memcpy(&id, R->slot(0), sizeof(id));
        shm.wait();
        const ncclResult_t rc = ncclCommInitRank(&comm, world, id, rank);
        if (rc != ncclSuccess) return fail(std::string("ncclCommInitRank: ") + ncclGetErrorString(rc));
        return 0;
    }
    void finish()
    {
        if (comm) ncclCommDestroy(comm);
        comm = nullptr;
    }
    int fail(const std::string& what)
    {
        error = what;
        fprintf(stderr, "RCCL transport (rank %d of %d): %s\n", rank, world, what.c_str());
        return -1;
    }
    int done(ncclResult_t rc, const char* what)
    {
        if (rc != ncclSuccess) return fail(std::string(what) + ": " + ncclGetErrorString(rc));
        // the engine unpacks on the same stream; the wait keeps the transport's contract (the buffer is ready on return)
        if (hipStreamSynchronize(stream) != hipSuccess) return fail(std::string(what) + ": the stream failed");
        return 0;
    }
    int sum_segments(void* buf, size_t seg)                     // `world` segments of `seg` doubles; mine gets the total
    {
        bytes_moved += seg * sizeof(double) * (size_t)(world - 1);
        return done(ncclReduceScatter(buf, (double*)buf + (size_t)rank * seg, seg, ncclDouble, ncclSum, comm, stream), "ncclReduceScatter");
    }
    int gather_segments(void* buf, size_t seg)                  // `world` segments of `seg` bytes; everybody gets all
    {
        bytes_moved += seg * (size_t)(world - 1);
        return done(ncclAllGather((const char*)buf + (size_t)rank * seg, buf, seg, ncclChar, comm, stream), "ncclAllGather");
    }
    static int call(void* user, int op, void* buf, size_t count, size_t seg)
    {
        RcclTransport* T = (RcclTransport*)user;
        switch (op) {
        case X_SUM_SEGMENTS: return T->sum_segments(buf, seg);
        case X_GATHER_SEGMENTS: return T->gather_segments(buf, seg);
        default: return ShmTransport::call(&T->shm, op, buf, count, seg);      // host values
        }
    }
};

}  // namespace cnf2host
#endif
