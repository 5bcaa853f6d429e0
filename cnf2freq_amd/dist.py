"""Multi-GPU plumbing: individuals are independent units of the sweep (the reference's own
dead MPI code partitions `dous` the same way, cnF2freq.cpp:5297-5299), so the path shards with
no data-path collective; the only exchange of a sweep is one gather of the per-individual results to
rank 0 (RCCL over xGMI when the backend is nccl).  Haplotyping iterations add the exchange of what the ranks' windows
share (Transport).  One process per GPU, torch.distributed."""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block of analysed individuals owned by `rank` (balanced to within one)."""
    base, extra = divmod(n, world)
    i0 = rank * base + min(rank, extra)
    return i0, i0 + base + (1 if rank < extra else 0)


def gather_to_root(t, dst=0, out=None):
    """Gather equally-shaped tensors to rank `dst`; returns the list there, None elsewhere.
    Works for nccl (device tensors) and gloo (CPU tensors).  `out` may hold preallocated
    receive buffers on `dst` (a list of world_size tensors)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [t]
    if dist.get_rank() == dst:
        if out is None:
            out = [torch.empty_like(t) for _ in range(world)]
    else:
        out = None
    dist.gather(t, out, dst=dst)
    return out


class TiledGather:
    """Streaming gather of a [n_ind, n_markers, k] device (or CPU) tensor to rank `dst` in marker tiles.

    Rank `dst` never holds more than `depth` tiles x world_size (at BASELINE config 4 the
    posteriors are 60 GB per GPU, 480 GB in all -- they do not fit one GPU, so the root consumes them tile by
    tile: `consume(m0, m1, parts)` is called on `dst` with the list of world_size tensors [n_ind, m1-m0, k]
    of ranks 0..world-1, valid until the next call that reuses the buffer).  The marker slice is strided in
    the source, so every rank packs it into a contiguous staging buffer first; with `depth` = 2 the gather of
    tile t runs (async_op) while tile t+1 is packed.  One collective (gather) per tile, nothing else."""

    def __init__(self, n_ind, n_markers, k, tile_markers, dtype, device, dst=0, depth=2):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.dst, self.depth = dst, max(1, depth)
        self.n_markers = n_markers
        self.tile = max(1, min(int(tile_markers), n_markers))
        shape = (n_ind, self.tile, k)
        self.stage = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.depth)]
        self.recv = None
        if self.rank == dst and self.world > 1:
            self.recv = [[torch.empty(shape, dtype=dtype, device=device) for _ in range(self.world)]
                         for _ in range(self.depth)]

    def root_bytes(self):
        """Bytes this object holds on the root for receiving (the bound the design promises)."""
        if self.recv is None:
            return 0
        return sum(t.numel() * t.element_size() for bufs in self.recv for t in bufs)

    def n_tiles(self):
        return (self.n_markers + self.tile - 1) // self.tile

    def run(self, src, consume=None):
        """Gather `src` [n_ind, n_markers, k]; returns the number of tiles sent."""
        pending = [None] * self.depth    # (work, m0, m1) per buffer
        def finish(i):
            if pending[i] is None:
                return
            work, m0, m1 = pending[i]
            if work is not None:
                work.wait()
            if self.rank == self.dst and consume is not None:
                w = m1 - m0
                parts = ([self.stage[i][:, :w]] if self.world == 1 else [t[:, :w] for t in self.recv[i]])
                consume(m0, m1, parts)
            pending[i] = None
        t = 0
        for m0 in range(0, self.n_markers, self.tile):
            m1 = min(m0 + self.tile, self.n_markers)
            i = t % self.depth
            finish(i)
            self.stage[i][:, :m1 - m0].copy_(src[:, m0:m1])
            work = None
            if self.world > 1:
                work = dist.gather(self.stage[i], self.recv[i] if self.rank == self.dst else None, dst=self.dst,
                                   async_op=True)
            pending[i] = (work, m0, m1)
            t += 1
        for j in range(self.depth):
            finish((t + j) % self.depth)
        return t


def balanced_blocks(costs, world):
    """Boundaries [b_0 = 0, ..., b_world = n] of `world` contiguous blocks with near-equal cost (the cost-balanced starting
    point of Engine::plan in csrc/host/cnf2_engine.cpp, for callers that hold the costs themselves)."""
    costs = np.asarray(costs, np.float64)
    n = len(costs)
    total = float(costs.sum())
    prefix = np.concatenate([[0.0], np.cumsum(costs)])[:-1]
    bounds = [0]
    for k in range(1, world):
        hit = np.flatnonzero(prefix + 0.5 * costs >= total * k / world)
        bounds.append(int(hit[0]) if len(hit) else n)
    bounds.append(n)
    return bounds


class _DeviceBuffer:
    """A device address as an object torch.as_tensor() accepts (CUDA array interface; zero copy)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr=typestr, data=(int(ptr), False), version=2)


def device_view(ptr, count, dtype, device):
    """`count` elements of `dtype` (torch.float64 or torch.uint8) at device address `ptr` as a flat tensor that aliases them."""
    from . import capi
    capi.require_single_hip_runtime()          # the address belongs to the runtime libcnf2hip.so uses
    typestr = {torch.float64: "<f8", torch.uint8: "|u1"}[dtype]
    return torch.as_tensor(_DeviceBuffer(ptr, count, typestr), device=device)


class Transport:
    """The collectives of a multi-process haplotyping run (cnf2host.h: cnf2h_exchange_fn) over torch.distributed.  The engine
    has packed what the ranks' windows share into one buffer of `world` equal segments; this object only moves it:
      X_SUM_SEGMENTS     reduce-scatter(sum): segment `rank` of the buffer receives the sum over ranks of that segment
      X_SUM_HITS         all-reduce(sum) of a few host integers (the hit counters of an update pass)
      X_GATHER_SEGMENTS  all-gather: every rank's own segment to all
    nccl (= RCCL over xGMI): in place on the engine's device buffer, wrapped zero-copy.  gloo (a transport for host
    memory -- e.g. test ranks sharing one GPU): staged through pinned-size host tensors; gloo has no reduce-scatter, so the
    sum is an all-reduce of which the rank keeps its segment.  `bytes_moved` counts what this rank handed to collectives."""

    def __init__(self, device, ctx=None, skip_single=True):
        from . import host
        self.host = host
        self.skip_single = skip_single      # a world of one has nothing to exchange (False: run the collectives all the same)
        self.ctx = ctx              # cnf2_ctx handle of the run (the gloo path stages the context's exchange buffer)
        self.device = device if isinstance(device, torch.device) else (None if device is None else torch.device("cuda", int(device)))
        self.backend = dist.get_backend() if dist.is_initialized() else None
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.bytes_moved = 0
        self.calls = {0: 0, 1: 0, 2: 0, 3: 0, 4: 0}

    def __call__(self, op, buf, count, seg):
        import ctypes as C
        H = self.host
        self.calls[op] += 1
        if self.world == 1 and (self.skip_single or self.backend is None):
            return 0
        if op == H.X_BARRIER:
            dist.barrier()
            return 0
        if op == H.X_BCAST_HOST:
            # host bytes of rank 0 to every rank, in place (through the GPU for nccl, which moves device memory only)
            t = torch.from_numpy(np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(count,)))
            if self.backend == "nccl":
                d = t.to(self.device)
                dist.broadcast(d, src=0)
                if self.rank != 0:
                    t.copy_(d.cpu())
            else:
                dist.broadcast(t, src=0)
            self.bytes_moved += count
            return 0
        if op == H.X_SUM_HITS:
            a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_int32)), shape=(count,))
            t = torch.from_numpy(a.astype(np.int64))
            if self.backend == "nccl":
                t = t.to(self.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            a[:] = t.cpu().numpy().astype(np.int32)
            self.bytes_moved += 8 * count
            return 0
        dtype = torch.float64 if op == H.X_SUM_SEGMENTS else torch.uint8
        if count != seg * self.world:
            raise ValueError("the buffer must hold world x seg elements")
        if self.backend == "nccl":
            if torch.cuda.current_device() != self.device.index:
                raise RuntimeError("the transport's device %s is not torch's current device %d" % (self.device, torch.cuda.current_device()))
            t = device_view(buf, count, dtype, self.device)
            mine = t[self.rank * seg:(self.rank + 1) * seg]
            if op == H.X_SUM_SEGMENTS:
                dist.reduce_scatter_tensor(mine, t, op=dist.ReduceOp.SUM)
            else:
                dist.all_gather_into_tensor(t, mine)
            torch.cuda.synchronize(self.device)
        else:
            itemsize = 8 if op == H.X_SUM_SEGMENTS else 1
            L = None
            if self.ctx is None:
                # no context: `buf` is HOST memory (callers that stage the buffer themselves; the CPU tests)
                ctype = C.c_double if op == H.X_SUM_SEGMENTS else C.c_uint8
                h = torch.from_numpy(np.ctypeslib.as_array(C.cast(buf, C.POINTER(ctype)), shape=(count,)))
            else:
                # staged through the host: the engine's buffer is the context's exchange buffer (cnf2_exchange_buffer)
                from . import capi
                L = capi.load()
                h = torch.empty(count, dtype=dtype)
                if L.cnf2_exchange_download(self.ctx, C.c_void_p(h.data_ptr()), count * itemsize) != 0:
                    raise RuntimeError("cnf2_exchange_download: " + L.cnf2_last_error(self.ctx).decode())
            if op == H.X_SUM_SEGMENTS:
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
            else:
                parts = [torch.empty(seg, dtype=dtype) for _ in range(self.world)]
                dist.all_gather(parts, h[self.rank * seg:(self.rank + 1) * seg].clone())
                h.copy_(torch.cat(parts))
            if L is not None and L.cnf2_exchange_upload(self.ctx, C.c_void_p(h.data_ptr()), count * itemsize) != 0:
                raise RuntimeError("cnf2_exchange_upload: " + L.cnf2_last_error(self.ctx).decode())
        self.bytes_moved += count * (8 if op == H.X_SUM_SEGMENTS else 1)
        return 0


def start_iterations(ped, device=0, has_prior=None, postmarkerdata=True, deterministic=False, quiet=True):
    """One rank's run of a multi-process haplotyping job (BASELINE config 5; the reference's dead MPI code: partition
    cnF2freq.cpp:5297-5299, reduce 6245-6254, updates on the reduced values 6344-6392): the whole pedigree on every rank, the
    rank's block of analysed individuals cut where the fewest records straddle, the accumulators of the records ranks share
    summed by one reduce-scatter per iteration, every rank updating the records it owns, the shared records' new rows
    all-gathered.  Returns the cnf2freq_amd.host.Run (run.block, run.plan, run.transport); call run.iteration() -- and
    run.state() / run.dump(), which gather the whole state -- in step on every rank."""
    from . import host
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    run = host.Run(ped, has_prior=has_prior, quiet=quiet, device=device)
    run.transport = Transport(device, run.context())
    run.plan = run.set_partition(rank, world, run.transport)
    run.block = run.plan["block"]
    if postmarkerdata:
        run.postmarkerdata()            # rank 0 infers the genotypes for all and broadcasts the rows it leaves
    if deterministic:
        run.set_deterministic(True)
    run.reserve()                       # the rank's batch buffers now, not inside its first iteration
    return run


def gather_ragged_to_root(a, dst=0):
    """Gather numpy arrays whose first dimension differs per rank (block partition);
    returns the concatenation on `dst`, None elsewhere."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return a
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    n = torch.tensor([a.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = np.zeros((mx,) + a.shape[1:], dtype=a.dtype)
    pad[:a.shape[0]] = a
    parts = gather_to_root(torch.from_numpy(pad).to(dev), dst)
    if parts is None:
        return None
    return np.concatenate([p.cpu().numpy()[:k] for p, k in zip(parts, sizes)], axis=0)


def run_sharded(n_ind, sweep_fn, dst=0):
    """Every rank sweeps its block [i0, i1) with sweep_fn(i0, i1) -> dict of numpy arrays indexed
    by individual on axis 0; rank `dst` receives the dict for all individuals."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    i0, i1 = shard_range(n_ind, rank, world)
    local = sweep_fn(i0, i1)
    out = {}
    for k in sorted(local):
        out[k] = gather_ragged_to_root(np.ascontiguousarray(local[k]), dst)
    return out if rank == dst else None
