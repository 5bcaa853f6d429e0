"""Multi-GPU plumbing: individuals are independent units of the sweep (the reference's own
dead MPI code partitions `dous` the same way, cnF2freq.cpp:5297-5299), so the path shards with
no data-path collective; the only exchange is one gather of the per-individual results to
rank 0 (RCCL over xGMI when the backend is nccl).  One process per GPU, torch.distributed."""
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


def allreduce_accumulators(infprobs, haplobase, haplocount):
    """The one collective of a haplotyping iteration: ranks own disjoint blocks of analysed individuals but share
    ancestors, whose per-record accumulators (infprobs [R, M, 2, 2], haplobase / haplocount [R, M]; what
    moveinfprobs / movehaplos add up, cnF2freq.cpp:3577-3616) every rank holds a partial sum of.  One
    all-reduce(sum) of the three slabs (the reference's dead MPI code reduces them per individual,
    cnF2freq.cpp:6245-6254); every rank then runs the same update pass on the same numbers.  In place; tensors may
    live on the GPU (nccl = RCCL) or on the host (gloo)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True) for t in (infprobs, haplobase, haplocount)]
    for w in works:
        w.wait()


def balanced_blocks(costs, world):
    """Boundaries [b_0 = 0, ..., b_world = n] of `world` contiguous blocks with near-equal cost (the rule of
    Engine::balanced_block in csrc/host/cnf2_engine.cpp, for callers that hold the costs themselves)."""
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


class _DeviceSlab:
    """A device address as an object torch.as_tensor() accepts (CUDA array interface; zero copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr="<f8", data=(int(ptr), False), version=2)


def device_slabs(d_inf, d_hb, d_hc, n_rec, n_markers, device=None):
    """The three accumulator slabs at the device addresses the engine hands to its exchange callback, as flat float64
    torch tensors that alias them."""
    from . import capi
    capi.require_single_hip_runtime()          # the addresses belong to the runtime libcnf2hip.so uses
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    return [torch.as_tensor(_DeviceSlab(p, n), device=dev)
            for p, n in ((d_inf, n_rec * n_markers * 4), (d_hb, n_rec * n_markers), (d_hc, n_rec * n_markers))]


def make_exchange(run):
    """The exchange callback of a multi-process haplotyping run (cnf2host.h: cnf2h_set_exchange): one all-reduce(sum) of the
    three accumulator slabs per iteration.  nccl (= RCCL): in place on the device addresses the engine hands over;
    gloo (a transport that moves host memory, e.g. two test ranks sharing one GPU): through the host with
    cnf2_download_accumulators / cnf2_upload_accumulators."""
    from . import capi
    backend = dist.get_backend() if dist.is_initialized() else None
    L = capi.load()
    ctx = run.context()
    bufs = {}

    def exchange(d_inf, d_hb, d_hc, n_rec, n_markers):
        if backend is None or dist.get_world_size() == 1:
            return 0
        if backend == "nccl":
            allreduce_accumulators(*device_slabs(d_inf, d_hb, d_hc, n_rec, n_markers))
            torch.cuda.synchronize()
            return 0
        if "inf" not in bufs:
            bufs["inf"] = torch.zeros(n_rec * n_markers * 4, dtype=torch.float64)
            bufs["hb"] = torch.zeros(n_rec * n_markers, dtype=torch.float64)
            bufs["hc"] = torch.zeros(n_rec * n_markers, dtype=torch.float64)
        import ctypes as C
        ptr = lambda t: C.c_void_p(t.data_ptr())
        if L.cnf2_download_accumulators(ctx, ptr(bufs["inf"]), ptr(bufs["hb"]), ptr(bufs["hc"])) != 0:
            return -1
        allreduce_accumulators(bufs["inf"], bufs["hb"], bufs["hc"])
        if L.cnf2_upload_accumulators(ctx, ptr(bufs["inf"]), ptr(bufs["hb"]), ptr(bufs["hc"])) != 0:
            return -1
        return 0

    return exchange


def start_iterations(ped, device=0, has_prior=None, postmarkerdata=True, deterministic=False, quiet=True):
    """One rank's run of a multi-process haplotyping job (BASELINE config 5; the reference's dead MPI code: partition
    cnF2freq.cpp:5297-5299, reduce 6245-6254, updates on the reduced slabs 6344-6392): the whole pedigree on every rank,
    the rank's work-balanced block of analysed individuals, the all-reduce of the accumulators as the one exchange of
    an iteration.  Returns the cnf2freq_amd.host.Run; call run.iteration() in step on every rank."""
    from . import host
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    run = host.Run(ped, has_prior=has_prior, quiet=quiet, device=device)
    if postmarkerdata:
        run.postmarkerdata()            # replicated: every rank infers the same genotypes from the same rows
    b, e = run.balanced_block(rank, world)
    run.set_block(b, e)
    run.block = (b, e)
    run.set_exchange(make_exchange(run))
    if deterministic:
        run.set_deterministic(True)
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
