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
