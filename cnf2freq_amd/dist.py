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
