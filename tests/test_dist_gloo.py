"""CPU suite: the N > 1 path (block partition of individuals + one gather to rank 0) with
world_size 2 on the gloo backend.  The per-rank sweep is played by the oracle here (the
checker standing in for the GPU, which this container does not have); the code under test is
cnf2freq_amd/dist.py, the same functions bench.py and GPU runs use."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from cnf2freq_amd import dist as cdist
from cnf2freq_amd import synth
from conftest import oracle_ped


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ped = synth.make_f2(7, 20, 1, seed=5, chrom_cm=30.0, missing=0.1)
    o = oracle_ped(ped)

    def sweep(i0, i1):
        r = o.sweep_batch(ped.dous[i0:i1], ped.gen[ped.dous[i0:i1]], mode=2, n_threads=1)
        return dict(factors=r["factors"], loglik=r["factor"], dosage=r["dosage"])

    res = cdist.run_sharded(len(ped.dous), sweep)
    if rank == 0:
        full = o.sweep_batch(ped.dous, ped.gen[ped.dous], mode=2, n_threads=1)
        ok = (np.array_equal(res["factors"], full["factors"]) and np.array_equal(res["loglik"], full["factor"])
              and np.array_equal(res["dosage"], full["dosage"]))
        q.put(bool(ok))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 10, 10000):
        for world in (1, 2, 3, 8):
            ranges = [cdist.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_sweep_and_gather_equals_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _tile_worker(rank, world, port, q):
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, M = 5, 23
    src = torch.arange(n * M * 3, dtype=torch.float64).reshape(n, M, 3) + 1000.0 * rank
    tg = cdist.TiledGather(n, M, 3, 6, torch.float64, torch.device("cpu"), dst=0, depth=2)
    got = np.zeros((world, n, M, 3))
    calls = []

    def consume(m0, m1, parts):
        calls.append((m0, m1))
        for r, t in enumerate(parts):
            got[r, :, m0:m1] = t.numpy()

    nt = tg.run(src, consume)
    if rank == 0:
        want = np.stack([(np.arange(n * M * 3, dtype=np.float64).reshape(n, M, 3) + 1000.0 * r) for r in range(world)])
        ok = (np.array_equal(got, want) and calls == [(0, 6), (6, 12), (12, 18), (18, 23)] and nt == 4
              and tg.root_bytes() == 2 * world * n * 6 * 3 * 8)
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_marker_tiled_gather_two_ranks():
    """The root holds depth x world tiles, never the whole posterior (BASELINE config 4 would be 480 GB)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_marker_tiled_gather_single_process():
    import torch
    n, M = 3, 10
    src = torch.rand(n, M, 3, dtype=torch.float64)
    tg = cdist.TiledGather(n, M, 3, 4, torch.float64, torch.device("cpu"))
    out = torch.zeros_like(src)
    tg.run(src, lambda m0, m1, parts: out[:, m0:m1].copy_(parts[0]))
    assert torch.equal(out, src) and tg.root_bytes() == 0


def _acc_worker(rank, world, port, q):
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R, M = 4, 5
    g = torch.Generator().manual_seed(100 + rank)
    parts = [torch.rand((R, M, 2, 2), generator=g, dtype=torch.float64), torch.rand((R, M), generator=g, dtype=torch.float64),
             torch.rand((R, M), generator=g, dtype=torch.float64)]
    mine = [p.clone() for p in parts]
    cdist.allreduce_accumulators(*parts)
    # every rank must hold the sum of all ranks' partial accumulators
    want = [torch.zeros_like(p) for p in parts]
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        for w, shape in zip(want, [(R, M, 2, 2), (R, M), (R, M)]):
            w += torch.rand(shape, generator=gr, dtype=torch.float64)
    ok = all(torch.allclose(a, b, rtol=1e-14, atol=0) for a, b in zip(parts, want)) and not torch.equal(parts[0], mine[0])
    q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_accumulator_allreduce_two_ranks():
    """The collective of a haplotyping iteration (shared-ancestor accumulator slabs, cnF2freq.cpp:6245-6254)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_acc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True and q.get(timeout=5) is True


def test_balanced_blocks_cover_and_balance():
    """Contiguous blocks balanced by cost: windows with tie groups cost up to 17 passes where an
    untied one costs 2."""
    from cnf2freq_amd import dist as cdist
    rs = np.random.RandomState(3)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 40, 1000):
            costs = np.where(rs.rand(n) < 0.05, 17.0, 2.0) * 1000
            b = cdist.balanced_blocks(costs, world)
            assert b[0] == 0 and b[-1] == n and len(b) == world + 1 and all(x <= y for x, y in zip(b, b[1:]))
            if n >= 100 * world:
                share = [costs[x:y].sum() for x, y in zip(b, b[1:])]
                assert max(share) - min(share) <= 2 * costs.max()
    # head-count would give [0, 4, 8]; the tied window at position 3 moves the boundary
    assert cdist.balanced_blocks([2, 2, 2, 17, 2, 2, 2, 2], 2) == [0, 4, 8]
    assert cdist.balanced_blocks([17, 2, 2, 2, 2, 2, 2, 2], 2) == [0, 1, 8]
