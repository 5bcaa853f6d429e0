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
from conftest import ROOT, oracle_ped


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


def _transport_worker(rank, world, port, q):
    import ctypes as C
    import torch
    from cnf2freq_amd import host
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = cdist.Transport(None, ctx=None)          # no context: the buffers are host memory
    seg = 7
    ok = True
    # X_SUM_SEGMENTS: segment `rank` must hold the sum over ranks of that segment
    mine = np.random.RandomState(100 + rank).rand(world * seg)
    buf = mine.copy()
    ok &= T(host.X_SUM_SEGMENTS, buf.ctypes.data, world * seg, seg) == 0
    want = sum(np.random.RandomState(100 + r).rand(world * seg) for r in range(world))
    ok &= bool(np.allclose(buf[rank * seg:(rank + 1) * seg], want[rank * seg:(rank + 1) * seg], rtol=1e-14, atol=0))
    # X_GATHER_SEGMENTS: every rank's own segment to all
    rows = np.zeros(world * seg, np.uint8)
    rows[rank * seg:(rank + 1) * seg] = 10 * (rank + 1) + np.arange(seg)
    ok &= T(host.X_GATHER_SEGMENTS, rows.ctypes.data, world * seg, seg) == 0
    ok &= bool(np.array_equal(rows, np.concatenate([10 * (r + 1) + np.arange(seg) for r in range(world)]).astype(np.uint8)))
    # X_SUM_HITS
    hits = np.array([rank + 1, 5], np.int32)
    ok &= T(host.X_SUM_HITS, hits.ctypes.data, 2, 2) == 0
    ok &= hits.tolist() == [world * (world + 1) // 2, 5 * world]
    # X_BCAST_HOST: rank 0's host bytes to every rank
    state = (np.arange(1000) * (7 if rank == 0 else 0) % 251).astype(np.uint8)
    ok &= T(host.X_BCAST_HOST, state.ctypes.data, 1000, 0) == 0
    ok &= bool(np.array_equal(state, (np.arange(1000) * 7 % 251).astype(np.uint8)))
    ok &= T.calls == {0: 1, 1: 1, 2: 1, 3: 0, 4: 1} and T.bytes_moved == world * seg * 9 + 16 + 1000
    q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_transport_collectives_two_ranks():
    """The three collectives of a haplotyping iteration's exchange (cnf2host.h: the reduce-scatter of the shared records'
    accumulators -- the reference reduces per individual, cnF2freq.cpp:6245-6254 --, the sum of the hit counters, the
    all-gather of the shared records' rows) over gloo on host buffers."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_transport_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True and q.get(timeout=5) is True


def _shim():
    from conftest import build_host_shim
    return build_host_shim()


def test_shared_memory_transport_of_the_executable():
    """The transport of `cnF2freq --gpus N` (csrc/host/cnf2_shm_transport.h: forked ranks, a shared region, a process-shared
    barrier) on host buffers, world sizes 2, 3 and 8: reduce-scatter, all-gather, hit counters, barrier -- with slots smaller
    than a segment (the chunked path) and larger."""
    shim = _shim()
    for world, seg_d, seg_b, slot in ((2, 1000, 777, 1 << 20), (3, 1000, 777, 512), (8, 64, 4099, 1024), (2, 1, 1, 8)):
        assert shim.shim_shm_transport_selftest(world, seg_d, seg_b, slot) == 0, (world, seg_d, seg_b, slot)


def _plan(shim, ped, world):
    import ctypes as C
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    bounds = np.zeros(world + 1, np.int32)
    owner = np.zeros(ped.n_rec, np.int32)
    shared = np.zeros(ped.n_rec, np.uint8)
    dous = np.ascontiguousarray(ped.dous, np.int32)
    seg = shim.shim_partition(ped.n_rec, P(ped.par), P(ped.empty), P(ped.gen), P(ped.row_of), P(dous), len(dous), ped.n_markers,
                              world, P(bounds), P(owner), P(shared))
    assert seg >= 0, "two ranks claim a record"
    return bounds, owner, shared, seg


def test_partition_keeps_families_whole_and_shares_only_what_straddles():
    """Engine::plan (csrc/host/cnf2_partition.h) on the CPU, through the host shim.  BASELINE config 5's shape -- disjoint
    three-generation families -- must give blocks that cut between families: nothing shared, nothing to exchange.  A
    pedigree in which every cut splits a family (two big half-sib families) must share exactly the records both sides'
    windows touch, each owned by one of its touchers; the F2 design shares its two founders."""
    from cnf2freq_amd import synth
    shim = _shim()

    def touched_by(ped, bounds):
        """per rank the records its windows touch: the analysed individual, its parents and grandparents"""
        out = []
        for b, e in zip(bounds, bounds[1:]):
            t = set()
            for ind in ped.dous[b:e]:
                t.add(int(ind))
                for p in ped.par[ind]:
                    if p >= 0:
                        t.add(int(p))
                        t.update(int(g) for g in ped.par[p] if g >= 0)
            out.append(t)
        return out

    # (1) config 5's shape: 23 families of 3 analysed children over 2, 3, 4 and 8 ranks
    ped = synth.make_outbred3(23, 3, 5, 1, seed=4, missing=0.2)
    for world in (2, 3, 4, 8):
        bounds, owner, shared, seg = _plan(shim, ped, world)
        assert bounds[0] == 0 and bounds[-1] == len(ped.dous) and np.all(np.diff(bounds) >= 0)
        assert np.all(bounds % 3 == 0), "a block boundary falls inside a family: %s" % bounds
        assert shared.sum() == 0 and seg == 0
        sizes = np.diff(bounds)
        assert sizes.max() - sizes.min() <= 3 * 2, sizes             # balanced to within two families
        t = touched_by(ped, bounds)
        for q in range(world):
            assert set(np.flatnonzero(owner == q)) == t[q]
    # (2) one big family: every cut splits it -- the parents and grandparents are shared, the children private
    big = synth.make_outbred3(1, 12, 5, 1, seed=5, missing=0.1)
    bounds, owner, shared, seg = _plan(shim, big, 2)
    assert 0 < bounds[1] < 12
    t = touched_by(big, bounds)
    both = t[0] & t[1]
    assert set(np.flatnonzero(shared)) == both and len(both) == 6
    assert all(owner[r] in (0, 1) for r in both) and abs(int((owner[list(both)] == 0).sum()) - 3) <= 1   # owners balanced
    for q in range(2):
        assert set(np.flatnonzero((owner == q) & (shared == 0))) == t[q] - both
    assert seg == 3
    # (3) the F2 design: private F1 parents, the two founders shared by every rank
    f2 = synth.make_f2(40, 5, 1, seed=6)
    bounds, owner, shared, seg = _plan(shim, f2, 4)
    assert list(np.diff(bounds)) == [10, 10, 10, 10] and shared.sum() == 2 and seg == 1
    # (4) a mix: private families and one family that is bigger than the tolerance lets a boundary move
    mix = synth.make_outbred3(6, 2, 5, 1, seed=7, missing=0.1)
    bounds, owner, shared, seg = _plan(shim, mix, 2)
    assert bounds[1] == 6 and shared.sum() == 0


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
