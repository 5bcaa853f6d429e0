"""Worker of tests/test_dist_gpu.py: one rank of a multi-process sweep.  Launched through
`python -m torch.distributed.run`; every rank sweeps its block of analysed individuals with the HIP library
(cnf2freq_amd.capi -> libcnf2hip.so) on the GPU and the results travel to rank 0 through cnf2freq_amd.dist
(block gather of the likelihoods, marker-tiled streaming gather of the posterior rows).  Rank 0 writes them
to an .npz.  usage: dist_worker.py OUT.npz BACKEND TILE_MARKERS"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from cnf2freq_amd import capi, synth
from cnf2freq_amd import dist as cdist


def make_ped():
    # F2 + explicit-parent intercross in one pedigree would need two files; the F2 with missing data and three
    # ragged chromosomes is enough to catch any mix-up of individual blocks or marker tiles
    ped = synth.make_f2(11, 37, 1, seed=21, chrom_cm=40.0, missing=0.1)
    ped.chromstarts = np.array([0, 5, 19, 38], np.int32)
    ped.pos = np.concatenate([np.arange(5) * 1.1, np.arange(14) * 0.6, np.arange(19) * 0.9])
    return ped


def main():
    out, backend, tile = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    ped = make_ped()
    ctx = capi.Context(local)
    ctx.upload(ped)                       # ancestors' rows are replicated on every rank
    n = len(ped.dous)
    i0, i1 = cdist.shard_range(n, rank, world)
    # equal block shapes for the collective: pad the shorter blocks (balanced to within one individual)
    nb = (n + world - 1) // world
    dev = torch.device("cuda", local)
    M, C = ped.n_markers, len(ped.chromstarts) - 1
    factors = torch.zeros((nb, C, 8), dtype=torch.float64, device=dev)
    loglik = torch.zeros((nb, C), dtype=torch.float64, device=dev)
    dosage = torch.zeros((nb, M, 3), dtype=torch.float64, device=dev)
    if i1 > i0:
        ctx.sweep_device(i0, i1, factors.data_ptr(), loglik.data_ptr(), dosage.data_ptr(), 0)
    ctx.sync()
    staged = backend == "gloo"            # gloo moves host memory: rehearsal of the same control flow
    gdev = torch.device("cpu") if staged else dev
    ll_parts = cdist.gather_to_root(loglik.cpu() if staged else loglik, 0)
    f_parts = cdist.gather_to_root(factors.cpu() if staged else factors, 0)
    tg = cdist.TiledGather(nb, M, 3, tile, torch.float64, gdev, dst=0, depth=2)
    full = np.zeros((world, nb, M, 3)) if rank == 0 else None
    seen = []

    def consume(m0, m1, parts):
        seen.append((m0, m1))
        for r, t in enumerate(parts):
            full[r, :, m0:m1] = t.cpu().numpy()

    src = dosage.cpu() if staged else dosage
    tg.run(src, consume)
    # a haplotyping sweep with its accumulators: every rank adds its individuals' share to per-record slabs on its
    # own GPU (family members of other ranks' individuals included), one all-reduce sums them
    desc = ctx.descendants()
    R = ped.n_rec
    inf = torch.zeros((R, M, 2, 2), dtype=torch.float64, device=dev)
    hb = torch.zeros((R, M), dtype=torch.float64, device=dev)
    hc = torch.zeros((R, M), dtype=torch.float64, device=dev)
    hz = torch.zeros((nb, M, 2), dtype=torch.float64, device=dev)
    if i1 > i0:
        ctx.sweep_accumulate_device(desc, i0, i1, factors.data_ptr(), loglik.data_ptr(), dosage.data_ptr(), inf.data_ptr(),
                                    hb.data_ptr(), hc.data_ptr(), hz.data_ptr())
    ctx.sync()
    acc = [t.cpu() if staged else t for t in (inf, hb, hc)]
    for t in acc:                                   # caller-owned slabs (CNF2_ACC_DEVICE): the caller sums them as it likes
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        sizes = [cdist.shard_range(n, r, world) for r in range(world)]
        dos = np.concatenate([full[r, :b - a] for r, (a, b) in enumerate(sizes)])
        ll = np.concatenate([p.cpu().numpy()[:b - a] for p, (a, b) in zip(ll_parts, sizes)])
        fa = np.concatenate([p.cpu().numpy()[:b - a] for p, (a, b) in zip(f_parts, sizes)])
        np.savez(out, dosage=dos, loglik=ll, factors=fa, tiles=np.array(seen), root_bytes=tg.root_bytes(),
                 tile_bytes=nb * tile * 3 * 8, world=world, acc_infprobs=acc[0].cpu().numpy(),
                 acc_haplobase=acc[1].cpu().numpy(), acc_haplocount=acc[2].cpu().numpy())
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
