"""Worker of tests/test_dist_gpu.py::test_two_rank_iterations_equal_single_rank: one rank of a multi-process haplotyping
run (BASELINE config 5's shape).  Launched through `python -m torch.distributed.run`; every rank holds the pedigree, sweeps
its block of analysed individuals on the GPU, the accumulators of the records both ranks' windows touch meet in one
reduce-scatter per iteration, every rank updates the records it owns and the shared records' new rows are all-gathered
(cnf2freq_amd.dist.start_iterations -> libcnf2host.so: cnf2h_set_partition).
usage: dist_iter_worker.py OUT_PREFIX BACKEND N_ITER [deterministic=0]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from cnf2freq_amd import dist as cdist
from cnf2freq_amd import synth


def make_ped():
    # 3 families x 6 analysed children: with two ranks the split falls inside family 1, which is too big for the boundary
    # to move around it (its parents and grandparents collect evidence from both ranks: shared records); families 0 and 2
    # stay whole on their rank (private records: nothing of them is exchanged); two chromosomes
    ped = synth.make_outbred3(3, 6, 17, 2, seed=77, missing=0.2)
    return ped


def main():
    prefix, backend, n_iter = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    if world > 1 or backend == "nccl":
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    ped = make_ped()
    run = cdist.start_iterations(ped, device=local, deterministic=len(sys.argv) > 4 and sys.argv[4] == "1")
    hits = []
    for k in range(n_iter):
        rows = "%s_rows_it%d_rank%d.txt" % (prefix, k + 1, rank)
        if os.path.exists(rows):
            os.remove(rows)
        run.iteration(rows)
        hits.append(run.passes(accumulators=False)["hits"])
    st = run.state()
    ps = run.passes()
    np.savez("%s_rank%d.npz" % (prefix, rank), allele=st["allele"], sure=st["sure"], hw=st["hw"], hits=np.array(hits),
             scalefactor=st["scalefactor"], haplobase=ps["haplobase"], haplocount=ps["haplocount"], block=np.array(run.block),
             owned=run.plan["owned"], n_shared=run.plan["n_shared"], bytes_payload=run.plan["bytes_payload"],
             bytes_buffers=run.plan["bytes_accumulators"] + run.plan["bytes_rows"] + run.plan["bytes_hits"],
             bytes_moved=run.transport.bytes_moved, calls=np.array([run.transport.calls[k] for k in (0, 1, 2, 4)]))
    run.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
