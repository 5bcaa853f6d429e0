"""Worker of test_rccl_exchange_on_one_rank: ONE rank with the nccl (= RCCL) backend on GPU 0 -- all a one-GPU box can run
of the multi-GPU path.  A world of one skips the exchange in the product (dist.make_exchange), so the callback here calls
the same functions directly: torch tensors aliasing the engine's device slabs (dist.device_slabs) go through
dist.allreduce_accumulators on RCCL and must come back unchanged; a tiled gather of device rows runs through the same
backend.  Prints OK <checksum>."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from cnf2freq_amd import capi, host, synth
from cnf2freq_amd import dist as cdist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29577")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ped = synth.make_outbred3(3, 3, 12, 2, seed=12, missing=0.2)
run = host.Run(ped)
run.postmarkerdata()
seen = {}


def exchange(a, b, c, n_rec, n_markers):
    before = capi.Context.accumulators_of(run.context(), n_rec, n_markers)
    ts = cdist.device_slabs(a, b, c, n_rec, n_markers, torch.device("cuda", 0))
    works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True) for t in ts]     # what allreduce_accumulators does for world > 1
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    after = capi.Context.accumulators_of(run.context(), n_rec, n_markers)
    seen["same"] = all(np.array_equal(before[k], after[k], equal_nan=True) for k in before)
    seen["sum"] = float(sum(t.sum().item() for t in ts))
    return 0


run.set_exchange(exchange)
run.iteration()
assert seen.get("same") and seen["sum"] > 0, seen
# the streaming gather on the same backend (a world of one has no peer: the root consumes its own staged tile)
src = torch.arange(4 * 10 * 3, dtype=torch.float64, device="cuda").reshape(4, 10, 3)
got = []
tg = cdist.TiledGather(4, 10, 3, 4, torch.float64, torch.device("cuda", 0))
tg.run(src, lambda m0, m1, parts: got.append((m0, m1, parts[0].clone())))
assert torch.equal(torch.cat([g[2] for g in got], dim=1), src)
dist.barrier()
run.close()
dist.destroy_process_group()
print("OK %.9g" % seen["sum"])
