"""Worker of test_rccl_exchange_on_one_rank: ONE rank with the nccl (= RCCL) backend on GPU 0 -- all a one-GPU box can run
of the multi-GPU path.  A world of one has nothing to exchange in the product, so this worker drives the transport
(cnf2freq_amd.dist.Transport, skip_single=False) by hand on the context's exchange buffer: the reduce-scatter of packed
accumulators and the all-gather of packed rows run through RCCL on tensors that alias the engine's device buffer and must
leave it unchanged (one segment = the whole buffer); the hit counters go through an RCCL all-reduce; a tiled gather of
device rows runs through the same backend.  Prints OK <checksum>."""
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
ctx = capi.Context(0)
ctx.upload_for_updates(ped)
ctx.sweep_accumulate_keep(ctx.descendants())
M = ped.n_markers
recs = np.arange(0, ped.n_rec, 2, dtype=np.int32)
S, B = M * 6, ((M * 25 + 7) // 8) * 8
buf = ctx.exchange_buffer(len(recs) * max(S * 8, B))
T = cdist.Transport(0, ctx.h, skip_single=False)
dev = torch.device("cuda", 0)
# accumulators: pack, reduce-scatter over a world of one (in place, through RCCL), compare
ctx.pack_accumulators(recs, buf)
before = cdist.device_view(buf, len(recs) * S, torch.float64, dev).clone()
assert T(host.X_SUM_SEGMENTS, buf, len(recs) * S, len(recs) * S) == 0
after = cdist.device_view(buf, len(recs) * S, torch.float64, dev)
assert torch.equal(before, after) and float(before.abs().sum()) > 0
checksum = float(before.sum())
# rows: pack, all-gather, compare
ctx.pack_rows(recs, buf)
before = cdist.device_view(buf, len(recs) * B, torch.uint8, dev).clone()
assert T(host.X_GATHER_SEGMENTS, buf, len(recs) * B, len(recs) * B) == 0
assert torch.equal(before, cdist.device_view(buf, len(recs) * B, torch.uint8, dev))
# hit counters
hits = np.array([3, 4], np.int32)
assert T(host.X_SUM_HITS, hits.ctypes.data, 2, 2) == 0 and hits.tolist() == [3, 4]
assert T.calls == {0: 1, 1: 1, 2: 1, 3: 0, 4: 0}
ctx.close()
# a whole iteration with a (trivial) partition through the host library: world of one, no exchange
run = cdist.start_iterations(ped, device=0)
run.iteration()
assert run.plan["n_shared"] == 0 and run.transport.bytes_moved == 0
run.close()
# the streaming gather on the same backend (a world of one has no peer: the root consumes its own staged tile)
src = torch.arange(4 * 10 * 3, dtype=torch.float64, device="cuda").reshape(4, 10, 3)
got = []
tg = cdist.TiledGather(4, 10, 3, 4, torch.float64, torch.device("cuda", 0))
tg.run(src, lambda m0, m1, parts: got.append((m0, m1, parts[0].clone())))
assert torch.equal(torch.cat([g[2] for g in got], dim=1), src)
dist.barrier()
dist.destroy_process_group()
print("OK %.9g" % checksum)
