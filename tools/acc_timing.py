"""Measurement aid: kernel time of sweep + HOT LOOP 2 accumulators (cnf2_sweep_accumulate, accumulators kept in the context)
against the plain sweep of the same windows, on BASELINE config 5's shape.  With CNF2HIP_LIB pointing at a timing-ablation
build (e.g. -DCNF2_X_FUSEDACC: results wrong) only the times mean anything.
usage: python tools/acc_timing.py [families=500] [snps=2500] [chroms=4] [batch_jobs=families*chroms]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import numpy as np
import torch
import bench
from cnf2freq_amd import capi

a = [int(x) for x in args] + [500, 2500, 4][len(args):3]
fams, snps, chroms = a[:3]
batch = int(args[3]) if len(args) > 3 else fams * chroms
dev = torch.device("cuda", 0)
ped = bench.generate_outbred_on_gpu(fams, 4, snps, chroms, 2, 0.2, dev)
ctx = capi.Context(0)
ctx.upload(ped)
ctx.set_batch_jobs(batch)
desc = np.ones(ped.n_rec, np.int32)
n, M = len(ped.dous), ped.n_markers
f = torch.empty((n, chroms, 8), dtype=torch.float64, device=dev)
ll = torch.empty((n, chroms), dtype=torch.float64, device=dev)
dos = torch.empty((n, M, 3), dtype=torch.float64, device=dev)
plain, acc = [], []
for _ in range(3):
    ctx.sweep_device(0, n, f.data_ptr(), ll.data_ptr(), dos.data_ptr(), 0)
    ctx.sync()
    plain.append(ctx.last_kernel_ms())
    ctx.sweep_accumulate_keep(desc)
    ctx.sync()
    acc.append(ctx.last_kernel_ms())
print("%s: %d analysed x %d markers, batches of %d jobs: plain sweep %.2f ms, sweep + accumulators %.2f ms = %.3f x"
      % (os.path.basename(os.environ.get("CNF2HIP_LIB", "libcnf2hip.so")), n, M, batch, min(plain), min(acc), min(acc) / min(plain)))
ctx.close()
