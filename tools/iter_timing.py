"""Measurement aid: one haplotyping iteration of BASELINE config 5's shape (3-generation outbred, 20 % missing) on one
GPU, through the C ABI with everything device-resident: plain sweep, sweep + HOT LOOP 2 accumulators
(cnf2_sweep_accumulate), the update passes of every chromosome.
usage: python tools/iter_timing.py [families] [snps per chromosome] [chromosomes] [iterations] [scalefactor] [flow|plain] [det]
(det = CNF2_DETERMINISTIC accumulators)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cnf2freq_amd import capi, synth

fams = int(sys.argv[1]) if len(sys.argv) > 1 else 250
snps = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
chroms = int(sys.argv[3]) if len(sys.argv) > 3 else 2
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2
t0 = time.time()
ped = synth.make_outbred3(fams, 4, snps, chroms, seed=2, missing=0.2)
a, s, h = ped.dense()
ped.allele = np.concatenate([a[:1] * 0, a]).astype(np.uint8)
ped.sure = np.concatenate([s[:1] * 0, s])
ped.hw = np.concatenate([h[:1] * 0 + 0.5, h])
ped.row_of = np.arange(1, ped.n_rec + 1, dtype=np.int32)
del a, s, h
n, M, R = len(ped.dous), ped.n_markers, ped.n_rec
print("pedigree: %d records, %d analysed, %d markers (gen %.0f s)" % (R, n, M, time.time() - t0), flush=True)
ctx = capi.Context(0)
ctx.upload(ped)
ctx.snapshot_priors((1 - ped.empty).astype(np.uint8))
desc = ctx.descendants()
children = np.zeros(R, np.int32)
for r in ped.dous:
    for k in range(2):
        if ped.par[r, k] >= 0:
            children[ped.par[r, k]] += 1
dev = torch.device("cuda", 0)
f64 = torch.float64
factors = torch.empty((n, chroms, 8), dtype=f64, device=dev)
loglik = torch.empty((n, chroms), dtype=f64, device=dev)
dosage = torch.empty((n, M, 3), dtype=f64, device=dev)
inf = torch.empty((R, M, 2, 2), dtype=f64, device=dev)
hb = torch.empty((R, M), dtype=f64, device=dev)
hc = torch.empty((R, M), dtype=f64, device=dev)
hz = torch.empty((n, M, 2), dtype=f64, device=dev)
units = n * M
ctx.sweep_device(0, n, factors.data_ptr(), loglik.data_ptr(), dosage.data_ptr())
ctx.sync()
t = time.time()
ctx.sweep_device(0, n, factors.data_ptr(), loglik.data_ptr(), dosage.data_ptr())
ctx.sync()
t_sweep = time.time() - t
print("sweep              %.3f s  %.3g units/s" % (t_sweep, units / t_sweep), flush=True)
# batched turn scan (HOT LOOP 3) on the first individuals: per-turn log-sum-exp to a device buffer (1 KB per unit)
nt = min(n, 1000)
lse = torch.empty((nt, M, 128), dtype=f64, device=dev)
for rep in range(2):
    t = time.time()
    ctx._chk(ctx.L.cnf2_sweep_turn_scan(ctx.h, 0, nt, None, C.c_void_p(lse.data_ptr()), capi.OUT_DEVICE), "cnf2_sweep_turn_scan")
    ctx.sync()
    t_turn = time.time() - t
print("turn scan          %.3f s for %d individuals  %.3g units/s (%.1f x sweep per unit)   lse[0,0,0]=%.6f"
      % (t_turn, nt, nt * M / t_turn, (t_turn / (nt * M)) / (t_sweep / units), float(lse[0, 0, 0].item())), flush=True)
del lse
# the allocations of the accumulate sweep before the first iteration (what the engine does after its uploads)
t = time.time()
ctx._chk(ctx.L.cnf2_reserve_accumulate(ctx.h, 0, n, 0), "cnf2_reserve_accumulate")
print("reserve            %.3f s (batch buffer of the accumulate sweep: a first hipMalloc of half the free memory)" % (time.time() - t), flush=True)
sf = float(sys.argv[5]) if len(sys.argv) > 5 else 0.013
upd_flags = capi.ACC_DEVICE | (capi.UPDATE_PLAIN if len(sys.argv) > 6 and sys.argv[6] == "plain" else 0)
acc_flags = capi.DETERMINISTIC if len(sys.argv) > 7 and sys.argv[7] == "det" else 0
for it in range(iters):
    t = time.time()
    ctx.sweep_accumulate_device(desc, 0, n, factors.data_ptr(), loglik.data_ptr(), dosage.data_ptr(), inf.data_ptr(),
                                hb.data_ptr(), hc.data_ptr(), hz.data_ptr(), acc_flags)
    ctx.sync()
    t_acc = time.time() - t
    t = time.time()
    hits_total = 0
    for c in range(chroms):
        hits = np.zeros(1, np.int32)
        ctx._chk(ctx.L.cnf2_update_pass(ctx.h, c, children.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p),
                                        C.c_void_p(inf.data_ptr()), C.c_void_p(hb.data_ptr()), C.c_void_p(hc.data_ptr()),
                                        sf, 1.0, hits.ctypes.data_as(C.c_void_p), upd_flags), "cnf2_update_pass")
        hits_total += int(hits[0])
    t_upd = time.time() - t
    print("iteration %d: sweep+accumulate %.3f s (%.2f x sweep, %.3g units/s)   update passes %.3f s   hits %d   "
          "loglik sum %.6f" % (it, t_acc, t_acc / t_sweep, units / t_acc, t_upd, hits_total, float(loglik.sum().item())),
          flush=True)
ctx.close()
