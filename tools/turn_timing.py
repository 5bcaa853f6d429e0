"""Measurement aid: the batched turn scan (HOT LOOP 3, cnf2_sweep_turn_scan) on BASELINE config 5's shape, per-turn
log-sum-exp to a device buffer.  Prints wall time per call; run under rocprofv3 --kernel-trace --stats for the split
between the turn-scan sweep (fb_fast_kernel<true, 2>) and turn_rows_kernel.
usage: python tools/turn_timing.py [families] [snps per chromosome] [chromosomes] [individuals] [repeats] [full|lse] [valu]"""
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
nt = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
full = len(sys.argv) > 6 and sys.argv[6] == "full"
valu = capi.TURN_VALU if "valu" in sys.argv[6:] else 0
ped = synth.make_outbred3(fams, 4, snps, chroms, seed=2, missing=0.2)
n, M = len(ped.dous), ped.n_markers
nt = min(nt, n)
ctx = capi.Context(0)
ctx.upload(ped)
dev = torch.device("cuda", 0)
lse = torch.empty((nt, M, 128), dtype=torch.float64, device=dev)
raw = torch.empty((nt, M, 1024), dtype=torch.float64, device=dev) if full else None
for rep in range(reps):
    t = time.time()
    ctx._chk(ctx.L.cnf2_sweep_turn_scan(ctx.h, 0, nt, C.c_void_p(raw.data_ptr()) if full else None,
                                        C.c_void_p(lse.data_ptr()), capi.OUT_DEVICE | valu), "cnf2_sweep_turn_scan")
    ctx.sync()
    dt = time.time() - t
    print("turn scan %d: %.4f s for %d individuals x %d markers  %.3g units/s  checksum %.9g"
          % (rep, dt, nt, M, nt * M / dt, float(lse.sum().item())), flush=True)
