"""Measurement aid: kernel time of the sweep on non-F2 window shapes (BASELINE configs 3 and 5
shapes at reduced size), to show that the rate does not depend on the F2 special case.
Run on a GPU box: python tools/shape_timing.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from cnf2freq_amd import capi, synth

cases = {
    "F2 (config 2 shape), 2000 x 2501": lambda: synth.make_f2(2000, 2500, 1, seed=1),
    "outbred 3-gen, 20% missing (config 5 shape), 2000 kids x 2501": lambda: synth.make_outbred3(500, 4, 2500, 1, seed=2, missing=0.2),
    "advanced intercross (config 3 shape), 1600 ind x 2501": lambda: synth.make_ail(64, 200, 8, 2500, 1, seed=3),
}
if "--full" in sys.argv:
    # BASELINE configs 3 and 5 at their stated sizes (config 5: M = 10 000 as assumed in SURVEY section 8)
    cases = {
        "config 3: AIL 2 founders + 64 F1 + 8 x 625 analysed, 8 x 2500 SNPs (+dummies)":
            lambda: synth.make_ail(64, 625, 8, 2500, 8, seed=3),
        "config 5 shape: outbred 3-gen, 20% missing, 10 000 analysed x 10 000 SNPs (4 x 2500)":
            lambda: synth.make_outbred3(2500, 4, 2500, 4, seed=2, missing=0.2),
    }
for name, mk in cases.items():
    t0 = time.time()
    ped = mk()
    ctx = capi.Context(0)
    ctx.upload(ped)
    n = len(ped.dous)
    tied = sum(1 for j in range(0, n, max(1, n // 200)) if (ctx.window_info(j)["tie"] >= 0).any())
    ctx.sweep()
    ms = []
    for _ in range(3):
        ctx.sweep()
        ms.append(ctx.last_kernel_ms())
    units = n * ped.n_markers
    print("%-70s units %.3g  kernel %.1f ms  %.3g units/s   (gen %.0fs)" % (name, units, np.mean(ms), units / np.mean(ms) * 1e3, time.time() - t0), flush=True)
    ctx.close()
