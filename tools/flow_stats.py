"""Tuning aid: what the persistent flow kernels of the update pass do in the steady state of a haplotyping run
(config 5's shape): per iteration the time and, for the last chromosome's pass, flows, bisection steps per flow and lane
utilisation (cnf2_update_stats).
usage: python tools/flow_stats.py [families=500] [snps_per_chrom=2500] [chroms=4] [iterations=30]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.environ.get("CNF2_NO_STATS"):
    os.environ.setdefault("CNF2_UPDATE_STATS", "1")       # the kernels' diagnostics (a few atomics per wavefront)
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: before libcnf2hip.so)

from cnf2freq_amd import capi, host, synth

fams = int(sys.argv[1]) if len(sys.argv) > 1 else 500
snps = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
chroms = int(sys.argv[3]) if len(sys.argv) > 3 else 4
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
ped = synth.make_outbred3(fams, 4, snps, chroms, seed=2, missing=0.2)
run = host.Run(ped)
if os.environ.get("CNF2_AB_FLAGS"):
    run.set_update_flags(int(os.environ["CNF2_AB_FLAGS"]))       # e.g. 524288 = CNF2_UPDATE_LITERAL_FINISH
run.postmarkerdata()
L = capi.load()
for it in range(iters):
    t = time.time()
    run.iteration(None)
    dt = time.time() - t
    out = np.zeros(24, np.uint64)
    L.cnf2_update_stats(run.context(), out.ctypes.data_as(C.c_void_p))
    L.cnf2_update_stats_guided(run.context(), out[16:].ctypes.data_as(C.c_void_p))
    st = run.L.cnf2h_get_state  # noqa: F841
    sf = C.c_double(0)
    hits = C.c_int(0)
    run.L.cnf2h_get_state(run.h, None, None, None, None, None, None, C.byref(sf), C.byref(hits))
    c, h = out[:4].astype(float), out[4:8].astype(float)
    cq, hq = out[8:12].astype(float), out[12:16].astype(float)
    print("it %2d  %.3f s  sf %.4f hits %8d" % (it + 1, dt, sf.value, hits.value), flush=True)
    for name, a, b in (("certainty  ", c, cq), ("haploweight", h, hq)):
        flows, todo = a[0], a[0] - a[2] - a[3]
        print("       %s: %.3g flows: %.0f%% pinned, %.0f%% ended in the scout (%.1f evaluations per scouted flow), %.0f%% set aside: "
              "%.1f steps and %.2f quadratures each, lanes %.2f, %.0f%% ended by the tolerance"
              % (name, flows, 100 * a[3] / max(flows, 1), 100 * a[2] / max(flows, 1), a[1] / max(flows - a[3], 1), 100 * todo / max(flows, 1),
                 b[0] / max(todo, 1), b[2] / max(todo, 1), b[0] / max(b[1], 1), 100 * b[3] / max(todo, 1)), flush=True)
    # the guided kernels (last chromosome's pass): lock-step rounds [16..23], persistent tail [8..15]
    for name, a, b in (("certainty  ", out[16:20].astype(float), out[8:12].astype(float)), ("haploweight", out[20:24].astype(float), out[12:16].astype(float))):
        print("       %s guided: lock-step kernels %.3g points in %.3g lane-rounds (%.3g evaluations, %.3g flows ended by the tolerance); "
              "persistent tail %.3g points in %.3g lane-rounds (%.3g evaluations)" % (name, a[0], a[1], a[2], a[3], b[0], b[1], b[2]), flush=True)
run.close()
