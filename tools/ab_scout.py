"""Clean A/B of the update pass's variants on one box: the same deterministic run (CNF2_DETERMINISTIC accumulators: the
variants are bit-identical, so the trajectories are) repeated with different update forms (flags of cnf2h_set_update_flags);
per-iteration wall time of each.  usage: python tools/ab_scout.py [families=500] [snps=2500] [chroms=4] [iterations=30] [forms]
forms = comma list of mirror, both, both_one_scout, plain, mirror_literal_finish, both_literal_finish (default: both,both_one_scout)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from cnf2freq_amd import capi, host, synth

fams = int(sys.argv[1]) if len(sys.argv) > 1 else 500
snps = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
chroms = int(sys.argv[3]) if len(sys.argv) > 3 else 4
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
name = "form"
values = (sys.argv[5] if len(sys.argv) > 5 else "both,both_one_scout").split(",")
FORMS = {"mirror": 0, "both": capi.UPDATE_BOTH_FLOWS, "both_one_scout": capi.UPDATE_BOTH_FLOWS | capi.UPDATE_ONE_SCOUT,
         "plain": capi.UPDATE_PLAIN, "mirror_literal_finish": capi.UPDATE_LITERAL_FINISH,
         "both_literal_finish": capi.UPDATE_BOTH_FLOWS | capi.UPDATE_LITERAL_FINISH}
ped = synth.make_outbred3(fams, 4, snps, chroms, seed=2, missing=0.2)
times, states = {}, {}
for rep in range(2):
    for v in values:
        run = host.Run(ped)
        run.set_deterministic(True)
        run.set_update_flags(FORMS[v])
        run.postmarkerdata()
        t = []
        for it in range(iters):
            t0 = time.time()
            run.iteration(None)
            t.append(time.time() - t0)
        st = run.state()
        run.close()
        times.setdefault(v, []).append(np.array(t))
        states.setdefault(v, st)
        print("%s=%s rep %d: total %.2f s, iterations 2.. mean %.4f s, scale factor at the end %.6f" % (name, v, rep, sum(t), np.mean(t[1:]), st["scalefactor"]), flush=True)
ref = states[values[0]]
for v in values[1:]:
    same = all(np.array_equal(np.asarray(ref[k]), np.asarray(states[v][k]), equal_nan=True) for k in ("hw", "sure", "allele"))
    print("state of %s=%s identical to %s=%s: %s" % (name, v, name, values[0], same))
for v in values:
    best = np.minimum.reduce(times[v])
    print("%s=%s best-of-2 per iteration: total %.3f s; by tens: %s" % (name, v, best.sum(), [round(float(best[i:i + 10].sum()), 3) for i in range(0, iters, 10)]))
