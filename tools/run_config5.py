"""BASELINE config 5 end to end on one GPU: 3-generation outbred pedigree, 20 % missing genotypes, 10 000 analysed
individuals (2 500 families of 4 grandparents, 2 parents, 4 children = 25 000 individuals), M = 4 x 2 500 SNPs + 4 dummy
markers (M is not given in BASELINE.json; 10 000 as assumed in SURVEY.md section 8), 100 haplotyping iterations:
postmarkerdata, then `iterations` rounds of sweep + HOT LOOP 2 accumulators + parameter updates, through libcnf2host.so
(the engine of the `cnF2freq` executable; rows and dumps of the non-final rounds are not formatted: --rows-last-only /
--dump-last-only semantics).  Prints one JSON line with the timings.
usage: python tools/run_config5.py [families=2500] [snps_per_chrom=2500] [chroms=4] [iterations=100]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from cnf2freq_amd import host, synth

fams = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
snps = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
chroms = int(sys.argv[3]) if len(sys.argv) > 3 else 4
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 100
budget = float(sys.argv[5]) if len(sys.argv) > 5 else 1e9      # wall-clock seconds after which no further iteration starts
t0 = time.time()
ped = synth.make_outbred3(fams, 4, snps, chroms, seed=2, missing=0.2)
n, M, R = len(ped.dous), ped.n_markers, ped.n_rec
print("pedigree: %d individuals, %d analysed, %d markers (generated in %.0f s)" % (R, n, M, time.time() - t0), flush=True)
t = time.time()
run = host.Run(ped)
t_up = time.time() - t
t = time.time()
run.postmarkerdata()
t_pm = time.time() - t
t = time.time()
run.reserve()               # the batch buffers of the iterations now (a first hipMalloc of ~100 GB takes seconds on a fresh device)
t_res = time.time() - t
print("upload %.1f s, postmarkerdata %.1f s, reserve %.1f s" % (t_up, t_pm, t_res), flush=True)
s0 = run.state()
acc0 = synth.dosage_accuracy(ped, s0)          # the withheld genotypes against the generator's truth, before any iteration
print("withheld genotypes before the iterations: %s" % json.dumps(acc0), flush=True)
locked = int(((s0["hw"] == 0) | (s0["hw"] == 1)).sum())
t_it = []
for it in range(iters):
    if time.time() - t0 > budget:
        print("wall-clock budget reached after %d iterations" % it, flush=True)
        break
    t = time.time()
    run.iteration(None)
    t_it.append(time.time() - t)
    if it < 3 or (it + 1) % 10 == 0:
        print("iteration %d: %.2f s" % (it + 1, t_it[-1]), flush=True)
t = time.time()
s1 = run.state()
t_pull = time.time() - t
free = (s0["hw"] > 0) & (s0["hw"] < 1)
out = {
    "config": "BASELINE config 5: outbred 3-generation, 20%% missing, %d analysed of %d individuals, %d markers, %d iterations, 1 GPU"
              % (n, R, M, len(t_it)),
    "iterations_requested": iters, "iterations_done": len(t_it), "units_per_iteration": n * M,
    "upload_s": t_up, "postmarkerdata_s": t_pm, "reserve_s": t_res, "iteration_s_mean": float(np.mean(t_it)), "iteration_s_first": t_it[0],
    "iterations_total_s": float(np.sum(t_it)), "state_download_s": t_pull,
    "units_per_s_per_iteration": n * M / float(np.mean(t_it)),
    "locked_weights": locked,
    "mean_abs_haploweight_move": float(np.abs(s1["hw"] - s0["hw"])[free].mean()),
    "phased_fraction": float((np.abs(s1["hw"][free] - 0.5) > 0.4).mean()),
    "genotypes_imputed": int(((s0["allele"] == 0) & (s1["allele"] != 0)).sum()),
    "scalefactor_end": s1["scalefactor"],
    # the 20 % of genotypes withheld from the input, against the generator's truth: after postmarkerdata (inference from
    # relatives only) and after the iterations
    "withheld_before": acc0, "withheld_after": synth.dosage_accuracy(ped, s1),
}
print(json.dumps(out), flush=True)
run.close()
