"""Tuning aid: inputs of a sample of the update pass's flows in the steady state of a run (config 5's shape), for analysis on
the host (tests/shim: shim_flow_trace).  Runs `iterations` haplotyping iterations, captures the accumulators and the rows
the last one's first update pass starts from, and writes gpurun_out/flows_sample.npz.
usage: python tools/dump_flows.py [families=200] [snps_per_chrom=1000] [chroms=2] [iterations=36] [records=200]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from cnf2freq_amd import capi, host, synth

fams = int(sys.argv[1]) if len(sys.argv) > 1 else 200
snps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
chroms = int(sys.argv[3]) if len(sys.argv) > 3 else 2
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 36
nrec = int(sys.argv[5]) if len(sys.argv) > 5 else 200
ped = synth.make_outbred3(fams, 4, snps, chroms, seed=2, missing=0.2)
run = host.Run(ped)
run.postmarkerdata()
cap = {}
for it in range(iters - 1):
    run.iteration(None)
st = run.state()
# the accumulators the last iteration's updates will see: a sweep without updates leaves them in the context
run.iteration(None, update=False)
cap["acc"] = capi.Context.accumulators_of(run.context(), ped.n_rec, ped.n_markers)
run.iteration(None)
st2 = run.state()
rs = np.random.RandomState(1)
recs = np.sort(rs.choice(ped.n_rec, min(nrec, ped.n_rec), replace=False))
m1 = int(ped.chromstarts[1])
a0, s0, _ = ped.dense()
children = np.zeros(ped.n_rec, np.int32)
for r in ped.dous:
    for k in range(2):
        if ped.par[r, k] >= 0:
            children[ped.par[r, k]] += 1
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/flows_sample.npz", recs=recs, inf=cap["acc"]["infprobs"][recs, :m1], hb=cap["acc"]["haplobase"][recs],
                    hc=cap["acc"]["haplocount"][recs], allele=st["allele"][recs], sure=st["sure"][recs], hw=st["hw"][recs],
                    prior_allele=a0[recs], prior_sure=s0[recs], empty=ped.empty[recs], children=children[recs],
                    descendants=st["descendants"][recs], scalefactor=st["scalefactor"], chromstarts=ped.chromstarts,
                    allele_after=st2["allele"][recs], sure_after=st2["sure"][recs], hw_after=st2["hw"][recs])
print("scalefactor", st["scalefactor"], "hits", st["hits"], "records", len(recs))
