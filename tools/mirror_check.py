"""How far the mirrored certainty flows (default) move a run away from the form that runs both flows (CNF2_UPDATE_BOTH_FLOWS):
the same deterministic run both ways, states compared after every iteration.
usage: python tools/mirror_check.py [families=200] [snps=1000] [chroms=2] [iterations=5]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from cnf2freq_amd import capi, host, synth

fams = int(sys.argv[1]) if len(sys.argv) > 1 else 200
snps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
chroms = int(sys.argv[3]) if len(sys.argv) > 3 else 2
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ped = synth.make_outbred3(fams, 4, snps, chroms, seed=2, missing=0.2)
runs = {}
cap = {}
for name, form in (("mirror", 0), ("both", capi.UPDATE_BOTH_FLOWS)):
    r = host.Run(ped)
    r.set_deterministic(True)
    r.set_update_flags(form)
    r.postmarkerdata()
    if name == "both":
        # the accumulators of the first sweep, for the print-out below: a sweep without updates leaves them in the context
        r.iteration(None, update=False)
        cap["acc"] = capi.Context.accumulators_of(r.context(), ped.n_rec, ped.n_markers)
    runs[name] = r
prev = runs["both"].state()
a0, s0, _ = ped.dense()
for it in range(iters):
    st = {}
    for name in ("mirror", "both"):
        runs[name].iteration(None)
        st[name] = runs[name].state()
    for k in ("sure", "hw"):
        d = np.abs(np.asarray(st["mirror"][k]) - np.asarray(st["both"][k]))
        print("iteration %d %-5s max |diff| %.3e   > 1e-12: %d   > 1e-9: %d   > 1e-6: %d   of %d" % (it + 1, k, np.nanmax(d), (d > 1e-12).sum(), (d > 1e-9).sum(), (d > 1e-6).sum(), d.size), flush=True)
    print("iteration %d alleles differ at %d, hits %s / %s, scale factor %.9f / %.9f" % (it + 1, (np.asarray(st["mirror"]["allele"]) != np.asarray(st["both"]["allele"])).sum(), st["mirror"]["hits"], st["both"]["hits"], st["mirror"]["scalefactor"], st["both"]["scalefactor"]), flush=True)
    if it == 0:
        d = np.abs(np.asarray(st["mirror"]["sure"]) - np.asarray(st["both"]["sure"]))
        idx = np.argsort(d.ravel())[::-1][:12]
        for i in idx:
            r, m, side = np.unravel_index(i, d.shape)
            print("   inf", cap["acc"]["infprobs"][r, m], "children", int(sum(1 for q in ped.dous for k in range(2) if ped.par[q, k] == r)))
            print("   rec %d (gen %d, empty %d) marker %d side %d: before allele %s sure %s | read allele %s | mirror -> allele %s sure %s | both -> allele %s sure %s"
                  % (r, ped.gen[r], ped.empty[r], m, side, np.asarray(prev["allele"])[r, m], np.asarray(prev["sure"])[r, m], a0[r, m],
                     np.asarray(st["mirror"]["allele"])[r, m], np.asarray(st["mirror"]["sure"])[r, m],
                     np.asarray(st["both"]["allele"])[r, m], np.asarray(st["both"]["sure"])[r, m]), flush=True)
