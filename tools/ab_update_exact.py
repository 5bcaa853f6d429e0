"""Evidence that the fast update kernels (scout / finish: root memo, bounds that spare quadratures) make the decisions of
the literal algorithm: the same run twice -- once with them, once with CNF2_UPDATE_PLAIN (one thread per element, every
quadrature) -- with accumulators added in a fixed order (CNF2_DETERMINISTIC), states compared bit for bit after every
iteration.
usage: python tools/ab_update_exact.py [families=200] [snps_per_chrom=1000] [chroms=2] [iterations=30]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import torch  # noqa: F401

from cnf2freq_amd import capi, host, synth

fams = int(sys.argv[1]) if len(sys.argv) > 1 else 200
snps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
chroms = int(sys.argv[3]) if len(sys.argv) > 3 else 2
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
ped = synth.make_outbred3(fams, 4, snps, chroms, seed=2, missing=0.2)
runs = {}
for name in ("fast", "literal"):
    r = host.Run(ped)
    r.set_deterministic(True)
    # the exact comparison is with both values' flows run (the mirror shortcut off) against the literal kernels
    r.set_update_flags(capi.UPDATE_BOTH_FLOWS if name == "fast" else capi.UPDATE_PLAIN)
    r.postmarkerdata()
    runs[name] = r
flows = 0
t = {"fast": 0.0, "literal": 0.0}
for it in range(iters):
    st = {}
    for name, r in runs.items():
        t0 = time.time()
        r.iteration(None)
        t[name] += time.time() - t0
        st[name] = r.state()
        st[name].update(r.passes())
    a, b = st["fast"], st["literal"]
    same = all(np.array_equal(a[k], b[k], equal_nan=True) for k in ("allele", "sure", "hw", "haplobase", "haplocount", "hits")) \
        and a["scalefactor"] == b["scalefactor"]
    flows += ped.n_rec * ped.n_markers * 4 + ped.n_rec * sum(int(ped.chromstarts[c + 1]) for c in range(chroms))
    print("iteration %2d  scale factor %.4f  hits %s  identical %s" % (it + 1, a["scalefactor"], a["hits"].tolist(), same), flush=True)
    if not same:
        for k in ("allele", "sure", "hw", "haplobase", "haplocount"):
            d = np.argwhere(~((a[k] == b[k]) | (np.isnan(a[k].astype(float)) & np.isnan(b[k].astype(float)))))
            print("   ", k, len(d), "differences, first", d[:3].tolist())
            for idx in d[:6]:
                idx = tuple(idx)
                print("        ", idx, "fast", repr(a[k][idx]), "literal", repr(b[k][idx]), "|", "allele", a["allele"][idx[0], idx[1]], b["allele"][idx[0], idx[1]],
                      "sure", a["sure"][idx[0], idx[1]], b["sure"][idx[0], idx[1]])
        sys.exit(1)
print("%d iterations, ~%.3g flows each way: identical to the bit; time in iterations: fast %.1f s, literal %.1f s" % (iters, flows, t["fast"], t["literal"]))
