"""Diagnosis aid (GPU box): how far the product drifts from a trajectory golden (G13) iteration by iteration -- hit counters,
scale factor, and the largest differences of certainties / haplotype weights / haplobase, continuing from the reference's own
state after postmarkerdata.  usage: python tools/diag_trajectory.py [case=outbred3_long] [form: mirror|both]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from conftest import load_trajectory, pedigree_components
from cnf2freq_amd import capi, host

case = sys.argv[1] if len(sys.argv) > 1 else "outbred3_long"
form = sys.argv[2] if len(sys.argv) > 2 else "mirror"
ped, z, n_iter = load_trajectory(case)
run = host.Run(ped)
run.set_update_flags(capi.UPDATE_BOTH_FLOWS if form == "both" else 0)
run.postmarkerdata()
path = os.path.join(tempfile.mkdtemp(), "pm.txt")
with open(path, "w") as f:
    for r in range(ped.n_rec):
        f.write("%d r%d\n" % (r + 1, r))
        for m in range(ped.n_markers):
            f.write("%.17g\t%d\t%d\t\t%f\t%.17g %.17g %f\n" % (z["pm_hw"][r, m], z["pm_allele"][r, m, 0], z["pm_allele"][r, m, 1], 0.0,
                                                           z["pm_sure"][r, m, 0], z["pm_sure"][r, m, 1], 0.5))
sys.stdout.flush()
saved = os.dup(1)
os.dup2(2, 1)
run.deserialize(path)
out = []
comp = pedigree_components(ped)
tainted = np.zeros(ped.n_rec, bool)
for k in range(1, n_iter + 1):
    run.iteration()
    st = run.state()
    ps = run.passes()
    flagged = z["it%d_unstable" % k].astype(bool)
    line = "it%2d hits %s / %s  sf %.12g / %.12g" % (k, ps["hits"].tolist(), z["it%d_hits" % k].tolist(), st["scalefactor"], float(z["it%d_scalefactor" % k]))
    for key, got in (("sure", st["sure"]), ("hw", st["hw"]), ("haplobase", ps["haplobase"])):
        want = z["it%d_%s" % (k, key)]
        d = np.abs(got - want)
        rel = d / np.maximum(np.abs(want), 1e-300)
        mask = ~flagged if key == "sure" else np.ones(d.shape, bool)
        line += "  %s: max abs %.2e (> 1e-9: %d, > 1e-6: %d of %d)" % (key, d[mask].max(), int((d[mask] > 1e-9).sum()), int((d[mask] > 1e-6).sum()), int(mask.sum()))
    line += "  alleles differ %d (unstable %d)" % (int((st["allele"] != z["it%d_allele" % k]).sum()), int(flagged.sum()))
    # the trajectory test's rule (tests/conftest.py TrajectoryChecker): an unstable element that differs taints its component
    differs = ~np.isclose(st["sure"], z["it%d_sure" % k], rtol=1e-9, atol=1e-10) | (st["allele"] != z["it%d_allele" % k])
    bad = (flagged & differs).any(axis=(1, 2))
    tainted |= np.isin(comp, np.unique(comp[bad]))
    ok = ~tainted
    line += "\n      untainted %d/%d:" % (int(ok.sum()), ped.n_rec)
    for key, got in (("sure", st["sure"]), ("hw", st["hw"]), ("haplobase", ps["haplobase"])):
        want = z["it%d_%s" % (k, key)]
        d = np.abs(got - want)[ok]
        rel = d / np.maximum(np.abs(want[ok]), 1e-12)
        line += "  %s max abs %.2e max rel %.2e" % (key, d.max(), rel.max())
    line += "  alleles differ %d" % int((st["allele"] != z["it%d_allele" % k])[ok].sum())
    out.append(line)
os.dup2(saved, 1)
print("\n".join(out))
run.close()
