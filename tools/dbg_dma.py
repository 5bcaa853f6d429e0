import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cnf2freq_amd import capi, synth
ped = synth.make_f2(300, 2500, 1, seed=2, chrom_cm=100.0)
ctx = capi.Context(0); ctx.upload(ped)
h = ctx.sweep()
f = ctx.sweep(full_spill=True)
d = h["dosage"]
neg = np.argwhere(d < 0)
print("neg count", len(neg), "first", neg[:10].tolist())
inds = np.unique(neg[:, 0]); print("individuals with negatives", len(inds), inds[:20])
ms = np.unique(neg[:, 1]); print("markers", ms[:40], "...", ms[-10:])
print("marker mod 8 hist", np.bincount(ms % 8, minlength=8))
print("marker mod 2 hist", np.bincount(neg[:,1] % 2, minlength=2))
if f is not None:
    df = np.abs(f["dosage"] - d).max(axis=2)
    bad = np.argwhere(df > 1e-9)
    print("half vs full mismatches", len(bad), bad[:10].tolist())
    print("full neg", (f["dosage"] < 0).sum())
print("most negative half", d.min(), "full", f["dosage"].min())
print("neg values sample", d[d < 0][:10])
