"""Diagnosis aid (GPU box): where lockhaplos (cnF2freq.cpp:3045-3081) of the product locks another marker than the reference
did in a trajectory golden, and how close the two markers' variances are.  usage: python tools/diag_lockhaplos.py [case]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from conftest import load_trajectory
from cnf2freq_amd import host

case = sys.argv[1] if len(sys.argv) > 1 else "outbred3_long"
ped, z, _ = load_trajectory(case)
run = host.Run(ped)
run.postmarkerdata()
st = run.state()
var = st["variances"]
cs = ped.chromstarts
for r in range(ped.n_rec):
    for c in range(len(cs) - 1):
        sl = slice(int(cs[c]), int(cs[c + 1]))
        mine = np.flatnonzero((st["hw"][r, sl] == 0) | (st["hw"][r, sl] == 1))
        ref = np.flatnonzero((z["pm_hw"][r, sl] == 0) | (z["pm_hw"][r, sl] == 1))
        if list(mine) != list(ref):
            v = var[r, sl]
            a, b = (int(mine[0]) if len(mine) else -1), (int(ref[0]) if len(ref) else -1)
            print("rec %3d chrom %d: product locks %3d (var %.17g), reference %3d (var %.17g), rel diff %.3e, max var %.17g at %d"
                  % (r, c, a, v[a] if a >= 0 else float("nan"), b, v[b] if b >= 0 else float("nan"),
                     (v[a] - v[b]) / max(abs(v[a]), 1e-300) if a >= 0 and b >= 0 else float("nan"), v.max(), int(v.argmax())))
run.close()
