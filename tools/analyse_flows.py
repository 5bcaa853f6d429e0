"""Tuning aid: replay the certainty flows of gpurun_out/flows_sample.npz (tools/dump_flows.py) on the host and classify them."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_host_shim  # noqa: E402
L = build_host_shim()
D = C.c_double
L.shim_certainty_flow_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, D, C.c_int, D, C.c_int, D, C.c_void_p, C.POINTER(C.c_int), C.POINTER(D), C.POINTER(D)]
z = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "flows_sample.npz"))
sf = float(z["scalefactor"])
inf, allele, sure, pa, ps, ch = z["inf"], z["allele"], z["sure"], z["prior_allele"], z["prior_sure"], z["children"]
nrec, M1 = inf.shape[:2]
rs = np.random.RandomState(0)
rows = []
log = np.zeros((64, 4))
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20000):
    r, m, side, v = rs.randint(nrec), rs.randint(M1), rs.randint(2), rs.randint(2)
    pair = np.ascontiguousarray(inf[r, m, side])
    why, res, g0 = C.c_int(0), D(0), D(0)
    n = L.shim_certainty_flow_trace(pair.ctypes.data_as(C.c_void_p), v, int(allele[r, m, side]), float(sure[r, m, side]),
                                    int(pa[r, m, side]) if not z["empty"][r] else 0, float(ps[r, m, side]), int(ch[r]), sf,
                                    log.ctypes.data_as(C.c_void_p), C.byref(why), C.byref(res), C.byref(g0))
    if n < 0:
        continue
    lg = log[:n]
    quads = int(lg[:, 1].sum())
    tq = lg[lg[:, 1] == 1, 2]
    y = abs((1 if allele[r, m, side] == v + 1 else 0) - sure[r, m, side]) if allele[r, m, side] else 0.5
    rows.append((n, quads, why.value, y, res.value, 1.0 / g0.value if g0.value else np.inf, (tq < 0.5).sum(), (tq > 2).sum(),
                 ((tq >= 0.5) & (tq <= 2)).sum(), pair[v] / pair.sum()))
a = np.array(rows)
print("flows", len(a), "sf", sf)
for w, name in ((1, "tolerance"), (2, "interval<1e-10"), (3, "51 steps/bounds"), (0, "no step")):
    k = a[:, 2] == w
    if not k.any():
        continue
    b = a[k]
    print("%-16s %5.1f%%  steps %.1f  quad steps %.1f (t<sf/2: %.1f, t>2sf: %.1f, between: %.1f)  |move| median %.2e  |G(orig)| median %.2e"
          % (name, 100 * k.mean(), b[:, 0].mean(), b[:, 1].mean(), b[:, 6].mean(), b[:, 7].mean(), b[:, 8].mean(),
             np.median(np.abs(b[:, 4] - b[:, 3])), np.median(np.abs(b[:, 5]))))
    for lo, hi in ((0, 1e-4), (1e-4, 1e-2), (1e-2, 0.5), (0.5, 0.99), (0.99, 0.9999), (0.9999, 1.1)):
        kk = (b[:, 3] >= lo) & (b[:, 3] < hi)
        if kk.any():
            print("      start in [%g, %g): %5.1f%% of these, quad steps %.1f, evidence share - belief median %.2e" %
                  (lo, hi, 100 * kk.mean(), b[kk, 1].mean(), np.median(np.abs(b[kk, 9] - b[kk, 3]))))

if len(sys.argv) > 3:
    # print a few example traces of the 51-step class
    shown = 0
    rs = np.random.RandomState(5)
    while shown < int(sys.argv[3]):
        r, m, side, v = rs.randint(nrec), rs.randint(M1), rs.randint(2), rs.randint(2)
        pair = np.ascontiguousarray(inf[r, m, side])
        why, res, g0 = C.c_int(0), D(0), D(0)
        n = L.shim_certainty_flow_trace(pair.ctypes.data_as(C.c_void_p), v, int(allele[r, m, side]), float(sure[r, m, side]),
                                        int(pa[r, m, side]) if not z["empty"][r] else 0, float(ps[r, m, side]), int(ch[r]), sf,
                                        log.ctypes.data_as(C.c_void_p), C.byref(why), C.byref(res), C.byref(g0))
        if n < 0 or why.value != 3:
            continue
        y = abs((1 if allele[r, m, side] == v + 1 else 0) - sure[r, m, side]) if allele[r, m, side] else 0.5
        if not (1e-4 < y < 1e-2):
            continue
        shown += 1
        print("flow: inf", pair, "v", v, "allele", allele[r, m, side], "sure", sure[r, m, side], "prior", pa[r, m, side], ps[r, m, side], "children", ch[r],
              "y", y, "G(orig)", 1 / g0.value, "result", res.value)
        for k in range(n):
            print("   step %2d mid %.12g kind %d t/sf %.4g G(mid) %.4g" % (k, log[k, 0], log[k, 1], log[k, 2], 1 / log[k, 3] if log[k, 3] else np.inf))
