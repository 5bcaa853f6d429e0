"""Kernel time by name over the LAST part of a rocprofv3 --kernel-trace of a long run (tuning aid: the update pass's
regime changes over a run's iterations).  usage: python tools/late_kernels.py <kernel_trace.csv> [fraction=0.15]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - frac * (t1 - t0)
agg, n = defaultdict(float), defaultdict(int)
for r in rows:
    if int(r["Start_Timestamp"]) >= cut:
        agg[r["Kernel_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        n[r["Kernel_Name"]] += 1
tot = sum(agg.values())
print("last %.0f%% of the trace: %.1f ms of kernels in %.1f ms of wall time" % (100 * frac, tot, (t1 - cut) / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:16]:
    print("   %-72s %5d %9.1f ms" % (k[:72], n[k], v))
