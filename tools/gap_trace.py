"""Where the GPU waits during an iteration (tuning aid): the largest gaps between consecutive activities (kernels and copies) over
the last part of a rocprofv3 --kernel-trace --memory-copy-trace run, with what ran before and after each gap, and the copies by
direction.  usage: python tools/gap_trace.py <dir with *_kernel_trace.csv / *_memory_copy_trace.csv> [fraction=0.3]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
acts = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acts.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
copies = defaultdict(lambda: [0, 0.0, 0])
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = "copy " + r.get("Direction", r.get("Kind", "?"))
        acts.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
acts.sort()
t0, t1 = acts[0][0], max(a[1] for a in acts)
cut = t1 - frac * (t1 - t0)
late = [a for a in acts if a[0] >= cut]
busy, gaps, end = 0.0, [], late[0][0]
by = defaultdict(float)
for s, e, n in late:
    if s > end:
        gaps.append((s - end, prev, n))
        end = s
    if e > end:
        busy += e - end
        by[n] += e - end
        end = e
    prev = n
wall = (late[-1][1] - late[0][0]) / 1e6
print("last %.0f%%: %.1f ms wall, GPU busy %.1f ms (%.0f%%)" % (100 * frac, wall, busy / 1e6, 100 * busy / 1e6 / wall))
for n, v in sorted(by.items(), key=lambda kv: -kv[1])[:14]:
    print("   busy %-62s %9.1f ms" % (n, v / 1e6))
agg = defaultdict(lambda: [0, 0.0])
for g, a, b in gaps:
    k = (a, b)
    agg[k][0] += 1
    agg[k][1] += g
print("gaps by (before -> after), total ms:")
for (a, b), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("   %6d x %9.1f ms   %s -> %s" % (n, v / 1e6, a, b))
