"""Where a kernel's scratch traffic sits: scratch stores / loads per basic block of a hipcc -S dump, with the loop depth
the compiler's comments give (tuning aid).  usage: python tools/isa_scratch.py k.s <mangled-kernel-name-substring> [-v]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
lab, order, cnt, size, depth = "entry", ["entry"], {}, {}, {}
for i in range(start, end):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if m:
        lab = m.group(1)
        order.append(lab)
        d = re.search(r"Depth=(\d+)", m.group(2))
        depth[lab] = int(d.group(1)) if d else 0
    s = l.strip()
    if s and not s.startswith(";") and not s.startswith("."):
        size[lab] = size.get(lab, 0) + 1
    if "scratch_" in l:
        cnt.setdefault(lab, [0, 0, []])
        cnt[lab][0 if "store" in l else 1] += 1
        cnt[lab][2].append(s)
for k in order:
    if k in cnt:
        print("%-12s depth %d  stores %2d loads %2d  of %4d instructions" % (k, depth.get(k, 0), cnt[k][0], cnt[k][1], size.get(k, 0)))
        if "-v" in sys.argv:
            for s in cnt[k][2]:
                print("      ", s)
