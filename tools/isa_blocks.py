"""Static instruction mix per basic block of one kernel in a hipcc -S dump (tuning aid).
usage: python tools/isa_blocks.py k.s <mangled-kernel-name-substring>"""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l and ":" in l and "@" in l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))


def cls(op):
    if op.startswith("v_mov_b32_dpp") or "_dpp" in op: return "dpp"
    if op.startswith("ds_swizzle"): return "swz"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_load"): return "gld"
    if op.startswith("global_store"): return "gst"
    if op.startswith("scratch_"): return "scr"
    if op.startswith("v_") and "f64" in op: return "f64"
    if op.startswith("v_"): return "v32"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_"): return "s"
    return "other"


blocks, cur, name = [], Counter(), "entry"
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append((name, cur)); cur = Counter(); name = m.group(1)
        continue
    op = s.split()[0]
    cur[cls(op)] += 1
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        cur["->" + s.split()[-1]] += 0
blocks.append((name, cur))
tot = Counter()
for n, c in blocks:
    k = sum(v for kk, v in c.items() if not kk.startswith("->"))
    tot.update({kk: v for kk, v in c.items() if not kk.startswith("->")})
    if k >= 40:
        tg = [kk for kk in c if kk.startswith("->")]
        print("%-12s n=%4d " % (n, k), {kk: v for kk, v in sorted(c.items()) if not kk.startswith("->")}, tg)
print("total", sum(tot.values()), dict(tot))
