"""The drop-in executable on files of a real size (GPU box): an outbred 3-generation pedigree is written as PlantImpute
map / ped / gen files and `cnF2freq` runs on them with CNF2_TIMING=1 -- what a user of the reference does -- so that the time
of the file reader, the set-up, the iterations and the output writer can be read off one log.
usage: python tools/cli_scale.py <families> <kids> <snps-per-chrom> <chroms> <rounds> [gpus]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cnf2freq_amd import synth  # noqa: E402

fam, kids, snps, chroms, rounds = (int(x) for x in sys.argv[1:6])
gpus = int(sys.argv[6]) if len(sys.argv) > 6 else 1
t0 = time.time()
ped = synth.make_outbred3(fam, kids, snps, chroms, seed=5, missing=0.2, fast=True)
d = tempfile.mkdtemp(prefix="cnf2cli_")
M = ped.n_markers
with open(os.path.join(d, "x.map"), "w") as f:
    f.write("\n".join("%.10g" % p for p in ped.pos) + "\n")
with open(os.path.join(d, "x.ped"), "w") as f:
    for r in range(ped.n_rec):
        p0, p1 = (ped.names[q] if q >= 0 else "0" for q in ped.par[r])
        f.write("%s %s %s %d\n" % (ped.names[r], p0, p1, ped.gen[r]))
# genotype tokens 0 / 1 / 2 / 9 from the allele pairs (1,1) (1,2) (2,2) (0,0): one character and a blank per marker
code = np.full((3, 3), ord("9"), np.uint8)
code[1, 1], code[1, 2], code[2, 1], code[2, 2] = ord("0"), ord("1"), ord("1"), ord("2")
with open(os.path.join(d, "x.gen"), "wb") as f:
    for r in range(ped.n_rec):
        a = ped.allele[ped.row_of[r]]
        line = np.full(2 * M, ord(" "), np.uint8)
        line[1::2] = code[a[:, 0], a[:, 1]]
        f.write(ped.names[r].encode() + line.tobytes() + b"\n")
size = sum(os.path.getsize(os.path.join(d, n)) for n in ("x.map", "x.ped", "x.gen"))
print("pedigree: %d individuals (%d analysed) x %d markers; files %.1f MB, written in %.1f s" % (ped.n_rec, len(ped.dous), M, size / 1e6,
                                                                                                time.time() - t0), flush=True)
out = os.path.join(d, "out.txt")
cmd = [os.path.join(ROOT, "cnf2freq_amd", "cnF2freq"), "--mapfile", d + "/x.map", "--pedfile", d + "/x.ped", "--genfile", d + "/x.gen",
       "--output", out, "--count", str(rounds), "--quiet", "--dump-last-only", "--tmppath", d]
if gpus > 1:
    cmd += ["--gpus", str(gpus), "--single-device"]
env = dict(os.environ, CNF2_TIMING="1")
t1 = time.time()
r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
wall = time.time() - t1
keep = [l for l in r.stderr.split("\n") if l.startswith("  [") or "ranks" in l]
laps = {}
for l in keep:
    if l.startswith("  [iteration]") or l.startswith("  [postmarkerdata]"):
        k = l.split("]")[0] + "] " + " ".join(l.split("]")[1].split()[:-2])
        laps[k] = laps.get(k, 0.0) + float(l.split()[-2])
    else:
        print(l)
for k, v in laps.items():
    print("%-60s %8.3f s (sum)" % (k, v))
print("exit %d, wall %.1f s, output %.1f MB" % (r.returncode, wall, os.path.getsize(out) / 1e6 if os.path.exists(out) else 0))
for n in os.listdir(d):
    os.remove(os.path.join(d, n))
os.rmdir(d)
sys.exit(r.returncode)
