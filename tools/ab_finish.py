"""A/B of the update pass's finish kernels on one box in one call: the iteration probe of bench.py with the guided bisection
(default) and with one literal step per round (CNF2_UPDATE_LITERAL_FINISH), each form twice, alternating.
usage: python tools/ab_finish.py [families=500] [snps=2500] [chroms=4] [warmup=2] [timed=3]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import bench  # noqa: E402
import torch  # noqa: E402
from cnf2freq_amd import capi  # noqa: E402

a = [int(x) for x in args] + [500, 2500, 4, 2, 3][len(args):]
torch.cuda.set_device(0)
for rep in range(2):
    for name, flags in (("guided", 0), ("literal_finish", capi.UPDATE_LITERAL_FINISH)):
        r = bench.iteration_probe(0, torch.device("cuda", 0), fams=a[0], snps_per_chrom=a[1], chroms=a[2], warmup=a[3], timed=a[4], update_flags=flags, with_stats=False)
        print("%-15s iteration %.4f s  update %.4f s  sweep+acc %.4f s  scalefactor %.6f hits %s" %
              (name, r["iteration_s"], r["update_s"], r["sweep_accumulate_s"], r["scalefactor"], r["last_hits"]), flush=True)
