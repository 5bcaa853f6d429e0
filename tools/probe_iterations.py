"""The iteration probe of bench.py on its own (for rocprofv3): python3 tools/probe_iterations.py [families=500] [snps=2500] [chroms=4] [warmup=2] [timed=3]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import bench  # noqa: E402
import torch  # noqa: E402

a = [int(x) for x in args] + [500, 2500, 4, 2, 3][len(args):]
torch.cuda.set_device(0)
print(json.dumps(bench.iteration_probe(0, torch.device("cuda", 0), fams=a[0], snps_per_chrom=a[1], chroms=a[2], warmup=a[3], timed=a[4], with_stats=False), indent=1))
