#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel sums over the dispatches seen (and per-unit values
when --units, the units of ONE dispatch, is given: sum / dispatches / units).
Usage: pmc_summarize.py <dir>... [--units N] [--kernel substr]"""
import collections
import csv
import glob
import sys

args = sys.argv[1:]
units = None
kern = "fb_"
dirs = []
i = 0
while i < len(args):
    if args[i] == "--units":
        units = float(args[i + 1]); i += 2
    elif args[i] == "--kernel":
        kern = args[i + 1]; i += 2
    else:
        dirs.append(args[i]); i += 1
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float)
        n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
        for k in sorted(agg):
            line = "%-28s %.6g (dispatch rows %d)" % (k, agg[k], n[k])
            if units:
                line += "   per unit %.4g" % (agg[k] / n[k] / units)
            print(line)
