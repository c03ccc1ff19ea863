#!/usr/bin/env python
"""Sums rocprofv3 PMC counters (csv output) for the dispatches of ONE kernel:  tools/pmc_kernel.py <dir> <kernel-substr>"""
import csv, glob, os, sys
from collections import defaultdict
acc, n = defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:36s} {acc[k] / n[k]:16.0f}  (avg of {n[k]} dispatches)")
