#!/usr/bin/env python3
"""Condense rocprofv3 --pmc output (…_counter_collection.csv files under a directory) into kernel,counter,dispatches,mean.
usage: summarize_pmc.py <rocprof output dir> [<dir> ...] > summary.csv"""
import csv, glob, os, re, sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = re.split(r"[<(]", k)[0].split("::")[-1].strip()
            a = acc[(k, row["Counter_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
print("kernel,counter,dispatches,mean_per_dispatch")
for (k, c), (n, s) in sorted(acc.items()):
    print(f"{k},{c},{n},{s / n:.6g}")
