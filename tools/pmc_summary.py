#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, counters summed over dispatches / dispatch count."""
import collections
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        n = len(disp[k])
        print(k, f"dispatches={n}")
        for c, x in sorted(v.items()):
            print(f"   {c:32s} {x / n:16.0f} per dispatch")
