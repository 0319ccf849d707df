#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean of each counter over dispatches."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k)
            for c, v in sorted(cs.items()):
                print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
