#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output per kernel: mean of every counter (development tool)."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")
        if "fir_" not in k and "copy" not in k:
            continue
        short = k.split("(")[0][-70:]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
