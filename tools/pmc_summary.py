#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel, mean counter value per dispatch."""
import csv
import collections
import glob
import sys

files = []
for d in sys.argv[1:]:
    files += glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("void ", "").replace("nrc::(anonymous namespace)::", "").split("(")[0][:60]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    if not k.startswith("k_"):
        continue
    print("%s  (dispatches %d)" % (k, n))
    for c, v in sorted(cs.items()):
        print("    %-32s %.6g" % (c, sum(v) / len(v)))
