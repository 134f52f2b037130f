#!/usr/bin/env python3
"""Steady-state duration of a kernel in a rocprofv3 --kernel-trace CSV: the launches behind the GPU's clock ramp.

    python tools/trace_tail.py <kernel_trace.csv> <kernel name substring> [last N = 100]

Prints calls, the average / minimum of all launches (what --stats reports) and of the last N."""
import csv
import sys

path, key = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rows = [r for r in csv.DictReader(open(path)) if key in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
if not d:
    sys.exit("no launch of %r in %s" % (key, path))
t = d[-last:]
print("%s: %d launches, all: average %.1f us, minimum %.1f us; last %d: average %.1f us, minimum %.1f us, maximum %.1f us"
      % (key, len(d), sum(d) / len(d), min(d), len(t), sum(t) / len(t), min(t), max(t)))
blocks = [d[i:i + 50] for i in range(0, len(d), 50)]
print("average per 50 launches (us):", " ".join("%.1f" % (sum(b) / len(b)) for b in blocks))
