#!/usr/bin/env python3
"""Developer tool: deciles of the per-dispatch durations in a rocprofv3 --kernel-trace CSV, per kernel (the --stats
average hides bimodal kernels); optional second argument = a kernel-name substring whose durations are listed in order."""
import csv
import sys
from collections import defaultdict
rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in sorted(rows.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    if len(v) < 20:
        continue
    ds = sorted(d for _, d in v)
    n = len(ds)
    print("%-60s n=%5d  " % (k[:60], n) + " ".join("%6.1f" % (ds[min(n - 1, n * q // 10)] / 1e3) for q in range(11)))
if len(sys.argv) > 2:
    for k, v in rows.items():
        if sys.argv[2] in k:
            v.sort()
            skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
            print(k[:60], "in order from dispatch", skip)
            print(" ".join("%.0f" % (d / 1e3) for _, d in v[skip:skip + 150]))
