#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc csv output: mean counter value per dispatch, per kernel.
usage: pmc_summary.py <dir> [substring-filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "evs::"
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "")
                if filt not in k:
                    continue
                k = k.split("(")[0].replace("void ", "")
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("    %-44s mean %18.1f   n=%d" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
