#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs: per kernel name, the mean of every counter over its dispatches
(after dropping the counting/warm-up dispatches: the last 3 dispatches per kernel are the timed steps)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            if "mipt::" not in name:
                continue
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mipt::", "")
            acc[short][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k in sorted(acc):
    print("==", k)
    for c in sorted(acc[k]):
        vals = [v for _, v in sorted(acc[k][c])][-3:]
        print("   %-40s %16.1f   (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
