#!/usr/bin/env python3
"""Prints the top kernels of a rocprofv3 --kernel-trace --stats run (the *_kernel_stats.csv under the given directory)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(f)))[:n]:
    print("%-100s calls %5s avg %10.1f us  total %8.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
