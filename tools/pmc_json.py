#!/usr/bin/env python3
"""summary.txt of tools/prof_l1.sh / prof_pmc.sh (tools/pmc_summary.py's output) -> profiles/pmc_kernels.json:
per kernel, the mean of every counter over the timed dispatches, tagged with the commit the profile was taken on.
bench.py reads this file for the counter-based ceilings of its roofline block (HBM, L2, vector L1, VALU issue).
usage: tools/pmc_json.py <summary.txt> <out.json> <tag> [commit]"""
import json
import re
import subprocess
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
commit = sys.argv[4] if len(sys.argv) > 4 else subprocess.run(
    ["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
blob = bench.kernels_blob_sha()      # the id of the kernel sources (kernels.hip + its headers) the profile was taken on: bench.py checks it
out = {"source": tag, "commit": commit, "kernels_hip_blob": blob,
       "note": "rocprofv3 --pmc, separate passes (tools/prof_l1.sh); per-launch means of the timed dispatches; FETCH_SIZE / "
               "WRITE_SIZE in KB as rocprofv3 reports them (FETCH_SIZE counts 64 B per 128-B request on gfx950: double it)",
       "kernels": {}}
cur = None
for line in open(src):
    m = re.match(r"== (.*)", line)
    if m:
        cur = m.group(1).strip()
        if cur:
            out["kernels"][cur] = {}
        continue
    m = re.match(r"\s+(\S+)\s+([\d.]+)\s+\(n=", line)
    if m and cur:
        out["kernels"][cur][m.group(1)] = float(m.group(2))
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("%d kernels -> %s" % (len(out["kernels"]), dst))
