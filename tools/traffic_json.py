#!/usr/bin/env python3
"""pmc_summary.txt (tools/prof_traffic.sh) -> profiles/pmc_traffic.json"""
import json, re, sys
src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
out = {"source": tag, "unit": "KB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, mean of the timed dispatches)", "kernels": {}}
cur = None
for line in open(src):
    m = re.match(r"== (.*)", line)
    if m:
        cur = m.group(1).strip()
        out["kernels"][cur] = {}
        continue
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([\d.]+)", line)
    if m and cur:
        out["kernels"][cur][m.group(1) + "_KB"] = float(m.group(2))
out["kernels"] = {k: v for k, v in out["kernels"].items() if "FETCH_SIZE_KB" in v and "WRITE_SIZE_KB" in v}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
