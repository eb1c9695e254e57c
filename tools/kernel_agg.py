#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 --kernel-trace database (results.db): calls, total, average and longest launch, kernels whose
name matches the regular expression given (default: the builders' kernels).  usage: kernel_agg.py results.db [regex]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"k_(sp|sah|lbvh|treelet|pack|b4|cluster)")
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
agg = collections.OrderedDict()
for name, s, e in c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    k = m.group(1) if m else name[:40]
    if not pat.match(k):
        continue
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[2] = max(a[2], (e - s) / 1e3)
tot = 0.0
for k, (n, t, mx) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("%-28s calls %4d total %8.3f ms avg %8.1f us max %8.1f us" % (k, n, t / 1e3, t / n, mx)); tot += t
print("sum %.3f ms" % (tot / 1e3))
