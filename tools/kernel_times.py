#!/usr/bin/env python3
"""Per-launch kernel durations of ONE frame from a rocprofv3 --kernel-trace database (rocpd .db): the launches between the last two
launches of a marker kernel (default k_wh_raygen), in order.  usage: kernel_times.py results.db [marker]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "k_wh_raygen"
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch_')][0]; sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
rows = list(db.execute("select s.kernel_name, d.start, d.end, d.grid_size_x from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, sym)))
idx = [i for i, r in enumerate(rows) if marker in r[0]]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
lo = idx[which]; hi = idx[which + 1] if which + 1 < len(idx) and which != -1 else len(rows)
tot = 0.0
for r in rows[lo:hi]:
    if "rocclr" in r[0]: continue
    us = (r[2] - r[1]) / 1e3; tot += us
    print("%-78s %9.1f us  grid %d" % (r[0].replace("_ZN4mipt", "")[:78], us, r[3]))
print("sum %.1f us, first start to last end %.1f us" % (tot, (rows[hi - 1][2] - rows[lo][1]) / 1e3))
