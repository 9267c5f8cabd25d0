#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (run on the GPU box; the db is too big to ship).
    python tools/pmc_summary.py <results.db> <COUNTER> [name-filter ...]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
counter = sys.argv[2]
filters = sys.argv[3:]
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
namecol = "kernel_name" if "kernel_name" in cols else "name"
ccol = "counter_name" if "counter_name" in cols else "counter"
vcol = "value" if "value" in cols else "counter_value"
extra = ", grid_size" if "grid_size" in cols else ""
rows = db.execute(f"select {namecol}, {ccol}, {vcol}, dispatch_id from counters_collection").fetchall()
per_dispatch = collections.defaultdict(float)
name_of = {}
for n, c, v, d in rows:
    if c != counter:
        continue
    per_dispatch[d] += float(v)          # summed over XCC/SE instances
    name_of[d] = n
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for d, v in per_dispatch.items():
    a = agg[name_of[d]]
    a[0] += 1; a[1] += v; a[2] = max(a[2], v)
print(f"# counter {counter}: columns = calls, mean per launch, max per launch, total   (columns of table: {cols})")
for n, (k, s, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if filters and not any(f in n for f in filters):
        continue
    print(f"{k:6d} {s/k:14.1f} {mx:14.1f} {s:16.1f}  {n[:120]}")
