#!/usr/bin/env python3
"""The ATen / runtime 'glue' launches of one replayed step by (kernel, grid): what is left outside the package's kernels.
    python tools/rocpd_glue.py <results.db> <replays>"""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1]); replays = int(sys.argv[2])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gx = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
q = f"select name, start, end{', ' + gx if gx else ''} from kernels order by start"
rows = c.execute(q).fetchall()
marks = [i for i, r in enumerate(rows) if 'dcn_bwd' in r[0]]
# whole steps between consecutive "first dcn_bwd of a step" marks, as rocpd_categories.py: replays - 1 steps
starts = [marks[-6 * k] for k in range(replays, 0, -1)]
sel = rows[starts[0]:starts[-1]]
replays -= 1
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    n = r[0]
    if any(k in n for k in ("at::", "rocclr", "s2f_zero")):
        key = (n[:90], r[3] if gx else 0)
        agg[key][0] += 1; agg[key][1] += (r[2] - r[1]) / 1e3
for (n, g), (k, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{us / replays:9.1f} us/step {k / replays:7.1f}x  grid {g:>9}  {n}")
