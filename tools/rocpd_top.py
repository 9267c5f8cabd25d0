#!/usr/bin/env python3
"""Top kernels by (name, grid) per step over the graph replays of a bench profile + the idle time between kernels.
    python tools/rocpd_top.py <results.db> <replays> [substring ...]   # with substrings: every (kernel, grid) whose name has one"""
import collections, re, sqlite3, sys
c = sqlite3.connect(sys.argv[1]); replays = int(sys.argv[2])
rows = c.execute("select name,start,end,grid_x,grid_y from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if 'dcn_bwd' in r[0]]
starts = [marks[-6 * k] for k in range(replays, 0, -1)]
sel = rows[starts[0]:starts[-1]]; n = replays - 1
agg = collections.Counter(); cnt = collections.Counter()
busy = 0; gaps = collections.Counter(); prev_end = None
for name, s, e, gx, gy in sel:
    short = re.sub(r'void |at::native::|\(anonymous namespace\)::|<.*', '', name)[:44]
    if name.startswith('Cijk'): short = 'Cijk..' + name[-36:]
    agg[(short, gx, gy)] += e - s; cnt[(short, gx, gy)] += 1
    busy += e - s
    if prev_end is not None and s > prev_end: gaps['idle'] += s - prev_end
    prev_end = max(prev_end or 0, e)
span = sel[-1][2] - sel[0][1]
print(f"# {n} steps: span {span/n/1e6:.2f} ms/step, kernel time {busy/n/1e6:.2f} ms/step, idle between kernels {gaps['idle']/n/1e6:.2f} ms/step, {len(sel)//n} launches/step")
pats = sys.argv[3:]
for k, v in (agg.most_common() if pats else agg.most_common(60)):
    if pats and not any(p in k[0] for p in pats): continue
    print(f"{v/n/1e3:9.1f} us {cnt[k]//n:4d}x avg {v/cnt[k]/1e3:8.2f} us  {k}")
