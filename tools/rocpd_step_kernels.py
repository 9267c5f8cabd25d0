#!/usr/bin/env python3
"""Per-step kernel table of the graph replays in a rocprofv3 rocpd database of bench.py: launches / us per step / average us for
EVERY kernel (template arguments kept), so that two builds or switches can be compared line by line.
    python tools/rocpd_step_kernels.py <results.db> <replays> [other.db]     # with a second database: a side-by-side diff"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:96]


def table(path, replays):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    import os
    marker = os.environ.get("S2F_STEP_MARKER", "dcn_bwd")          # a kernel launched exactly 6 times per step (predict: dcn_fwd)
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    starts = [marks[-6 * k] for k in range(replays, 0, -1)]
    sel = rows[starts[0]:starts[-1]]
    n = replays - 1
    agg = collections.defaultdict(lambda: [0, 0])
    for name, s, e in sel:
        a = agg[short(name)]
        a[0] += 1
        a[1] += e - s
    return {k: (v[0] / n, v[1] / n / 1e3) for k, v in agg.items()}


def main():
    replays = int(sys.argv[2])
    a = table(sys.argv[1], replays)
    if len(sys.argv) > 3:
        b = table(sys.argv[3], replays)
        keys = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1]))
        print(f"# total {sum(v[1] for v in a.values())/1e3:.3f} vs {sum(v[1] for v in b.values())/1e3:.3f} ms/step")
        print(f"{'n/step A':>9} {'us/step A':>10} {'n/step B':>9} {'us/step B':>10} {'B - A us':>9}  kernel")
        for k in keys:
            (na, ta), (nb, tb) = a.get(k, (0, 0)), b.get(k, (0, 0))
            print(f"{na:9.1f} {ta:10.1f} {nb:9.1f} {tb:10.1f} {tb - ta:9.1f}  {k}")
        return
    print(f"# total {sum(v[1] for v in a.values())/1e3:.3f} ms/step, {sum(v[0] for v in a.values()):.0f} launches/step")
    print(f"{'n/step':>8} {'us/step':>10} {'avg us':>8}  kernel")
    for k, (n, t) in sorted(a.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:8.1f} {t:10.1f} {t / n:8.2f}  {k}")


if __name__ == "__main__":
    main()
