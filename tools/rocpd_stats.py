#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / share, like `--stats` CSV.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--top 40] [--skip-first-frac 0.0]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if len(name) > 110:
        name = name[:107] + "..."
    return name


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end from kernels order by start").fetchall() if "name" in cols else None
    if rows is None:
        print("columns:", cols); return
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1; a[1] += e - s
    total = sum(a[1] for a in agg.values())
    span = rows[-1][2] - rows[0][1]
    print(f"# {len(rows)} dispatches, {len(agg)} kernels, sum of kernel time {total/1e6:.2f} ms, trace span {span/1e6:.2f} ms")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'share%':>7}  kernel")
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{n:7d} {t/1e6:10.3f} {t/n/1e3:9.2f} {100*t/total:7.2f}  {short(name)}")


if __name__ == "__main__":
    main()
