#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 run stored as a rocpd database (the profiler's default output on this image):
launches, total, average, shortest and longest duration per kernel.   usage: python tools/rocpd_stats.py <dir or .db> [> profiles/...]"""
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
if os.path.isdir(path):
    hits = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    assert hits, "no .db under " + path
    path = max(hits, key=os.path.getsize)
c = sqlite3.connect(path)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
print("%-100s %8s %12s %12s %12s %12s" % ("kernel", "launches", "total ms", "avg us", "min us", "max us"))
for name, n, tot, avg, mn, mx in rows:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-100s %8d %12.3f %12.1f %12.1f %12.1f" % (name.split("(")[0][:100], n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3))
