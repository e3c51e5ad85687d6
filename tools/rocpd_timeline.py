#!/usr/bin/env python3
"""Time line of a rocprofv3 run stored as a rocpd database: every kernel dispatch longer than a threshold, in start order, with its
queue, start, duration and the idle gap of the whole device in front of it.
usage: python tools/rocpd_timeline.py <dir or .db> [min us = 200] [from ms] [to ms]"""
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
t_from = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
t_to = float(sys.argv[4]) if len(sys.argv) > 4 else 1e18
if os.path.isdir(path):
    hits = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    assert hits, "no .db under " + path
    path = max(hits, key=os.path.getsize)
c = sqlite3.connect(path)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
if not rows:
    sys.exit("no kernels")
t0 = rows[0][1]
busy_until = t0
print("%10s %10s %8s %6s  %s" % ("start ms", "dur ms", "gap ms", "queue", "kernel"))
for name, st, en, q in rows:
    gap = max(0, st - busy_until)
    busy_until = max(busy_until, en)
    ms = (st - t0) / 1e6
    if ms < t_from or ms > t_to or (en - st) / 1e3 < min_us:
        continue
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    print("%10.3f %10.3f %8.3f %6s  %s" % (ms, (en - st) / 1e6, gap / 1e6, q, name))

# device-busy fraction per run: kernels clustered by idle gaps of more than 100 ms (a tool's runs are that far apart), intervals merged
clusters, cur = [], None
for name, st, en, q in rows:
    if cur is None or st - cur["end"] > 100e6:
        cur = {"start": st, "end": en, "busy": 0, "last": st, "n": 0}
        clusters.append(cur)
    lo = max(st, cur["last"])
    if en > lo:
        cur["busy"] += en - lo
        cur["last"] = en
    cur["end"] = max(cur["end"], en)
    cur["n"] += 1
print()
print("device busy per run (kernels at most 100 ms apart; copies by DMA engines are not kernels):")
for c_ in clusters:
    span = c_["end"] - c_["start"]
    if c_["n"] >= 50 and span > 0:
        print("  from %9.3f ms: %6d kernels over %8.3f ms, busy %8.3f ms = %.3f" % ((c_["start"] - t0) / 1e6, c_["n"], span / 1e6, c_["busy"] / 1e6, c_["busy"] / span))
