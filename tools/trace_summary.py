#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace csv of a whole job: per-kernel launches / total / average time, and how much of the
span between the first and the last kernel SOME kernel was running (the GPU-busy fraction), with the longest idle gaps.
usage: python tools/trace_summary.py <dir or *_kernel_trace.csv> [--skip-before NAME]  (--skip-before: the span starts at the first launch of that kernel)"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    if os.path.isdir(path):
        hits = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
        assert hits, "no *kernel_trace.csv under " + path
        path = max(hits, key=os.path.getsize)
    start_at = sys.argv[sys.argv.index("--skip-before") + 1] if "--skip-before" in sys.argv else None
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0][:70]))
    rows.sort()
    if start_at:
        first = next((i for i, r in enumerate(rows) if start_at in r[2]), 0)
        rows = rows[first:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    per = {}
    for s, e, k in rows:
        c = per.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += e - s
    busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
    last_name = rows[0][2]
    for s, e, k in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name, k, (cur_e - rows[0][0]) / 1e6))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= cur_e:
            last_name = k
    busy += cur_e - cur_s
    span = t1 - t0
    print("kernel trace: %s" % os.path.basename(path))
    print("span first..last kernel %.1f ms, some kernel running %.1f ms = %.3f of the span; %d launches" % (span / 1e6, busy / 1e6, busy / span, len(rows)))
    print("%-72s %8s %11s %10s %7s" % ("kernel", "launches", "total ms", "avg us", "% busy"))
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:24]:
        print("%-72s %8d %11.2f %10.1f %7.1f" % (k, n, t / 1e6, t / n / 1e3, 100.0 * t / busy))
    gaps.sort(reverse=True)
    print("idle: %.1f ms in %d gaps; the longest:" % (sum(g[0] for g in gaps) / 1e6, len(gaps)))
    for g, a, b, at in gaps[:16]:
        print("  %8.2f ms at %8.1f ms  after %-36s before %s" % (g / 1e6, at, a[:36], b[:36]))


main()
