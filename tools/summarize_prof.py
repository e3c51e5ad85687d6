#!/usr/bin/env python3
"""Condenses gpurun_out/prof (rocprofv3 runs of bench.py) into the files committed under profiles/.
usage: python tools/summarize_prof.py <round-tag> [config] [pairs]"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

# 1. kernel-trace --stats: keep the rows of this repo's kernels (+ the header)
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
assert stats, "no kernel_stats.csv"
with open(stats[0]) as fh:
    rows = list(csv.reader(fh))
ours = [r for r in rows[1:] if "demux_" in r[0] or "reduce_partials" in r[0]]
with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, cfg)), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(rows[0])
    w.writerows(ours)
    w.writerow(["# command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 100 --warmup 30 "
                "--no-cpu-baseline (config %s, %d pairs); other rows (torch data generation) omitted" % (cfg, pairs)])
avg_ns = None
for r in ours:
    if "demux_fast" in r[0] or "demux_generic" in r[0]:
        avg_ns = float(r[3])
        print("kernel", r[0][:60], "calls", r[1], "avg ns", r[3])

# 2. PMC passes
summ = collections.OrderedDict()
for name in ["pmc_fetch", "pmc_write", "pmc_sq"]:
    fs = glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(fs[0])):
        if "demux_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count")}
    for k, v in agg.items():
        summ[k] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "dispatches": len(v)}
    summ["_dispatch_" + name] = meta
out = {"config": cfg, "n_pairs": pairs, "counters": summ}
if "FETCH_SIZE" in summ and "WRITE_SIZE" in summ:
    # MI355X_MICROARCH.md / HBM: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
    # exactly half of a wide coalesced streaming read -> doubled.  WRITE_SIZE is exact.
    fetch = summ["FETCH_SIZE"]["mean"] * 1024 * 2
    write = summ["WRITE_SIZE"]["mean"] * 1024
    out["hbm_bytes_per_launch"] = fetch + write
    out["fetch_bytes_corrected"] = fetch
    out["write_bytes"] = write
    out["correction"] = "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of 16 B/lane streams), WRITE_SIZE KiB x 1024"
    print("traffic per launch: %.3f GB read + %.3f GB write" % (fetch / 1e9, write / 1e9))
if avg_ns:
    out["kernel_avg_ns_rocprof"] = avg_ns
with open(os.path.join(dst, "%s_%s_pmc_summary.json" % (tag, cfg)), "w") as fh:
    json.dump(out, fh, indent=1)
with open(os.path.join(dst, "traffic_%s.json" % cfg), "w") as fh:
    json.dump({"n_pairs": pairs, "hbm_bytes_per_launch": out.get("hbm_bytes_per_launch"), "source": "%s_%s_pmc_summary.json" % (tag, cfg)}, fh)
bj = os.path.join(root, "gpurun_out", "bench.json")
if os.path.exists(bj):
    with open(bj) as fh, open(os.path.join(dst, "%s_%s_bench.json" % (tag, cfg)), "w") as fo:
        fo.write(fh.read())
print("wrote profiles/%s_%s_*" % (tag, cfg))
