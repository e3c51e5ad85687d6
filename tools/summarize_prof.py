#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<cfg> (tools/gpu_prof_cfg.sh) into the files committed under profiles/:
  <tag>_<cfg>_kernel_stats.csv   rocprofv3 --stats rows of this repo's kernels + a steady-state summary
  <tag>_<cfg>_pmc_summary.json   FETCH_SIZE / WRITE_SIZE -> HBM bytes per launch, kernel times, roofline
  <tag>_<cfg>_bench.json         the bench.py line of the same box
  traffic_<cfg>.json             what bench.py replays as roofline.traffic
usage: python tools/summarize_prof.py <round-tag> <config>"""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + cfg)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
ALGO = {"cfg2": 18, "cfg3": 34, "cfg4": 58, "cfg5": 34}
WARM = 50  # bench.py's untimed launches: the clock ramp lives there

bench = None
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    with open(bj) as fh:
        txt = fh.read()
    bench = json.loads(txt)
    with open(os.path.join(dst, "%s_%s_bench.json" % (tag, cfg)), "w") as fo:
        fo.write(txt)
pairs = bench["config"]["pairs_per_gpu"] if bench else None

# 1. per-dispatch durations of the demux kernel: all launches and the steady state (first WARM dropped)
out = {"config": cfg, "n_pairs": pairs, "algorithmic_bytes_per_pair": ALGO[cfg]}
newest = lambda pat: sorted(glob.glob(os.path.join(src, pat)), key=os.path.getmtime, reverse=True)  # noqa: E731 (merged runs pile up)
trace = newest("*_kernel_trace.csv")
if trace:
    durs, name = [], None
    for r in csv.DictReader(open(trace[0])):
        k = r.get("Kernel_Name", "")
        if "demux_" in k and "fixup" not in k:
            durs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
            name = k
    durs = [d for _, d in sorted(durs)]
    if durs:
        steady = durs[WARM:] if len(durs) > WARM + 10 else durs
        out["kernel"] = name
        out["rocprof_launches"] = len(durs)
        out["kernel_avg_ns_all_launches"] = sum(durs) / len(durs)
        out["kernel_avg_ns_steady"] = sum(steady) / len(steady)
        out["kernel_min_ns"], out["kernel_max_ns"] = min(durs), max(durs)
        out["steady_definition"] = "launches %d.. of the rocprofv3 --kernel-trace run (the first %d are bench.py's untimed warm-up: clock ramp)" % (WARM, WARM)
        print("kernel %s: %d launches, avg all %.1f us, steady %.1f us" % (name[:50], len(durs), out["kernel_avg_ns_all_launches"] / 1e3, out["kernel_avg_ns_steady"] / 1e3))
stats = newest("*_kernel_stats.csv")
if stats:
    rows = list(csv.reader(open(stats[0])))
    ours = [r for r in rows[1:] if "demux_" in r[0] or "reduce_partials" in r[0]]
    with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, cfg)), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(rows[0])
        w.writerows(ours)
        w.writerow(["# command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --config %s --steps 100 "
                    "--warmup 50 --no-cpu-baseline --no-extras (%s pairs); other rows (torch data generation) omitted" % (cfg, pairs)])
        if "kernel_avg_ns_steady" in out:
            w.writerow(["# steady state (launches %d.. of the same trace): average %.0f ns; all %d launches: %.0f ns" % (
                WARM, out["kernel_avg_ns_steady"], out["rocprof_launches"], out["kernel_avg_ns_all_launches"])])

# 2. HBM traffic: separate --pmc passes; FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of
#    a wide coalesced streaming read (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE is exact
ctr = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = os.path.join(src, "pmc_%s.csv" % c)
    if os.path.exists(f):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "demux_" in r["Kernel_Name"] and "fixup" not in r["Kernel_Name"] and r["Counter_Name"] == c]
        if v:
            ctr[c] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "dispatches": len(v)}
out["counters"] = ctr
if "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
    fetch = ctr["FETCH_SIZE"]["mean"] * 1024 * 2
    write = ctr["WRITE_SIZE"]["mean"] * 1024
    out["fetch_bytes_corrected"], out["write_bytes"] = fetch, write
    out["hbm_bytes_per_launch"] = fetch + write
    out["correction"] = "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of 16 B/lane streams), WRITE_SIZE KiB x 1024"
    if pairs:
        out["traffic_over_algorithmic"] = (fetch + write) / (pairs * ALGO[cfg])
    print("traffic per launch: %.3f GB read + %.3f GB write = %.4f x algorithmic" % (fetch / 1e9, write / 1e9, out.get("traffic_over_algorithmic", 0)))
if pairs and "kernel_avg_ns_steady" in out:
    b = pairs * ALGO[cfg]
    out["roofline"] = {"algorithmic_bytes_per_launch": b, "peak_GBps": 8000.0,
                       "achieved_GBps_steady": b / out["kernel_avg_ns_steady"], "frac_steady": b / out["kernel_avg_ns_steady"] / 8000.0,
                       "achieved_GBps_all_launches": b / out["kernel_avg_ns_all_launches"],
                       "frac_all_launches": b / out["kernel_avg_ns_all_launches"] / 8000.0}
    if bench:
        out["roofline"]["bench_kernel_ms_hip_events"] = bench["roofline"]["kernel_ms"]
        out["roofline"]["bench_frac"] = bench["roofline"]["frac"]
        out["roofline"]["bench_ms_per_step"] = bench["ms_per_step"]
with open(os.path.join(dst, "%s_%s_pmc_summary.json" % (tag, cfg)), "w") as fh:
    json.dump(out, fh, indent=1)
if out.get("hbm_bytes_per_launch"):
    with open(os.path.join(dst, "traffic_%s.json" % cfg), "w") as fh:
        import hashlib
        with open(os.path.join(root, "quade_amd", "csrc", "quade_kernels.hip"), "rb") as kf:
            ksha = hashlib.sha256(kf.read()).hexdigest()[:16]
        # "commit" is stamped afterwards in the repository (tools/stamp_traffic.py): the GPU box has no .git
        json.dump({"n_pairs": pairs, "hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "source": "%s_%s_pmc_summary.json" % (tag, cfg),
                   "kernel_source_sha16": ksha, "commit": None}, fh)
print("wrote profiles/%s_%s_*" % (tag, cfg))
