#!/usr/bin/env python3
"""Per-workgroup start / end times of one fast launch (a -DQD_DEBUG_TIMES build of the library): how long the
workgroups run, how evenly they end (the tail), which XCD / CU they ran on.
usage: python tools/wg_times.py cfg [lib] ; env WG_BLOCK, WG_PER_CU, WG_QUEUE (launch options)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "quade_amd", "lib", "variants", "libq_dbgtimes.so")
n = {"cfg2": 10_000_000, "cfg3": 100_000_000, "cfg4": 62_500_000, "cfg5": 125_000_000}[cfg]
e = Engine(0, lib_path=lib)
e.lib.qd_debug_times.restype = C.c_int
e.lib.qd_debug_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
lay = e.set_plan(synth.config_plan(cfg))
w = synth.generate(cfg, n, device="cuda", layout=lay)
e.set_barcodes(w.barcode_strings())
if os.environ.get("WG_BLOCK"):
    e.set_option("fast_block", int(os.environ["WG_BLOCK"]))
if os.environ.get("WG_QUEUE"):
    e.set_option("work_queue", int(os.environ["WG_QUEUE"]))
if os.environ.get("WG_PER_CU"):
    e.set_option("fast_workgroups_per_cu", int(os.environ["WG_PER_CU"]))
M = lay.mol_width
codes = torch.empty(n, dtype=torch.int16, device="cuda")
mol = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for rep in range(6):
    e.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual], codes.data_ptr(), mol.data_ptr() if M else None)
    e.synchronize()
buf = np.zeros((65536, 3), np.uint64)
assert e.lib.qd_debug_times(e._h, buf.ctypes.data, 65536) == 0
used = buf[:, 1] > 0
t = buf[used]
g = int(used.sum())
t0, t1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
tick = 0.01  # us per tick of the 100 MHz wall clock
start, end = (t0 - t0.min()) * tick, (t1 - t0.min()) * tick
dur = end - start
span = end.max()
xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
print("%s: %d workgroups, kernel span %.1f us (first start -> last end)" % (cfg, g, span))
print("  workgroup duration us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(dur, [0, 10, 50, 90, 100])))
print("  start time us:         p50 %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(start, [50, 90, 100])))
print("  end time us:           p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % tuple(np.percentile(end, [10, 50, 90, 99, 100])))
# how much of the span runs with few workgroups left
order = np.sort(end)
for frac in (0.5, 0.25, 0.1):
    k = int(g * (1 - frac))
    print("  last %2.0f %% of the workgroups end within the final %.1f us (%.1f %% of the span)" % (frac * 100, span - order[k], (span - order[k]) / span * 100))
# concurrency over time: workgroups running, sampled
ts = np.linspace(0, span, 21)
running = [(int(((start <= x) & (end > x)).sum())) for x in ts]
print("  workgroups running at 0 %, 5 %, ... 100 % of the span:", running)
slots = max(running)
print("  sum of workgroup durations / %d resident slots = %.1f us (the span if every slot were busy throughout: %.1f %% of the span)" % (slots, dur.sum() / slots, dur.sum() / slots / span * 100))
for x in range(8):
    m = xcc == x
    if m.any():
        print("  XCC %d: %5d workgroups, mean duration %.1f us, last end %.1f us" % (x, int(m.sum()), dur[m].mean(), end[m].max()))
