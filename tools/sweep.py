#!/usr/bin/env python3
"""Interleaved A/B timing of libquade_hip.so build variants on one GPU (run on the GPU box).
usage: python tools/sweep.py [cfg] [pairs] -- prints a table of kernel ms (min / median) and
algorithmic GB/s per (variant, workgroups-per-CU)."""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
wgs = [int(x) for x in os.environ.get("SWEEP_WG", "0").split(",")]
rounds, reps = 3, 5

w = synth.generate(cfg, n, device="cuda")
codes = torch.empty(n, dtype=torch.int16, device="cuda")
mol = torch.empty((n, max(w.layout.mol_width, 1)), dtype=torch.uint8, device="cuda")
libs = sorted(glob.glob(os.path.join(ROOT, "quade_amd/lib/variants/*.so")))
if os.environ.get("SWEEP_ONLY"):
    libs = [l for l in libs if any(t in l for t in os.environ["SWEEP_ONLY"].split(","))]
libs = [os.path.join(ROOT, "quade_amd/lib/libquade_hip.so")] + libs
engines = []
for lp in libs:
    e = Engine(0, lib_path=lp)
    e.set_plan(w.plan)
    e.set_barcodes(w.barcode_strings())
    engines.append(e)
st = torch.cuda.Stream()
res = {}
with torch.cuda.stream(st):
    for r in range(rounds):
        for lp, e in zip(libs, engines):
            for wg in wgs:
                e.set_option("fast_workgroups_per_cu", wg)
                for i in range(reps + 1):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(st)
                    e.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual],
                                   codes.data_ptr(), mol.data_ptr() if w.layout.mol_width else None,
                                   stream=st.cuda_stream)
                    b.record(st)
                    b.synchronize()
                    if i:
                        res.setdefault((os.path.basename(lp), wg), []).append(a.elapsed_time(b))
        ok = torch.equal(codes.view(torch.int16).to(torch.int32) & 0xFFFF, w.expected)
        assert ok or os.environ.get("SWEEP_NOCHECK")
B = synth.ALGO_BYTES[cfg]
print("%-28s %3s %9s %9s %9s" % ("variant", "wg", "min ms", "med ms", "GB/s(med)"))
for (lp, wg), v in sorted(res.items(), key=lambda kv: np.median(kv[1])):
    print("%-28s %3d %9.4f %9.4f %9.0f" % (lp, wg, min(v), np.median(v), n * B / np.median(v) / 1e6))
