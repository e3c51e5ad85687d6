#!/usr/bin/env python3
"""Interleaved A/B timing (one process, one GPU) of the fast kernel against launch shapes of the
wave-span kernel.  usage: python tools/tune_wave.py cfg [pairs]
env: TUNE_WBLOCK=256,512  TUNE_WQUADS=1,2,4,8,16  TUNE_LIBS=extra .so paths  TUNE_ROUNDS=3"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine, LIB_PATH  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
default_n = {"cfg2": 10_000_000, "cfg3": 100_000_000, "cfg4": 62_500_000, "cfg5": 125_000_000}
n = int(sys.argv[2]) if len(sys.argv) > 2 else default_n[cfg]
wblocks = [int(x) for x in os.environ.get("TUNE_WBLOCK", "256,512").split(",")]
wquads = [int(x) for x in os.environ.get("TUNE_WQUADS", "1,2,4,8,16").split(",")]
libs = [LIB_PATH] + [p for p in os.environ.get("TUNE_LIBS", "").split(",") if p]
rounds, reps = int(os.environ.get("TUNE_ROUNDS", "3")), 4

w = synth.generate(cfg, n, device="cuda")
M = w.layout.mol_width
codes = torch.empty(n, dtype=torch.int16, device="cuda")
mol = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
engines = []
for lp in libs:
    e = Engine(0, lib_path=lp)
    e.set_plan(w.plan)
    e.set_barcodes(w.barcode_strings())
    engines.append(e)
variants = [("fast", 1, 0, 0)] + [("wave", 3, b, q) for b in wblocks for q in wquads]
res = {}
seq_p, qual_p = [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual]
with torch.cuda.stream(st):
    for r in range(rounds):
        for lp, e in zip(libs, engines):
            for name, k, b, q in variants:
                e.set_option("kernel", k)
                e.set_option("wave_block", b)
                e.set_option("wave_quads", q)
                if e.kernel_kind() != name:
                    continue
                codes.zero_()
                for i in range(reps + 1):
                    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(st)
                    e.demux_device(n, seq_p, qual_p, codes.data_ptr(), mol.data_ptr() if M else None, stream=st.cuda_stream)
                    z.record(st)
                    z.synchronize()
                    if i:
                        res.setdefault((os.path.basename(lp), name, b, q), []).append(a.elapsed_time(z))
                ok = torch.equal(codes.view(torch.int16).to(torch.int32) & 0xFFFF, w.expected)
                assert ok or os.environ.get("TUNE_NOCHECK"), (lp, name, b, q)
B = synth.ALGO_BYTES[cfg]
print("%s n=%d  algorithmic %d B/pair" % (cfg, n, B))
print("%-22s %-5s %5s %5s %9s %9s %9s %6s" % ("lib", "kern", "block", "quads", "min ms", "med ms", "GB/s(med)", "frac"))
for (lp, name, b, q), v in sorted(res.items(), key=lambda kv: np.median(kv[1])):
    med = float(np.median(v))
    print("%-22s %-5s %5d %5d %9.4f %9.4f %9.0f %6.3f" % (lp, name, b, q, min(v), med, n * B / med / 1e6, n * B / med / 1e6 / 8000))
