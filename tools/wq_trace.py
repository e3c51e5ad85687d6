#!/usr/bin/env python3
"""Alternates the launch forms of the fast kernel in one process (for rocprofv3 --kernel-trace: the queued form is its own
instantiation, so the per-dispatch durations separate by kernel name).  usage: python3 tools/wq_trace.py cfg [launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = {"cfg2": 10_000_000, "cfg3": 100_000_000, "cfg4": 62_500_000, "cfg5": 125_000_000}[cfg]
e = Engine(0)
lay = e.set_plan(synth.config_plan(cfg))
w = synth.generate(cfg, n, device="cuda", layout=lay)
e.set_barcodes(w.barcode_strings())
M = lay.mol_width
codes = torch.empty(n, dtype=torch.int16, device="cuda")
mol = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for rnd in range(reps):
    for wq in (1, 2):
        e.set_option("work_queue", wq)
        for _ in range(3):
            e.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual], codes.data_ptr(), mol.data_ptr() if M else None)
        e.synchronize()
assert torch.equal(codes.view(torch.int16).to(torch.int32) & 0xFFFF, w.expected)
print("done", cfg)
