#!/usr/bin/env python3
"""A handful of launches of one config for a rocprofv3 counter pass (bench.py runs >= 50 warm-up
launches, far too many under --pmc).  usage: python3 tools/pmc_run.py cfg [launches] [kernel] [pairs]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine  # noqa: E402

cfg = sys.argv[1]
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kernel = int(sys.argv[3]) if len(sys.argv) > 3 else 0
default_n = {"cfg2": 10_000_000, "cfg3": 100_000_000, "cfg4": 62_500_000, "cfg5": 125_000_000, "wide10": 60_000_000}
n = int(sys.argv[4]) if len(sys.argv) > 4 else default_n[cfg]
w = synth.generate(cfg, n, device="cuda")
M = w.layout.mol_width
codes = torch.empty(n, dtype=torch.int16, device="cuda")
mol = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
with Engine(0) as e:
    e.set_plan(w.plan)
    e.set_barcodes(w.barcode_strings())
    e.set_option("kernel", kernel)
    if os.environ.get("PMC_FORCE_GENERIC"):
        e.set_option("force_generic", 1)
    st = torch.cuda.Stream()
    torch.cuda.synchronize()  # inputs were made on the default stream
    for _ in range(launches):
        e.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual], codes.data_ptr(),
                       mol.data_ptr() if M else None, stream=st.cuda_stream)
    e.synchronize()
    print("pmc_run", cfg, e.kernel_kind(), launches, "launches of", n, "pairs")
