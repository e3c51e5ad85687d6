#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: packed rows in pinned host memory -> H2D || kernel || D2H
through the library's slots (no torch on the path).  Prints one JSON line (NOT bench.py's metric:
bench.py's `value` is measured with rows resident in HBM)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
nslots = int(sys.argv[3]) if len(sys.argv) > 3 else 3
batches = int(sys.argv[4]) if len(sys.argv) > 4 else 24

w = synth.generate(cfg, B, device="cpu")
with Engine(0) as eng:
    lay = eng.set_plan(w.plan)
    eng.set_barcodes(w.barcode_strings())
    eng.slots_create(nslots, B)
    for s in range(nslots):
        v = eng.slot(s)
        for k in range(lay.n_streams):
            v["seq"][k][:] = w.seq[k].numpy()
            v["qual"][k][:] = w.qual[k].numpy()
    for s in range(nslots):  # warm-up
        eng.submit(s, B)
    for s in range(nslots):
        eng.wait(s)
    eng.reset_counts()
    t0 = time.perf_counter()
    for b in range(batches):
        s = b % nslots
        if b >= nslots:
            eng.wait(s)
        eng.submit(s, B)
    for s in range(nslots):
        eng.wait(s)
    dt = time.perf_counter() - t0
    ok = bool((eng.slot(0)["codes"] == w.expected.numpy().astype(np.uint16)).all())
    counts = eng.counts()
h2d = sum(lay.seq_stride[k] + lay.qual_stride[k] for k in range(lay.n_streams))
d2h = 2 + lay.mol_width
print(json.dumps({"mode": "streamed (pinned host rows, PCIe inclusive)", "config": cfg, "batch_pairs": B, "slots": nslots,
                  "batches": batches, "pairs_per_s": B * batches / dt, "h2d_GBps": B * batches * h2d / dt / 1e9,
                  "d2h_GBps": B * batches * d2h / dt / 1e9, "codes_ok": ok, "total": int(counts[0])}))
