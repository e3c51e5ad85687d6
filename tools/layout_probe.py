#!/usr/bin/env python3
"""Does the RELATIVE placement of the row arrays in HBM decide what the kernel reads?  (Two processes on one box read
cfg3 0.515 or 0.565 ms and stay there: each process draws its allocations anew.)  One buffer, the four row arrays and the
codes carved out of it at base_k = k * (array bytes rounded up to 2 MiB) + k * d for a list of staggers d; every
placement timed several times, interleaved, in one process.  usage: python tools/layout_probe.py [cfg] [pairs]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else {"cfg3": 100_000_000, "cfg5": 125_000_000, "cfg4": 62_500_000}[cfg]
e = Engine(0)
lay = e.set_plan(synth.config_plan(cfg))
w = synth.generate(cfg, n, device="cuda", layout=lay)
e.set_barcodes(w.barcode_strings())
M = lay.mol_width
arrays = [w.seq[0], w.qual[0], w.seq[1], w.qual[1]]
sizes = [a.numel() for a in arrays] + [2 * n] + ([n * M] if M else [])
slot = (max(sizes) + (2 << 20) - 1) // (2 << 20) * (2 << 20)
M1 = 1 << 20
staggers = [int(x) for x in os.environ["LAYOUT_STAGGERS"].split(",")] if os.environ.get("LAYOUT_STAGGERS") else \
    [0, 4096, M1, 2 * M1, 3 * M1, 6 * M1, 10 * M1, 14 * M1, 18 * M1, 30 * M1, 34 * M1, 62 * M1, 66 * M1, 126 * M1, 130 * M1, 254 * M1, 258 * M1]
if os.environ.get("LAYOUT_SPACINGS_MIB"):  # absolute distance between consecutive arrays, MiB
    staggers = [int(float(x) * M1) - slot for x in os.environ["LAYOUT_SPACINGS_MIB"].split(",")]
    assert min(staggers) >= 0
big = torch.empty(len(sizes) * (slot + max(staggers)) + (8 << 20), dtype=torch.uint8, device="cuda")
base = (big.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20)  # 2 MiB aligned start
print("%s n=%d: %d arrays of up to %.0f MB, buffer at 0x%x (2 MiB aligned 0x%x)" % (cfg, n, len(sizes), max(sizes) / 1e6, big.data_ptr(), base))


def place(d):
    ptrs = []
    for k, a in enumerate(arrays):
        off = base - big.data_ptr() + k * slot + k * d
        big[off:off + a.numel()].copy_(a.reshape(-1))
        ptrs.append(big.data_ptr() + off)
    k = len(arrays)
    codes_p = base + k * slot + k * d
    mol_p = base + (k + 1) * slot + (k + 1) * d if M else None
    return ptrs, codes_p, mol_p


st = torch.cuda.Stream()
res = {}
# the generator's own, separately allocated arrays first (what bench.py times)
own = ([t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual])
codes_own = torch.empty(n, dtype=torch.int16, device="cuda")
mol_own = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
print("separately allocated: seq0 0x%x qual0 0x%x seq1 0x%x qual1 0x%x codes 0x%x" % (own[0][0], own[1][0], own[0][1], own[1][1], codes_own.data_ptr()))
for rnd in range(3):
    for d in ["own"] + staggers:
        if d == "own":
            sp, qp, cp, mp = own[0], own[1], codes_own.data_ptr(), mol_own.data_ptr() if M else None
        else:
            ptrs, cp, mp = place(d)
            sp, qp = [ptrs[0], ptrs[2]], [ptrs[1], ptrs[3]]
        torch.cuda.synchronize()
        for i in range(4):
            a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            for _ in range(5):
                e.demux_device(n, sp, qp, cp, mp, stream=st.cuda_stream)
            z.record(st)
            z.synchronize()
            if i:
                res.setdefault(d, []).append(a.elapsed_time(z) / 5)
for d in ["own"] + staggers:
    v = res[d]
    print("stagger %-10s (spacing %8.2f MiB)  min %.4f  median %.4f ms" % (d, (slot + d) / M1 if d != "own" else 0, min(v), float(np.median(v))))
