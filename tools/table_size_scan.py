#!/usr/bin/env python3
"""Kernel rate of the dual 8+8 shape as a function of the number of samples (table size), with the
automatic launch policy and with forced block sizes / grids."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth
from quade_amd.hip_backend import Engine
n = 100_000_000
st = torch.cuda.Stream()
for S in [96, 384, 600, 768, 1024, 1536, 2048]:
    synth.CONFIGS["cfgX"] = dict(dual=True, S=S, read_len=8, mol=False, min_qual=25, pairs=n)
    synth.ALGO_BYTES["cfgX"] = 34
    w = synth.generate("cfg" + "X", n, device="cuda", seed=20260009)
    codes = torch.empty(n, dtype=torch.int16, device="cuda")
    with Engine(0) as e:
        e.set_plan(w.plan); e.set_barcodes(w.barcode_strings())
        out = []
        for block, wg in [(0, 0), (512, 2), (1024, 1), (1024, 2), (512, 16), (512, 48)]:
            e.set_option("fast_block", block); e.set_option("fast_workgroups_per_cu", wg)
            ts = []
            for i in range(6):
                a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                e.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual], codes.data_ptr(), None, stream=st.cuda_stream)
                z.record(st); z.synchronize()
                if i: ts.append(a.elapsed_time(z))
            out.append("%s/%s:%.3f" % (block or "auto", wg or "auto", np.median(ts)))
        ok = torch.equal(codes.view(torch.int16).to(torch.int32) & 0xFFFF, w.expected)
        print("S=%-5d ok=%s  " % (S, ok) + "  ".join(out), flush=True)
    del w, codes
