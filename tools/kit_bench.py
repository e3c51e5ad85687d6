#!/usr/bin/env python3
"""A kit layout on its static fast kernel against the generic kernel (forced): launch time by events on the launch's stream,
algorithmic GB/s and the fraction of the 8 TB/s HBM roofline, codes verified against the generator's construction truth and the
molecular bytes of the two kernels against each other.   usage: python tools/kit_bench.py [kit8u9x2,kit8u12x2,...] [pairs]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine  # noqa: E402

cfgs = (sys.argv[1] if len(sys.argv) > 1 else "kit8u9x2,kit8u12x2").split(",")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60_000_000
for cfg in cfgs:
    w = synth.generate(cfg, n, device="cuda")
    M = w.layout.mol_width
    st = torch.cuda.Stream()
    mols = {}
    with Engine(0) as e:
        e.set_plan(w.plan)
        e.set_barcodes(w.barcode_strings())
        torch.cuda.synchronize()
        for mode in ("auto", "force_generic"):
            e.set_option("force_generic", 1 if mode == "force_generic" else 0)
            kind = e.kernel_kind(False)
            codes = torch.full((n,), 0x7777, dtype=torch.int16, device="cuda")
            mol = torch.full((n, max(M, 1)), 0x55, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            ts = []
            for i in range(8):
                a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                e.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual], codes.data_ptr(), mol.data_ptr() if M else None, stream=st.cuda_stream)
                z.record(st)
                z.synchronize()
                if i >= 3:
                    ts.append(a.elapsed_time(z))
            ok = torch.equal(codes.view(torch.int16).to(torch.int32) & 0xFFFF, w.expected)
            mols[mode] = mol
            t = float(np.median(ts))
            gbs = n * synth.ALGO_BYTES[cfg] / t / 1e6
            print("%-10s %-13s kernel %-8s %8.3f ms  %6.2f G pairs/s  %6.0f GB/s algorithmic = %.3f of 8 TB/s  codes ok=%s" % (cfg, mode, kind, t, n / t / 1e6, gbs, gbs / 8000.0, ok), flush=True)
    print("%-10s molecular bytes of the two kernels equal: %s" % (cfg, torch.equal(mols["auto"], mols["force_generic"])), flush=True)
    del w, mols
    torch.cuda.empty_cache()
