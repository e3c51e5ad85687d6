#!/usr/bin/env python3
"""Diagnostic: where do the wave kernel's codes differ from the construction truth?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth
from quade_amd.hip_backend import Engine
for cfg in ("cfg3", "cfg4"):
    for n in (1536, 4097):
        w = synth.generate(cfg, n, seed=2000 + n)
        exp = w.expected.numpy().astype(np.uint16)
        for block, quads in ((256, 4), (256, 1), (512, 1), (512, 2)):
            with Engine(0) as e:
                e.set_plan(w.plan); e.set_barcodes(w.barcode_strings())
                e.set_option("kernel", 3); e.set_option("wave_block", block); e.set_option("wave_quads", quads)
                M = e.layout.mol_width
                codes = torch.full((n,), 0x7777, dtype=torch.int16, device="cuda")
                mol = torch.zeros((n, max(M, 1)), dtype=torch.uint8, device="cuda")
                seq = [t.cuda() for t in w.seq]; qual = [t.cuda() for t in w.qual]
                torch.cuda.synchronize()
                e.demux_device(n, [t.data_ptr() for t in seq], [t.data_ptr() for t in qual], codes.data_ptr(), mol.data_ptr() if M else None, stream=0)
                torch.cuda.synchronize()
                got = codes.cpu().numpy().view(np.uint16)
                bad = np.flatnonzero(got != exp)
                print(cfg, n, "block", block, "quads", quads, "kind", e.kernel_kind(), "mismatches", bad.size,
                      "first", bad[:6].tolist(), "quads hit", sorted(set((bad // 512).tolist()))[:12],
                      "got", [hex(x) for x in got[bad[:4]]], "exp", [hex(x) for x in exp[bad[:4]]])
