#!/usr/bin/env python3
"""Rate of the generic kernel (forced) per config, for DESIGN.md.  env GENERIC_LIBS=path1,path2: extra builds to compare."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth
from quade_amd.hip_backend import Engine, LIB_PATH
LIBS = [LIB_PATH] + [p for p in os.environ.get("GENERIC_LIBS", "").split(",") if p]
for cfg in os.environ.get("GENERIC_CFGS", "cfg2,cfg3,cfg4,cfg5").split(","):
    n = 20_000_000
    w = synth.generate(cfg, n, device="cuda")
    M = w.layout.mol_width
    codes = torch.empty(n, dtype=torch.int16, device="cuda")
    mol = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
    lens = [torch.full((n,), 255, dtype=torch.uint8, device="cuda") for _ in w.seq]
    for lib in LIBS:
      with Engine(0, lib_path=lib) as e:
          e.set_plan(w.plan); e.set_barcodes(w.barcode_strings())
          st = torch.cuda.Stream()
          torch.cuda.synchronize()  # inputs were made on the default stream
          for mode in ("force_generic", "with_len_rows"):
              e.set_option("force_generic", 1 if mode == "force_generic" else 0)
              ts = []
              for i in range(4):
                  a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                  a.record(st)
                  e.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual], codes.data_ptr(),
                                 mol.data_ptr() if M else None,
                                 lens=[t.data_ptr() for t in lens] if mode == "with_len_rows" else (None, None), stream=st.cuda_stream)
                  z.record(st); z.synchronize()
                  if i: ts.append(a.elapsed_time(z))
              ok = torch.equal(codes.view(torch.int16).to(torch.int32) & 0xFFFF, w.expected)
              t = float(np.median(ts))
              print(os.path.basename(lib), "%s %-14s %8.3f ms  %.2f G pairs/s  %.0f GB/s algorithmic  ok=%s" % (cfg, mode, t, n / t / 1e6, n * synth.ALGO_BYTES[cfg] / t / 1e6, ok))
