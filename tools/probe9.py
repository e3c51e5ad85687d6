#!/usr/bin/env python3
"""Runs tools/hbm_probe9.hip: the cfg4 kernel's bytes with two output streams (codes, molecular bytes: as shipped) against ONE
stream of interleaved 14-byte records, arms interleaved in one process.  usage: python tools/probe9.py"""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libhbm_probe9.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(here, "hbm_probe9.hip")])
lib = C.CDLL(so)
lib.probe9.argtypes = [C.c_int] * 3 + [C.c_void_p] * 7 + [C.c_int64, C.c_void_p]
n_units = 31_250_000 // 2048 * 2048  # cfg4's 62.5 M pairs = 31.25 M lanes' worth, a multiple of 512 * 4
s1, s2 = (torch.randint(0, 255, (n_units * 28 + 64,), dtype=torch.uint8, device="cuda") for _ in range(2))
q1, q2 = (torch.randint(0, 255, (n_units * 16 + 64,), dtype=torch.uint8, device="cuda") for _ in range(2))
codes = torch.zeros(n_units * 4 + 64, dtype=torch.uint8, device="cuda")
mol = torch.zeros(n_units * 24 + 64, dtype=torch.uint8, device="cuda")
rec = torch.zeros(n_units * 28 + 64, dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
cus = torch.cuda.get_device_properties(0).multi_processor_count
torch.cuda.synchronize()
NAMES = {0: "two streams: codes + molecular bytes (as shipped)", 1: "one stream of 14-byte records", 2: "no output"}
res = {}
with torch.cuda.stream(st):
    for rnd in range(6):
        for block in (512, 256):
            nsuper = n_units // (block * 4)
            for per in (2, 4, 8):  # super-tiles per workgroup
                grid = max(cus * 2, nsuper // per)
                for mode in (0, 1, 2):
                    for i in range(4):
                        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(st)
                        r = lib.probe9(mode, block, grid, s1.data_ptr(), q1.data_ptr(), s2.data_ptr(), q2.data_ptr(), codes.data_ptr(),
                                       mol.data_ptr(), rec.data_ptr(), n_units, st.cuda_stream)
                        assert r == 0, (mode, block)
                        e.record(st)
                        e.synchronize()
                        if i:
                            res.setdefault((mode, block, per), []).append(a.elapsed_time(e))
print("cfg4 bytes, %d pairs: 44 B in + 14 B out per pair" % (2 * n_units))
for (mode, block, per), v in sorted(res.items(), key=lambda kv: np.median(kv[1])):
    t = float(np.median(v))
    print("%-52s block=%d super-tiles/wg=%d  min %.4f  med %.4f ms  %.0f GB/s" % (NAMES[mode], block, per, min(v), t,
                                                                                 2 * n_units * (44 + (0 if mode == 2 else 14)) / t / 1e6))
