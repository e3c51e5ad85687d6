#!/usr/bin/env python3
"""Runs tools/hbm_probe2.hip: where is the read-bandwidth ceiling of this device?"""
import ctypes as C
import os

import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libhbm_probe2.so"))
lib.probe2.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
GB = 3.2e9
n_vec_total = int(GB // 16)
big = torch.randint(0, 255, (n_vec_total * 16,), dtype=torch.uint8, device="cuda")
out = torch.empty(n_vec_total * 16, dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
cus = torch.cuda.get_device_properties(0).multi_processor_count
rows = []
names = {0: "copy", 1: "read-only", 2: "4 streams + 4B store"}
with torch.cuda.stream(st):
    for mode, nt, u in [(0, 0, 1), (0, 0, 4), (0, 1, 4), (1, 0, 1), (1, 0, 4), (1, 1, 4), (1, 0, 8), (2, 0, 1), (2, 0, 2), (2, 1, 2), (2, 1, 1)]:
        nv = n_vec_total if mode != 2 else n_vec_total // 4
        ntiles = (nv + 256 * u - 1) // (256 * u)
        for wg in [4, 8, 16, 32, 64, 0]:
            grid = cus * wg if wg else ntiles
            grid = min(grid, ntiles)
            if mode == 1:
                grid = min(grid, n_vec_total * 16 // (256 * 8))
            ts = []
            for i in range(6):
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                ptrs = [big.data_ptr() + j * nv * 16 for j in range(4)]
                r = lib.probe2(mode, nt, u, grid, *ptrs, out.data_ptr(), nv, st.cuda_stream)
                assert r == 0, (r, mode, nt, u)
                e.record(st)
                e.synchronize()
                if i:
                    ts.append(a.elapsed_time(e))
            t = float(np.median(ts))
            rd = GB
            wr = {0: GB, 1: grid * 256 * 8, 2: nv * 4}[mode]
            rows.append((mode, nt, u, wg, t, rd, wr))
for mode, nt, u, wg, t, rd, wr in rows:
    print("%-22s nt=%d units=%d wg/cu=%-3s  %.4f ms  read %.0f GB/s  write %.0f GB/s  total %.0f GB/s" %
          (names[mode], nt, u, wg if wg else "all", t, rd / t / 1e6, wr / t / 1e6, (rd + wr) / t / 1e6))
