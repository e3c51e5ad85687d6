#!/usr/bin/env python3
"""Runs tools/hbm_probe.hip on the GPU box: bandwidth ceiling of the demux access pattern."""
import ctypes as C
import os
import sys

import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libhbm_probe.so"))
lib.probe_run.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
once = len(sys.argv) > 2  # single launch per config (for rocprofv3 --pmc calibration)
n_vec = n_pairs // 2
arrs = [torch.randint(0, 255, (n_vec * 16,), dtype=torch.uint8, device="cuda") for _ in range(4)]
big = torch.randint(0, 255, (n_vec * 16 * 4,), dtype=torch.uint8, device="cuda")
out = torch.empty(n_vec * 4, dtype=torch.int32, device="cuda")
st = torch.cuda.Stream()
cus = torch.cuda.get_device_properties(0).multi_processor_count
rows = []
with torch.cuda.stream(st):
    for ns, u, b in [(4, 1, 256), (4, 2, 256), (4, 4, 256), (4, 1, 512), (4, 2, 512), (4, 4, 512), (4, 1, 1024),
                     (4, 2, 1024), (1, 1, 256), (1, 4, 256), (1, 8, 256), (2, 4, 256)]:
        for wg in ([8] if once else [2, 4, 8, 16, 32]):
            grid = cus * wg
            ts = []
            nv = n_vec if ns == 4 else n_vec * 4 // ns  # same total bytes
            for i in range(1 if once else 6):
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                if ns == 4:
                    r = lib.probe_run(ns, u, b, grid, *[t.data_ptr() for t in arrs], out.data_ptr(), nv, st.cuda_stream)
                else:
                    ptrs = [big.data_ptr() + k * nv * 16 for k in range(ns)] + [0] * (4 - ns)
                    r = lib.probe_run(ns, u, b, grid, *ptrs, out.data_ptr(), nv, st.cuda_stream)
                assert r == 0, r
                e.record(st)
                e.synchronize()
                if i or once:
                    ts.append(a.elapsed_time(e))
            byts = nv * 16 * ns + nv * 4
            rows.append((np.median(ts), ns, u, b, wg, byts))
for t, ns, u, b, wg, byts in sorted(rows):
    print("streams=%d units=%d block=%4d wg/cu=%2d  %.4f ms  %.0f GB/s (read %.2f GB + write %.2f GB)" %
          (ns, u, b, wg, t, byts / t / 1e6, (byts - byts / (16 * ns + 4) * 4) / 1e9, byts / (16 * ns + 4) * 4 / 1e9))
