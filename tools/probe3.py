#!/usr/bin/env python3
import ctypes as C
import os
import numpy as np
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libhbm_probe3.so"))
lib.probe3.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
nv = 50_000_000  # 16-byte vectors per stream = 100 M pairs
arrs = [torch.randint(0, 255, (nv * 16,), dtype=torch.uint8, device="cuda") for _ in range(4)]
out = torch.empty(nv * 4 + (1 << 22), dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
cus = torch.cuda.get_device_properties(0).multi_processor_count
names = {0: "no store", 1: "4B store", 2: "4B nt store", 3: "16B store", 4: "16B nt store"}
res = []
with torch.cuda.stream(st):
    for store, u, b in [(0, 1, 256), (0, 2, 256), (0, 4, 256), (1, 1, 256), (1, 2, 256), (1, 4, 256), (2, 1, 256), (2, 2, 256),
                        (2, 4, 256), (3, 4, 256), (4, 4, 256), (0, 2, 512), (1, 2, 512), (2, 2, 512), (3, 4, 512), (4, 4, 512),
                        (0, 1, 512), (1, 1, 512), (2, 1, 512)]:
        ntiles = (nv + b * u - 1) // (b * u)
        for wg in [2, 4, 8, 16, 0]:
            grid = min(cus * wg if wg else ntiles, ntiles)
            if store == 0:
                grid = min(grid, (1 << 20) // b)
            ts = []
            for i in range(7):
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                r = lib.probe3(store, u, b, grid, *[t.data_ptr() for t in arrs], out.data_ptr(), nv, st.cuda_stream)
                assert r == 0, (r, store, u, b)
                e.record(st); e.synchronize()
                if i:
                    ts.append(a.elapsed_time(e))
            t = float(np.median(ts))
            res.append((t, store, u, b, wg))
for t, store, u, b, wg in res:
    wr = 0 if store == 0 else nv * 4
    print("%-13s units=%d block=%d wg/cu=%-3s %.4f ms  algorithmic(3.4GB) %.0f GB/s  actual %.0f GB/s" %
          (names[store], u, b, wg if wg else "all", t, 3.4e9 / t / 1e6, (nv * 64 + wr) / t / 1e6))
