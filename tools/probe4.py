#!/usr/bin/env python3
import ctypes as C
import os
import numpy as np
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libhbm_probe4.so"))
lib.probe4.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int, C.c_void_p]
nv = 50_000_000
arrs = [torch.randint(0, 255, (nv * 16,), dtype=torch.uint8, device="cuda") for _ in range(4)]
out = torch.empty(nv * 16, dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
cus = torch.cuda.get_device_properties(0).multi_processor_count
cases = [(0, 0), (64, 0), (16, 0), (8, 0), (4, 0), (2, 0), (1, 0), (4, 2), (2, 1), (1, 1), (2, 2), (1, 2)]
res = {}
with torch.cuda.stream(st):
    for rnd in range(3):
        for block, wg in [(512, 2), (512, 4), (256, 0)]:
            ntiles = (nv + block - 1) // block
            grid = min(cus * wg, ntiles) if wg else ntiles
            for stride, wide in cases:
                g = min(grid, (nv * 16 // 4) // block) if stride == 0 else grid
                for i in range(4):
                    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(st)
                    r = lib.probe4(block, g, *[t.data_ptr() for t in arrs], out.data_ptr(), nv, stride, wide, st.cuda_stream)
                    assert r == 0
                    e.record(st); e.synchronize()
                    if i:
                        res.setdefault((block, wg, stride, wide), []).append(a.elapsed_time(e))
print("read 3.2 GB in every case; written bytes vary")
for (block, wg, stride, wide), v in res.items():
    wb = 0 if stride == 0 else nv * [4, 8, 16][wide] / stride
    t = float(np.median(v))
    print("block=%d wg/cu=%-3s store every %2d tiles x %2d B/lane: written %6.1f MB (%4.1f%% of read)  %.4f ms  read-rate %.0f GB/s  total %.0f GB/s" %
          (block, wg if wg else "all", stride, [4, 8, 16][wide], wb / 1e6, wb / 3.2e9 * 100, t, 3.2e9 / t / 1e6, (3.2e9 + wb) / t / 1e6))
