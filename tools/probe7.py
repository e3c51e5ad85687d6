#!/usr/bin/env python3
import ctypes as C, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libhbm_probe7.so"))
lib.probe7.argtypes = [C.c_int] * 3 + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
n_units = 50_000_000  # 100 M pairs
big = torch.randint(0, 255, (n_units * 64,), dtype=torch.uint8, device="cuda")
parts = [big[k * n_units * 16:(k + 1) * n_units * 16] for k in range(4)]
out = torch.empty(n_units, dtype=torch.int32, device="cuda")
st = torch.cuda.Stream(); cus = torch.cuda.get_device_properties(0).multi_processor_count
torch.cuda.synchronize()
res = {}
with torch.cuda.stream(st):
    for rnd in range(5):
        for layout in (0, 1):
            for block in (256, 512):
                nt = n_units // block
                for div in (8, 16, 0):
                    grid = min(cus * 64, max(cus * 2, nt // div)) if div else cus * 4
                    for i in range(4):
                        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(st)
                        r = lib.probe7(layout, block, grid, parts[0].data_ptr(), parts[1].data_ptr(), parts[2].data_ptr(), parts[3].data_ptr(), out.data_ptr(), n_units, st.cuda_stream)
                        assert r == 0
                        e.record(st); e.synchronize()
                        if i: res.setdefault((layout, block, div), []).append(a.elapsed_time(e))
for (layout, block, div), v in sorted(res.items(), key=lambda kv: np.median(kv[1])):
    t = float(np.median(v))
    print("%s block=%d tiles/wg>=%-3s %.4f ms  %.0f GB/s (3.2 GB read + 0.2 GB written)" % ("one interleaved array" if layout else "four arrays         ", block, div if div else "persist x4", t, 3.4e9 / t / 1e6))
