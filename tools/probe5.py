#!/usr/bin/env python3
import ctypes as C, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libhbm_probe5.so"))
lib.probe5.argtypes = [C.c_int] * 3 + [C.c_void_p] * 6 + [C.c_int64, C.c_void_p]
n_units = 31_250_000  # 62.5 M pairs
s1 = torch.randint(0, 255, (n_units * 32 + 64,), dtype=torch.uint8, device="cuda"); s2 = torch.randint(0, 255, (n_units * 32 + 64,), dtype=torch.uint8, device="cuda")
q1 = torch.randint(0, 255, (n_units * 16,), dtype=torch.uint8, device="cuda"); q2 = torch.randint(0, 255, (n_units * 16,), dtype=torch.uint8, device="cuda")
codes = torch.empty(n_units * 4, dtype=torch.uint8, device="cuda"); mol = torch.empty(n_units * 24, dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream(); cus = torch.cuda.get_device_properties(0).multi_processor_count
res = {}
with torch.cuda.stream(st):
    for rnd in range(3):
        for exact in (0, 1):
            for block in (256, 512):
                for wg in (2, 4, 16, 64, 0):
                    nt = n_units // block
                    grid = min(cus * wg, nt) if wg else nt
                    for i in range(4):
                        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(st)
                        r = lib.probe5(exact, block, grid, s1.data_ptr(), q1.data_ptr(), s2.data_ptr(), q2.data_ptr(), codes.data_ptr(), mol.data_ptr(), n_units, st.cuda_stream)
                        assert r == 0
                        e.record(st); e.synchronize()
                        if i: res.setdefault((exact, block, wg), []).append(a.elapsed_time(e))
for (exact, block, wg), v in sorted(res.items(), key=lambda kv: np.median(kv[1])):
    t = float(np.median(v)); moved = n_units * ((28 if exact else 32) * 2 + 32 + 28)
    print("%s rows block=%d wg/cu=%-3s %.4f ms  algorithmic(58 B/pair) %.0f GB/s  moved %.0f GB/s" %
          ("exact 14-B" if exact else "padded 16-B", block, wg if wg else "all", t, n_units * 2 * 58 / t / 1e6, moved / t / 1e6))
