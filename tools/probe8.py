#!/usr/bin/env python3
import ctypes as C, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libhbm_probe8.so"))
lib.probe8.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
n_units = 50_003_968  # ~100 M pairs, a multiple of 512*16
big = torch.randint(0, 255, (n_units * 64,), dtype=torch.uint8, device="cuda")
parts = [big[k * n_units * 16:(k + 1) * n_units * 16] for k in range(4)]
out = torch.zeros(n_units, dtype=torch.int32, device="cuda")
ref = None
st = torch.cuda.Stream(); cus = torch.cuda.get_device_properties(0).multi_processor_count
torch.cuda.synchronize()
NAMES = {0: "tile, dword sc1 (kernel's form)", 1: "wave runs, dword sc1", 2: "wave runs, LDS -> dwordx4 sc1", 3: "tile, no store",
         4: "wave runs, LDS -> dwordx4 plain"}
cases = [(0, 512, 4), (3, 512, 4), (1, 512, 4), (2, 512, 4), (4, 512, 4), (1, 512, 8), (2, 512, 8), (2, 512, 16), (0, 256, 4), (2, 256, 4),
         (2, 256, 8)]
res = {}
with torch.cuda.stream(st):
    for rnd in range(5):
        for mode, block, run in cases:
            work = n_units // block if mode in (0, 3) else n_units // (block * run)
            for div in (8, 2, 0):
                if mode in (0, 3):
                    grid = min(cus * 64, max(cus * 2, work // div)) if div else cus * 4
                else:  # same number of workgroups as the tile form would get
                    grid = min(cus * 64, max(cus * 2, (work * run) // div)) if div else cus * 4
                    grid = min(grid, work)
                for i in range(4):
                    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(st)
                    r = lib.probe8(mode, block, run, grid, parts[0].data_ptr(), parts[1].data_ptr(), parts[2].data_ptr(),
                                   parts[3].data_ptr(), out.data_ptr(), n_units, st.cuda_stream)
                    assert r == 0, (mode, block, run)
                    e.record(st); e.synchronize()
                    if i: res.setdefault((mode, block, run, div), []).append(a.elapsed_time(e))
                if mode != 3 and rnd == 0:
                    if ref is None: ref = out.clone()
                    assert torch.equal(out, ref), (mode, block, run, div)
                    out.zero_()
for (mode, block, run, div), v in sorted(res.items(), key=lambda kv: np.median(kv[1])):
    t = float(np.median(v))
    print("%-34s block=%d run=%-2d grid=%-10s %.4f ms  %.0f GB/s" % (NAMES[mode], block, run, ("tiles/%d" % div) if div else "persist x4", t,
                                                                   (3.2e9 + (0 if mode == 3 else 0.2e9)) * n_units / 5e7 / t / 1e6))
