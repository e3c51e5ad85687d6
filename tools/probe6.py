#!/usr/bin/env python3
import ctypes as C, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libhbm_probe6.so"))
lib.probe6.argtypes = [C.c_int] * 3 + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
nv = 50_000_000
arrs = [torch.randint(0, 255, (nv * 16,), dtype=torch.uint8, device="cuda") for _ in range(4)]
out = torch.empty(nv * 4, dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream(); cus = torch.cuda.get_device_properties(0).multi_processor_count
names = ["plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc0"]
res = {}
with torch.cuda.stream(st):
    for rnd in range(3):
        for block, wg in [(512, 2), (512, 48), (256, 0)]:
            nt = nv // block
            grid = min(cus * wg, nt) if wg else nt
            for mode in range(6):
                for i in range(4):
                    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(st)
                    r = lib.probe6(mode, block, grid, *[t.data_ptr() for t in arrs], out.data_ptr(), nv, st.cuda_stream)
                    assert r == 0
                    e.record(st); e.synchronize()
                    if i: res.setdefault((block, wg, mode), []).append(a.elapsed_time(e))
for (block, wg, mode), v in sorted(res.items()):
    t = float(np.median(v))
    print("block=%d wg/cu=%-3s store %-11s %.4f ms  %.0f GB/s" % (block, wg if wg else "all", names[mode], t, 3.4e9 / t / 1e6))
