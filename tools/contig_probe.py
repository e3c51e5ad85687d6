#!/usr/bin/env python3
"""Do physically contiguous allocations (hipExtMallocWithFlags, hipDeviceMallocContiguous) read faster than ordinary ones?
cfg3's row arrays and codes: torch's own allocations, plain hipMalloc, contiguous hipExtMallocWithFlags -- same process, interleaved.
usage: python tools/contig_probe.py [cfg]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = {"cfg3": 100_000_000, "cfg5": 125_000_000, "cfg4": 62_500_000}[cfg]
e = Engine(0)
lay = e.set_plan(synth.config_plan(cfg))
w = synth.generate(cfg, n, device="cuda", layout=lay)
e.set_barcodes(w.barcode_strings())
M = lay.mol_width
arrays = [w.seq[0], w.qual[0], w.seq[1], w.qual[1]]
sizes = [a.numel() for a in arrays] + [2 * n] + ([n * M] if M else [])
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


def alloc_set(flag):
    ptrs = []
    for s in sizes:
        p = C.c_void_p()
        r = hip.hipMalloc(C.byref(p), s) if flag is None else hip.hipExtMallocWithFlags(C.byref(p), s, flag)
        assert r == 0, ("alloc failed", flag, r)
        ptrs.append(p.value)
    for k, a in enumerate(arrays):
        assert hip.hipMemcpy(ptrs[k], a.data_ptr(), a.numel(), 3) == 0  # device to device
    return ptrs


codes_own = torch.empty(n, dtype=torch.int16, device="cuda")
mol_own = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
sets = {"torch": [a.data_ptr() for a in arrays] + [codes_own.data_ptr()] + ([mol_own.data_ptr()] if M else [])}
sets["hipMalloc"] = alloc_set(None)
try:
    sets["contiguous"] = alloc_set(4)   # hipDeviceMallocContiguous
except AssertionError as ex:
    print("contiguous allocation refused:", ex)
sets["hipMalloc 2nd set"] = alloc_set(None)
st = torch.cuda.Stream()
res = {}
for rnd in range(4):
    for name, p in sets.items():
        torch.cuda.synchronize()
        for i in range(4):
            a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            for _ in range(5):
                e.demux_device(n, [p[0], p[2]], [p[1], p[3]], p[4], p[5] if M else None, stream=st.cuda_stream)
            z.record(st)
            z.synchronize()
            if i:
                res.setdefault(name, []).append(a.elapsed_time(z) / 5)
for name, v in res.items():
    print("%-18s min %.4f  median %.4f ms   (first array at 0x%x)" % (name, min(v), float(np.median(v)), sets[name][0]))
