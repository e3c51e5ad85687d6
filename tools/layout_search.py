#!/usr/bin/env python3
"""Placement lottery: the row arrays and the codes at RANDOM 2 MiB-aligned positions of one large arena, many draws, each
timed in the same process.  How wide is the spread, and does the best draw stay the best when it is measured again?
usage: python tools/layout_search.py [cfg] [draws] [arena GiB]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 24
arena_gib = int(sys.argv[3]) if len(sys.argv) > 3 else 24
n = {"cfg3": 100_000_000, "cfg5": 125_000_000, "cfg4": 62_500_000}[cfg]
e = Engine(0)
lay = e.set_plan(synth.config_plan(cfg))
w = synth.generate(cfg, n, device="cuda", layout=lay)
e.set_barcodes(w.barcode_strings())
M = lay.mol_width
arrays = [w.seq[0], w.qual[0], w.seq[1], w.qual[1]]
sizes = [a.numel() for a in arrays] + [2 * n] + ([n * M] if M else [])
G2 = 2 << 20
units = [(s + G2 - 1) // G2 for s in sizes]  # 2 MiB units per array
arena = torch.empty(arena_gib << 30, dtype=torch.uint8, device="cuda")
base = (arena.data_ptr() + G2 - 1) // G2 * G2
total_units = ((arena_gib << 30) - G2) // G2
rng = np.random.default_rng(7)
st = torch.cuda.Stream()


def draw():
    for _ in range(1000):
        starts = sorted(int(x) for x in rng.integers(0, total_units - max(units), len(units)))
        order = rng.permutation(len(units))
        pos, ok = {}, True
        for s, k in zip(starts, order):
            pos[int(k)] = s
        iv = sorted((pos[k], pos[k] + units[k]) for k in pos)
        for (a0, a1), (b0, b1) in zip(iv, iv[1:]):
            ok = ok and a1 <= b0
        if ok:
            return [pos[k] for k in range(len(units))]
    raise RuntimeError("arena too small")


def measure(p, reps=3):
    off = [base - arena.data_ptr() + u * G2 for u in p]
    for k, a in enumerate(arrays):
        arena[off[k]:off[k] + a.numel()].copy_(a.reshape(-1))
    ptr = [arena.data_ptr() + o for o in off]
    torch.cuda.synchronize()
    ts = []
    for i in range(reps + 1):
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(5):
            e.demux_device(n, [ptr[0], ptr[2]], [ptr[1], ptr[3]], ptr[4], ptr[5] if M else None, stream=st.cuda_stream)
        z.record(st)
        z.synchronize()
        if i:
            ts.append(a.elapsed_time(z) / 5)
    return float(np.median(ts))


res = []
for d in range(draws):
    p = draw()
    res.append((measure(p), p))
res.sort()
print("%s: %d random placements in a %d GiB arena: best %.4f  median %.4f  worst %.4f ms" % (cfg, draws, arena_gib, res[0][0], res[len(res) // 2][0], res[-1][0]))
for t, p in res[:4] + res[-2:]:
    print("  %.4f ms  starts (GiB): %s" % (t, ", ".join("%.3f" % (u * G2 / 2**30) for u in p)))
# are the best and the worst reproducible?
again = [(measure(p, 5), t) for t, p in (res[0], res[1], res[-1])]
print("measured again: best draw %.4f (was %.4f), second %.4f (was %.4f), worst %.4f (was %.4f)" % (again[0][0], again[0][1], again[1][0], again[1][1], again[2][0], again[2][1]))
