#!/usr/bin/env python3
"""Runs one launch form of a config in a tight loop for a few seconds while sampling rocm-smi (clocks, power):
does a persistent grid run at other clocks than an oversubscribed one?  usage: python tools/smi_watch.py cfg wq seconds"""
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine  # noqa: E402

cfg, wq, secs = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
n = {"cfg2": 10_000_000, "cfg3": 100_000_000, "cfg4": 62_500_000, "cfg5": 125_000_000}[cfg]
e = Engine(0)
lay = e.set_plan(synth.config_plan(cfg))
w = synth.generate(cfg, n, device="cuda", layout=lay)
e.set_barcodes(w.barcode_strings())
e.set_option("work_queue", wq)
M = lay.mol_width
codes = torch.empty(n, dtype=torch.int16, device="cuda")
mol = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
samples, stop = [], False


def watch():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out.strip().split("\n")[-1])
        except Exception as ex:
            samples.append("rocm-smi failed: %r" % ex)
            break
        time.sleep(0.2)


th = threading.Thread(target=watch)
th.start()
t0, k = time.time(), 0
rates = []  # (seconds since start, ms per launch of this group of 50)
while time.time() - t0 < secs:
    g0 = time.time()
    for _ in range(50):
        e.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual], codes.data_ptr(), mol.data_ptr() if M else None)
    e.synchronize()
    k += 50
    rates.append((time.time() - t0, (time.time() - g0) / 50 * 1e3))
dt = time.time() - t0
stop = True
th.join()
print("%s work_queue=%d: %d launches in %.2f s = %.4f ms per launch" % (cfg, wq, k, dt, dt / k * 1e3))
step = max(1, len(rates) // 24)
print("ms per launch over time (s: ms):", "  ".join("%.1f: %.4f" % r for r in rates[::step]))
hdr = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True).stdout.strip().split("\n")
print(hdr[0] if hdr else "")
for sline in samples[::max(1, len(samples) // 16)]:
    print(sline)
