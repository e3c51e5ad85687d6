#!/usr/bin/env python3
"""Interleaved timing of launch options (workgroup size x workgroups per CU) of libquade_hip.so on
one GPU.  usage: python tools/tune.py cfg [pairs] ; env TUNE_BLOCKS=0,256,512,1024 TUNE_WG=0,4,16,64
TUNE_LIBS=path1,path2 (optional extra builds)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import synth  # noqa: E402
from quade_amd.hip_backend import Engine, LIB_PATH  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
default_n = {"cfg2": 10_000_000, "cfg3": 100_000_000, "cfg4": 62_500_000, "cfg5": 125_000_000, "wide10": 60_000_000,
             "kit6": 100_000_000, "kit8u8": 60_000_000, "kit12": 60_000_000, "kit10u6": 60_000_000,
             "kit8u9": 60_000_000, "kit8u12": 60_000_000}
n = int(sys.argv[2]) if len(sys.argv) > 2 else default_n[cfg]
blocks = [int(x) for x in os.environ.get("TUNE_BLOCKS", "0,256,512,1024").split(",")]
wgs = [int(x) for x in os.environ.get("TUNE_WG", "0,4,16,64").split(",")]
wqs = [int(x) for x in os.environ.get("TUNE_WQ", "0").split(",")]  # work_queue option: 0 automatic, 1 always, 2 never
# TUNE_OPT="name:v1,v2": one more option dimension (set on builds that know it), e.g. keys_global:0,1
xopt = os.environ.get("TUNE_OPT", "")
xname, xvals = (xopt.split(":")[0], [int(v) for v in xopt.split(":")[1].split(",")]) if xopt else (None, [0])
libs = [LIB_PATH] + [p for p in os.environ.get("TUNE_LIBS", "").split(",") if p]
rounds, reps = int(os.environ.get("TUNE_ROUNDS", "3")), 4
burst = int(os.environ.get("TUNE_BURST", "1"))  # launches per timed sample, back to back (1: every launch timed alone)

engines, loads = [], {}
for lp in libs:
    e = Engine(0, lib_path=lp)
    lay = e.set_plan(synth.config_plan(cfg))
    key = (tuple(lay.seq_stride), tuple(lay.qual_stride))
    if key not in loads:  # each library gets rows in its own layout
        loads[key] = synth.generate(cfg, n, device="cuda", layout=lay)
    e.workload = loads[key]
    e.set_barcodes(e.workload.barcode_strings())
    engines.append(e)
w = engines[0].workload
M = w.layout.mol_width
codes = torch.empty(n, dtype=torch.int16, device="cuda")
mol = torch.empty((n, max(M, 1)), dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
torch.cuda.synchronize()  # inputs were made on the default stream
res = {}
with torch.cuda.stream(st):
    for r in range(rounds):
        for lp, e in zip(libs, engines):
            for b in blocks:
                e.set_option("fast_block", b)
                for wg, wq, xv in [(a, b, c) for a in wgs for b in wqs for c in xvals]:
                    e.set_option("fast_workgroups_per_cu", wg)
                    if xname:
                        try:
                            e.set_option(xname, xv)
                        except Exception:
                            if xv:
                                continue  # this build does not know the option: only its default arm runs
                    wq = wq * 10 + xv if xname else wq  # (shown in the wq column as <wq><value>)
                    if not xname and (wq or len(wqs) > 1):  # (the work-queue launch form was removed in r04: only builds given through TUNE_LIBS know it)
                        try:
                            e.set_option("work_queue", wq)
                        except Exception:
                            continue
                    for i in range(reps + 1):
                        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(st)
                        w = e.workload
                        for _ in range(burst):
                            e.demux_device(n, [t.data_ptr() for t in w.seq], [t.data_ptr() for t in w.qual],
                                           codes.data_ptr(), mol.data_ptr() if M else None, stream=st.cuda_stream)
                        z.record(st)
                        z.synchronize()
                        if i:
                            res.setdefault((os.path.basename(lp), b, wg, wq), []).append(a.elapsed_time(z) / burst)
                    assert torch.equal(codes.view(torch.int16).to(torch.int32) & 0xFFFF, w.expected) or os.environ.get("TUNE_NOCHECK")
                    if M and not os.environ.get("TUNE_NOCHECK"):  # molecular bytes: the columns behind the barcode in both index reads
                        iw, L = synth.CONFIGS[cfg].get("iw", 8), synth.CONFIGS[cfg]["read_len"]
                        want = w.seq[0][:, iw:L] if synth.CONFIGS[cfg].get("mol1_only") else torch.cat([w.seq[0][:, iw:L], w.seq[1][:, iw:L]], dim=1)
                        assert torch.equal(mol, want), (lp, b, wg, wq)
                        mol.zero_()
B = synth.ALGO_BYTES[cfg]
print("%s n=%d  algorithmic %d B/pair" % (cfg, n, B))
print("%-24s %5s %4s %3s %9s %9s %9s" % ("lib", "block", "wg", "wq", "min ms", "med ms", "GB/s(med)"))
for (lp, b, wg, wq), v in sorted(res.items(), key=lambda kv: np.median(kv[1])):
    print("%-24s %5d %4d %3d %9.4f %9.4f %9.0f" % (lp, b, wg, wq, min(v), np.median(v), n * B / np.median(v) / 1e6))
