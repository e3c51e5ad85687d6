#!/usr/bin/env python3
"""BGZF inflate: the device inflater (QUADE_INFLATE_FORM=1: one wave per block, 2 = the default: 512 lanes per block) against libdeflate on host threads, on synthetic 2x150 bp
fastq text.  usage: python tools/inflate_bench.py [MB of text] [run MB of compressed bytes]"""
import os
import struct
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import hip_backend as hb  # noqa: E402
from quade_amd.fastq_reader import FastqStream  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
run_mb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(1)
rec = []
n = 0
while n < mb << 20:
    L = 150
    r = b"@SIM:1:FC:1:%d:%d 1:N:0:\n%s\n+\n%s\n" % (len(rec), len(rec) * 7, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), L)),
                                                  bytes(rng.integers(35, 74, L).astype(np.uint8)))
    rec.append(r)
    n += len(r)
text = b"".join(rec)
lib = hb.load_library()
d = tempfile.mkdtemp(prefix="quade_inflate_")
path = os.path.join(d, "x.fastq.gz")
src = np.frombuffer(text, dtype=np.uint8)
assert lib.qd_write_gzip_file(path.encode(), hb._ptr(src), len(src), 1, -1) == hb.QD_OK
comp = open(path, "rb").read()
offs, pos = [], 0
while pos < len(comp):
    offs.append(pos)
    pos += struct.unpack_from("<H", comp, pos + 16)[0] + 1
offs.append(len(comp))
isz = [struct.unpack_from("<I", comp, offs[i + 1] - 4)[0] for i in range(len(offs) - 1)]
print("%d MB of text, %d MB compressed, %d blocks" % (len(text) >> 20, len(comp) >> 20, len(isz)))
with hb.Inflater(0) as inf:
    for rep in range(3):
        t0 = time.perf_counter()
        i, out_total = 0, 0
        while i < len(isz):
            j = i
            while j < len(isz) and offs[j + 1] - offs[i] <= run_mb << 20:
                j += 1
            j = max(j, i + 1)
            out = inf.run(comp[offs[i]:offs[j]], sum(isz[i:j]))
            out_total += len(out)
            i = j
        dt = time.perf_counter() - t0
        print("device inflater, runs of %d MB: %.3f s  %.2f GB/s of text (staging copies, H2D, kernel, D2H, CRC32 included)" % (run_mb, dt, out_total / dt / 1e9))
for dev in (-1, 0, -1, 0):
    c0 = os.times()
    t0 = time.perf_counter()
    st = FastqStream(path, 500_000, inflate_device=dev)
    nrec = 0
    while True:
        b = st.take()
        if b.n == 0:
            break
        nrec += b.n
        b.release()
    stats = st.inflate_stats()
    st.close()
    dt = time.perf_counter() - t0
    c1 = os.times()
    cpu = (c1.user - c0.user) + (c1.system - c0.system)
    print("native reader, inflate on %s: %.3f s  %.2f GB/s of text, %d records, runs device/host %s, CPU %.2f s = %.2f core-s per GB of text (user %.2f sys %.2f)"
          % ("the device" if dev >= 0 else "host threads", dt, len(text) / dt / 1e9, nrec, stats, cpu, cpu / (len(text) / 1e9), c1.user - c0.user, c1.system - c0.system))
os.remove(path)
os.rmdir(d)
