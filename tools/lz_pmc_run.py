#!/usr/bin/env python3
"""Two launches of the device's level-1 coder on 64 MB of binned-quality fastq text (32 pieces of 2 MB), for rocprofv3 --pmc passes
(profiles/r03_lz_kernel_pmc.txt).  usage: rocprofv3 --pmc <counters> -- python3 tools/lz_pmc_run.py"""
import ctypes as C
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "1"]
from quade_amd import hip_backend as hb  # noqa: E402

lib = hb.load_library()
rng = np.random.default_rng(3)
n = 64 << 20
nrec = n // 340 + 1
seqs = rng.choice(np.frombuffer(b"ACGT", np.uint8), (nrec, 150))
quals = np.full((nrec, 150), ord("F"), np.uint8)
cols = np.arange(150)[None, :]
for _ in range(3):
    a, w = rng.integers(0, 150, nrec)[:, None], rng.integers(0, 12, nrec)[:, None]
    quals = np.where((cols >= a) & (cols < a + w), rng.choice(np.frombuffer(b":,#", np.uint8), nrec)[:, None], quals)
out = []
for i in range(nrec):
    out.append(b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ACGTACGT+TTGCAATC\n" % (1101 + i // 3000, 1000 + (i * 37) % 30000, 1000 + (i * 101) % 35000))
    out.append(seqs[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")
text = np.frombuffer(b"".join(out), np.uint8)[:n].copy()
assert text.size == n
PIECE = 2 << 20
k = n // PIECE
d = C.c_void_p()
assert lib.qd_deflater_create(0, C.byref(d)) == 0 and lib.qd_deflater_set_level(d, 1) == 0
ptrs = (C.c_void_p * k)(*[text.ctypes.data + i * PIECE for i in range(k)])
lens = np.full(k, PIECE, np.int64)
crc = np.array([zlib.crc32(text[i * PIECE:(i + 1) * PIECE]) for i in range(k)], np.uint32)
stride = lib.qd_huffman_member_bound(PIECE)
buf = np.zeros(k * stride, np.uint8)
ml = np.zeros(k, np.int64)
for _ in range(2):
    assert lib.qd_deflater_run(d, k, ptrs, hb._ptr(lens), hb._ptr(crc), 0, hb._ptr(buf), stride, hb._ptr(ml)) == 0
assert zlib.decompress(bytes(buf[:int(ml[0])]), 31) == bytes(text[:PIECE])
print("64 MB of text -> %.2f %%" % (100.0 * ml.sum() / n))
lib.qd_deflater_destroy(d)
