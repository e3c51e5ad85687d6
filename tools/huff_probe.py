#!/usr/bin/env python3
"""Go / no-go probe for a GPU output stage: Huffman-only gzip members (the host's `gzip_level : -1` form) made by a
kernel (tools/huff_probe.hip) from formatted 2x150 bp fastq text, against the library's 16-core pool on the same
text.  Every member is checked with zlib.  usage: python tools/huff_probe.py [MB of text]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import hip_backend as hb, synth  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
so = os.path.join(ROOT, "tools", "libhuff_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "huff_probe.hip")])
lib = hb.load_library()
probe = C.CDLL(so)
probe.huff_probe.restype = C.c_int
probe.huff_probe.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
with tempfile.TemporaryDirectory() as d:
    n_pairs = mb * (1 << 20) // 349 + 1
    paths, _ = synth.write_fastq_dataset(d, n_pairs, plain=True)
    text = np.fromfile(paths["seq_R1"], dtype=np.uint8)[:mb << 20]
n = text.size
piece = 2 << 20
npieces = (n + piece - 1) // piece
print("text: %.0f MB of formatted 2x150 bp records, %d pieces of 2 MB; host cores %d" % (n / 1e6, npieces, lib.qd_host_cores()))
t0 = time.time()
crc = np.array([zlib.crc32(text[i * piece:(i + 1) * piece]) for i in range(npieces)], dtype=np.uint32)
t_crc = time.time() - t0
stride = piece * 9 // 8 + 2048
out = np.zeros(npieces * stride, dtype=np.uint8)
sizes = np.zeros(npieces, dtype=np.uint32)
ms = np.zeros(2, dtype=np.float64)
r = probe.huff_probe(text.ctypes.data, n, piece, crc.ctypes.data, 5, out.ctypes.data, stride, sizes.ctypes.data, ms.ctypes.data)
assert r == npieces, r
bad = 0
for i in range(npieces):
    m = bytes(out[i * stride:i * stride + int(sizes[i])])
    try:
        ok = zlib.decompress(m, 31) == bytes(text[i * piece:(i + 1) * piece])
    except Exception as e:
        ok = False
        if bad == 0:
            print("piece", i, "does not inflate:", e)
    bad += not ok
print("members checked with zlib: %d of %d good; compressed to %.1f %% of the text" % (npieces - bad, npieces, 100.0 * sizes.sum() / n))
print("GPU kernel alone (text resident in HBM): %.3f ms = %.1f GB/s of text" % (ms[0], n / ms[0] / 1e6))
print("GPU whole trip (pinned text -> H2D -> kernel -> D2H of the members, 3 streams): %.1f ms = %.1f GB/s of text" % (ms[1], n / ms[1] / 1e6))
print("host: zlib.crc32 of the text on ONE python thread %.2f s = %.1f GB/s (the members' CRC-32 stays on the host)" % (t_crc, n / t_crc / 1e9))
for level, what in ((-1, "Huffman only (huffman_member)"), (1, "libdeflate level 1"), (6, "libdeflate level 6")):
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        assert lib.qd_write_gzip_file(b"/dev/null", text.ctypes.data, n, level, piece) == 0
        best = min(best, time.time() - t0)
    print("host pool (%d threads), %s, 2 MB members to /dev/null: %.1f ms = %.2f GB/s of text" % (lib.qd_io_threads(-1), what, best * 1e3, n / best / 1e9))
assert bad == 0
