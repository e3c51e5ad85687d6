#!/usr/bin/env python3
"""`gzip_level : 1` on the device (LZ77 + dynamic Huffman, quade_deflate.hip) against the host's pool on the same formatted
fastq text: member sizes (device level 1 / device Huffman only / libdeflate 1 / libdeflate 6 / zlib 1) and rates
(qd_deflater_run = pinned text -> H2D -> kernels -> D2H of the members, one deflater, batches of 32 pieces; the pool through
qd_write_gzip_file to /dev/null).  Two texts: the synthetic dataset of the benchmarks (uniform random qualities: nothing to
match but the names) and records with Illumina-style names and binned qualities in runs.  Every device member is checked
with zlib.  usage: python tools/lz_bench.py [MB of text per flavour]"""
import ctypes as C
import os
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import hip_backend as hb, synth  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = hb.load_library()
PIECE = 2 << 20


def binned_text(n_bytes, seed=3):
    rng = np.random.default_rng(seed)
    n = n_bytes // 370 + 1
    seqs = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n, 150))
    quals = np.full((n, 150), ord("F"), np.uint8)
    for _ in range(3):  # up to three low-quality stretches per read
        a = rng.integers(0, 150, n)
        w = rng.integers(0, 12, n)
        c = rng.choice(np.frombuffer(b":,#", np.uint8), n)
        cols = np.arange(150)[None, :]
        m = (cols >= a[:, None]) & (cols < (a + w)[:, None])
        quals = np.where(m, c[:, None], quals)
    x = 1000 + (np.arange(n) * 37) % 30000
    y = 1000 + (np.arange(n) * 101) % 35000
    out = []
    for i in range(n):
        out.append(b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ACGTACGT+TTGCAATC\n" % (1101 + i // 3000, x[i], y[i]))
        out.append(seqs[i].tobytes())
        out.append(b"\n+\n")
        out.append(quals[i].tobytes())
        out.append(b"\n")
    return np.frombuffer(b"".join(out), np.uint8)[:n_bytes].copy()


def device(text, level):
    n = text.size
    npieces = (n + PIECE - 1) // PIECE
    d = C.c_void_p()
    assert lib.qd_deflater_create(0, C.byref(d)) == 0
    assert lib.qd_deflater_set_level(d, level) == 0
    pin = lib.qd_pinned_alloc(n + 64)
    C.memmove(pin, text.ctypes.data, n)
    stride = lib.qd_huffman_member_bound(PIECE)
    total, secs, bad = 0, [], 0
    for rep in range(3):
        total = 0
        t_all = 0.0
        for b0 in range(0, npieces, 32):
            k = min(32, npieces - b0)
            ptrs = (C.c_void_p * k)(*[pin + (b0 + i) * PIECE for i in range(k)])
            lens = np.array([min(PIECE, n - (b0 + i) * PIECE) for i in range(k)], np.int64)
            crc = np.array([zlib.crc32(text[(b0 + i) * PIECE:(b0 + i) * PIECE + int(lens[i])]) for i in range(k)], np.uint32)
            out = np.zeros(k * stride, np.uint8)
            ml = np.zeros(k, np.int64)
            t0 = time.time()
            rc = lib.qd_deflater_run(d, k, ptrs, hb._ptr(lens), hb._ptr(crc), 1, hb._ptr(out), stride, hb._ptr(ml))
            t_all += time.time() - t0
            assert rc == 0, lib.qd_deflater_last_error(d)
            total += int(ml.sum())
            if rep == 0:
                for i in range(k):
                    m = bytes(out[i * stride:i * stride + int(ml[i])])
                    a = (b0 + i) * PIECE
                    try:
                        ok = ml[i] > 0 and zlib.decompress(m, 31) == bytes(text[a:a + int(lens[i])])
                    except Exception as e:  # noqa: BLE001
                        ok = False
                        if bad == 0:
                            print("  piece", b0 + i, "does not inflate:", e)
                    bad += not ok
        secs.append(t_all)
    lib.qd_pinned_free(pin)
    lib.qd_deflater_destroy(d)
    return total, min(secs), bad, npieces


def host(text, level):
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x.gz").encode()
        assert lib.qd_write_gzip_file(p, text.ctypes.data, text.size, level, PIECE) == 0
        size = os.path.getsize(p)
    best = 1e9
    for _ in range(2):
        t0 = time.time()
        assert lib.qd_write_gzip_file(b"/dev/null", text.ctypes.data, text.size, level, PIECE) == 0
        best = min(best, time.time() - t0)
    return size, best


with tempfile.TemporaryDirectory() as d:
    paths, _ = synth.write_fastq_dataset(d, mb * (1 << 20) // 349 + 1, plain=True)
    uniform = np.fromfile(paths["seq_R1"], dtype=np.uint8)[:mb << 20].copy()
print("host cores %d, pool threads %d, %d MB of text per flavour, pieces of 2 MB" % (lib.qd_host_cores(), lib.qd_io_threads(-1), mb))
for name, text in (("synthetic dataset (uniform random qualities)", uniform), ("Illumina-style names, binned qualities in runs", binned_text(mb << 20))):
    n = text.size
    print(name)
    for level, what in ((1, "device LZ77 + Huffman (gzip_level 1)"), (-1, "device Huffman only (gzip_level -1)")):
        size, sec, bad, npieces = device(text, level)
        print("  %-40s %6.2f %% of the text   %7.2f GB/s of text through qd_deflater_run   (%d of %d members good)"
              % (what, 100.0 * size / n, n / sec / 1e9, npieces - bad, npieces))
        assert bad == 0
    for level, what in ((1, "host pool, libdeflate level 1"), (6, "host pool, libdeflate level 6"), (-1, "host pool, Huffman only")):
        size, sec = host(text, level)
        print("  %-40s %6.2f %% of the text   %7.2f GB/s of text" % (what, 100.0 * size / n, n / sec / 1e9))
    t0 = time.time()
    z1 = sum(len(zlib.compress(bytes(text[a:a + PIECE]), 1)) for a in range(0, min(n, 32 << 20), PIECE))
    print("  %-40s %6.2f %% of the text   (first 32 MB, one python thread)" % ("zlib level 1", 100.0 * z1 / min(n, 32 << 20)))
