#!/usr/bin/env python
"""Parallel gunzip of ONE ordinary gzip member (quade_pgz.cpp through qd_gunzip_buffer) against zlib on one thread:
synthetic 2x150 bp fastq text (SURVEY 8d recipe: uniform bases and qualities, the worst case for deflate) and a
'binned' variant (4 quality values in runs, as current instruments write them: long matches, markers persist).
Usage: python tools/gunzip_bench.py [pairs] [io_threads]"""
import ctypes as C
import gzip
import os
import sys
import tempfile
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quade_amd import hip_backend as hb  # noqa: E402
from quade_amd import synth  # noqa: E402


def text_of(n_pairs, binned):
    with tempfile.TemporaryDirectory() as d:
        paths, _ = synth.write_fastq_dataset(d, n_pairs, plain=True)
        text = open(paths["seq_R1"], "rb").read()
    if binned:  # qualities in runs of four values: what RTA3 / NovaSeq write
        a = np.frombuffer(text, np.uint8).copy()
        rec = a.reshape(n_pairs, -1)
        L = 150
        q0 = rec.shape[1] - 1 - L
        rng = np.random.default_rng(3)
        runs = rng.choice(np.frombuffer(b"F:,#", np.uint8), size=(n_pairs, L // 10), p=[0.8, 0.1, 0.07, 0.03])
        rec[:, q0:q0 + L] = np.repeat(runs, 10, axis=1)
        text = a.tobytes()
    return text


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    lib = hb.load_library()
    if len(sys.argv) > 2:
        lib.qd_io_threads(int(sys.argv[2]))
    print("io threads %d, host cores %d" % (lib.qd_io_threads(-1), lib.qd_host_cores()))
    for binned in (False, True):
        text = text_of(n_pairs, binned)
        for level in (1, 6):
            comp = zlib.compressobj(level, zlib.DEFLATED, 31).compress(text)
            c = zlib.compressobj(level, zlib.DEFLATED, 31)
            comp = c.compress(text) + c.flush()
            t0 = time.time()
            ref = zlib.decompress(comp, 31)
            t_zlib = time.time() - t0
            assert ref == text
            src = np.frombuffer(comp, np.uint8)
            out = np.empty(len(text) + 64, np.uint8)
            line = "%s level %d: %.0f MB text, %.0f MB gz | zlib 1 thread %.2f GB/s" % (
                "binned" if binned else "uniform", level, len(text) / 1e6, len(comp) / 1e6, len(text) / t_zlib / 1e9)
            for chunk in (1 << 20, 4 << 20, 8 << 20):
                best = 1e9
                for _ in range(3):
                    n = C.c_int64(0)
                    st = np.zeros(5, np.int64)
                    t0 = time.time()
                    rc = lib.qd_gunzip_buffer(hb._ptr(src), len(comp), chunk, hb._ptr(out), out.size, C.byref(n), hb._ptr(st))
                    best = min(best, time.time() - t0)
                    assert rc == 0 and n.value == len(text), (rc, lib.qd_gunzip_last_error())
                assert bytes(out[:n.value]) == text
                line += " | chunk %d MB: %.2f GB/s (%d/%d chunks parallel)" % (chunk >> 20, len(text) / best / 1e9, st[1], st[0])
            print(line, flush=True)


if __name__ == "__main__":
    main()
