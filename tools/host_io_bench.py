#!/usr/bin/env python3
"""Host-side stage rates (no GPU): libdeflate on the synthetic record text, the native reader alone,
the native sink alone -- to see which stage bounds the end-to-end rate.  usage: python tools/host_io_bench.py [pairs]"""
import ctypes as C
import gzip
import os
import shutil
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import hip_backend as hb, synth  # noqa: E402
from quade_amd.fastq_reader import FastqStream  # noqa: E402
from quade_amd.fastq_writer import FastqSink, host_cores, io_backend, io_threads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
work = tempfile.mkdtemp(prefix="quade_hostio_")
try:
    cores = host_cores()
    print("usable cores", cores, "io pool", io_threads(), io_backend())
    paths, bcs = synth.write_fastq_dataset(work, n)
    text = gzip.open(paths["seq_R1"]).read(64 << 20)
    L = C.CDLL("libdeflate.so.0")
    L.libdeflate_alloc_compressor.restype = C.c_void_p
    L.libdeflate_gzip_compress.restype = C.c_size_t
    L.libdeflate_gzip_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]

    def compress_loop(res, i, reps):
        c = L.libdeflate_alloc_compressor(1)
        out = C.create_string_buffer(len(text) + 4096)
        t0 = time.perf_counter()
        for _ in range(reps):
            L.libdeflate_gzip_compress(c, text, len(text), out, len(out))
        res[i] = len(text) * reps / (time.perf_counter() - t0) / 1e6

    for T in (1, cores):
        res = [0] * T
        th = [threading.Thread(target=compress_loop, args=(res, i, 2)) for i in range(T)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        print("libdeflate L1 compress, %2d threads: %.0f MB/s per thread, %.0f MB/s total" % (T, sum(res) / T, len(text) * 2 * T / (time.perf_counter() - t0) / 1e6))

    def read_all(path, res, i):
        st = FastqStream(path, 250_000)
        nb, t0 = 0, time.perf_counter()
        while True:
            b = st.take()
            nb += b.text.size
            m = b.n
            b.release()
            if m < 250_000:
                break
        st.close()
        res[i] = nb / (time.perf_counter() - t0) / 1e6

    for files in (["seq_R1"], ["seq_R1", "seq_R2"], ["seq_R1", "seq_R2", "index_R1", "index_R2"] * 2):
        res = [0] * len(files)
        th = [threading.Thread(target=read_all, args=(paths[f], res, i)) for i, f in enumerate(files)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        print("reader x%d: %s MB/s of text each; %.2f M records/s of the slowest" % (len(files), [int(r) for r in res[:4]], n / (time.perf_counter() - t0) / 1e6))

    # sink alone: the batch texts come from the reader, codes are random over the sample table
    st1, st2 = FastqStream(paths["seq_R1"], n), FastqStream(paths["seq_R2"], n)
    b1, b2 = st1.take(), st2.take()
    rng = np.random.default_rng(1)
    S = len(bcs)
    codes = rng.integers(0, 2 * S + 20, n).astype(np.uint16)
    codes[codes >= 2 * S] = 0xFFFF
    tags = np.zeros((n, 18), np.uint8)
    tags[:] = np.frombuffer(b":ACGTACGTACGTACGT0", np.uint8)
    tl = np.full(n, 17, np.uint8)
    out = os.path.join(work, "out")
    os.mkdir(out)
    for level in (1, 6):
        sink = FastqSink(out, ["S%d" % i for i in range(S)], level, quiet=True)
        reps = 4
        t0 = time.perf_counter()
        for _ in range(reps):
            sink.route(n, codes, b1.text, b1.off, b2.text, b2.off, tags, tl)
        t_route = time.perf_counter() - t0
        sink.close()
        dt = time.perf_counter() - t0
        print("sink alone, level %d: %.2f M pairs/s (route calls returned after %.2f s of %.2f s), %s" % (level, n * reps / dt / 1e6, t_route, dt, sink and ""))
    b1.release(); b2.release(); st1.close(); st2.close()
finally:
    shutil.rmtree(work, ignore_errors=True)
