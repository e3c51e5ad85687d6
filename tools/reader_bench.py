#!/usr/bin/env python3
"""Native reader alone (no GPU) on the four input formats of the end-to-end table: BGZF, one gzip member (what the
reference's own fixtures are), 8 MB gzip members, with the parallel gunzip on and off; one file, two, and the four
files of a chunk at once.  Reports text MB/s per file, records/s of the slowest, CPU seconds per GB of text.
usage: python tools/reader_bench.py [pairs] [gz level]"""
import os
import resource
import shutil
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quade_amd import hip_backend as hb, synth  # noqa: E402
from quade_amd.fastq_reader import FastqStream  # noqa: E402
from quade_amd.fastq_writer import host_cores, io_threads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lib = hb.load_library()
work = tempfile.mkdtemp(prefix="quade_reader_")


def cpu():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


DEV = int(os.environ.get("READER_DEVICE_INFLATE", "-1"))  # >= 0: BGZF runs are inflated on that GPU (qd_reader_open_on)


def read_all(path, res, i):
    st = FastqStream(path, 250_000, inflate_device=DEV)
    nb, t0 = 0, time.perf_counter()
    while True:
        b = st.take()
        nb += b.text.size
        m = b.n
        b.release()
        if m < 250_000:
            break
    res[i] = (nb, time.perf_counter() - t0, st.gunzip_stats())
    st.close()


try:
    print("usable cores", host_cores(), "io pool", io_threads(), "pairs", n, "gzip level", level)
    for fmt, member in ((("BGZF", "bgzf"),) if os.environ.get("READER_ONLY_BGZF") else (("BGZF", "bgzf"), ("one member", 0), ("8 MB members", 8 << 20))):
        d = os.path.join(work, fmt.replace(" ", "_"))
        os.mkdir(d)
        paths, _ = synth.write_fastq_dataset(d, n, gz_level=level, member_bytes=member)
        for par in ((1,) if member == "bgzf" else (1, 0)):
            lib.qd_io_set_option(b"parallel_gunzip", par)
            for files in (["seq_R1"], ["seq_R1", "seq_R2"], ["seq_R1", "seq_R2", "index_R1", "index_R2"]):
                res = [None] * len(files)
                th = [threading.Thread(target=read_all, args=(paths[f], res, i)) for i, f in enumerate(files)]
                c0, t0 = cpu(), time.perf_counter()
                [t.start() for t in th]
                [t.join() for t in th]
                dt, dc = time.perf_counter() - t0, cpu() - c0
                total = sum(r[0] for r in res)
                print("%-13s %s x%d: %s MB/s of text per file | %.2f M records/s | %.2f GB/s total | %.2f CPU-s per GB | chunks (parallel, serial) %s" % (
                    fmt, "parallel" if par else "1 thread", len(files), [int(r[0] / r[1] / 1e6) for r in res], n / dt / 1e6,
                    total / dt / 1e9, dc / (total / 1e9), res[0][2]), flush=True)
        lib.qd_io_set_option(b"parallel_gunzip", 1)
        shutil.rmtree(d, ignore_errors=True)
finally:
    shutil.rmtree(work, ignore_errors=True)
