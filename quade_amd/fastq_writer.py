# -*- coding: utf-8 -*-
"""
Output side of the demultiplexer: one pair of fastq.gz files per destination.

Keeps what the reference's FastqWriter fixes (src/FastqWriter.py): file names
<name>_R1.fastq.gz / <name>_R2.fastq.gz (:29-31), creation at the first routed pair only (:55-57),
truncation of a pre-existing file at that moment (:76-81), appended gzip members afterwards
(:83-90).  What changes is granularity: a call takes the already formatted records of a whole
batch (qd_format_records) instead of one FastqSeq pair, so there is one gzip member per batch
instead of one per 20 pairs; the decompressed bytes are identical.
"""
from __future__ import annotations

import os
import zlib
from collections import deque
from concurrent.futures import ThreadPoolExecutor

# gzip members are compressed on a shared thread pool (zlib releases the GIL) and appended to their
# file strictly in submission order, so the decompressed stream keeps the input order.
_POOL = None
_POOL_THREADS = max(1, min(32, (os.cpu_count() or 2) - 1))


def _pool():
    global _POOL
    if _POOL is None:
        _POOL = ThreadPoolExecutor(max_workers=_POOL_THREADS, thread_name_prefix="quade-gzip")
    return _POOL


def _gzip_member(data, level):
    co = zlib.compressobj(level, zlib.DEFLATED, 31)  # 31 = gzip framing
    return co.compress(data) + co.flush()


class FastqWriter(object):
    def __init__(self, name="Unknown", outdir=".", gzip_level=6):
        self.R1_fastq_name = os.path.join(outdir, name + "_R1.fastq.gz")
        self.R2_fastq_name = os.path.join(outdir, name + "_R2.fastq.gz")
        self.gzip_level = gzip_level
        self.counter = -1  # -1 = files not created yet, as in the reference
        self._fh = None
        self._pending = deque()  # (future_R1, future_R2) in submission order

    def __repr__(self):
        return "<Instance of {} from {} >\n".format(self.__class__.__name__, self.__module__)

    def __call__(self, records_R1: bytes, records_R2: bytes, n_pairs: int):
        """Append the formatted records of n_pairs routed pairs."""
        if n_pairs == 0:
            return
        if self.counter == -1:
            self.init_files()
            self.counter = 0
        self.counter += n_pairs
        pool = _pool()
        self._pending.append((pool.submit(_gzip_member, records_R1, self.gzip_level),
                              pool.submit(_gzip_member, records_R2, self.gzip_level)))
        self._drain(block=len(self._pending) > 8)

    def _drain(self, block):
        """Write finished members, oldest first; never out of order."""
        while self._pending:
            f1, f2 = self._pending[0]
            if not block and not (f1.done() and f2.done()):
                break
            self._fh[0].write(f1.result())
            self._fh[1].write(f2.result())
            self._pending.popleft()

    def init_files(self):
        print("\tCreate {} file".format(self.R1_fastq_name))
        print("\tCreate {} file".format(self.R2_fastq_name))
        self._fh = (open(self.R1_fastq_name, "wb"), open(self.R2_fastq_name, "wb"))

    def flush_buffers(self):
        if self._fh:
            self._drain(block=True)
            for fh in self._fh:
                fh.flush()

    def close(self):
        if self._fh:
            self._drain(block=True)
            for fh in self._fh:
                fh.close()
            self._fh = None
