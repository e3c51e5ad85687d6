# -*- coding: utf-8 -*-
"""
Chunk reader: streams a fastq(.gz) file as batches of whole records.

Replaces what the reference draws from pyFastq.FastqReader one record at a time
(src/Quade.py:203-214).  A batch is a byte buffer plus the offsets of its kept records, found by
the native scanner qd_fastq_index (a record whose sequence and quality lengths differ is skipped
inside its own stream -- SURVEY.md F6).  No per-read Python objects are built.
"""
from __future__ import annotations

import gzip

import numpy as np

from . import hip_backend as hb


class FastqStream(object):
    def __init__(self, path, read_bytes=32 << 20):
        self.path = path
        self._fh = gzip.open(path, "rb") if path.lower().endswith(".gz") else open(path, "rb")
        self._buf = b""
        self._eof = False
        self._read_bytes = read_bytes
        self._avg = 0.0  # running bytes per record, to size reads

    def _fill(self, want_bytes):
        while not self._eof and len(self._buf) < want_bytes:
            chunk = self._fh.read(max(self._read_bytes, want_bytes - len(self._buf)))
            if not chunk:
                self._eof = True
                if self._buf and not self._buf.endswith(b"\n"):
                    self._buf += b"\n"  # a last line without newline still ends a record
                break
            self._buf += chunk

    def take(self, max_records):
        """Up to max_records kept records -> (text uint8 array, rec_off int64[n+1]).
        Fewer than max_records only at the end of the file."""
        want = int(max_records * (self._avg or 128) * 1.05) + 4096
        while True:
            self._fill(want)
            buf = np.frombuffer(self._buf, dtype=np.uint8)
            off, consumed = hb.fastq_index(buf, max_records)
            n = off.size - 1
            if n == max_records or self._eof:
                break
            want = max(want * 2, len(self._buf) + self._read_bytes)
        text = buf[:consumed]
        self._buf = self._buf[consumed:]
        if n:
            self._avg = consumed / n
        return text, off

    def take_packed(self, max_records, layout, k, seq_rows, qual_rows, len_rows):
        """Same scan, but the records' index windows are packed straight into the given
        (pinned) row buffers.  Returns (n, all_full)."""
        want = int(max_records * (self._avg or 64) * 1.05) + 4096
        while True:
            self._fill(want)
            buf = np.frombuffer(self._buf, dtype=np.uint8)
            n, full, consumed = hb.pack_index_fastq(layout, k, buf, seq_rows, qual_rows, len_rows, max_records)
            if n == max_records or self._eof:
                break
            want = max(want * 2, len(self._buf) + self._read_bytes)
        self._buf = self._buf[consumed:]
        if n:
            self._avg = consumed / n
        return n, full

    def close(self):
        self._fh.close()
